"""The gradient of an UNCONVERGED mel-cepstral analysis at high dynamic range (tools/fuzz_big.py's inputs): error against float64 of
(a) the tuned float32 kernels, (b) the generic float32 kernels, (c) the reference's own op sequence in float32 on the host
(oracle/torch_port.py: what the reference computes in float32) -- the yardstick for what float32 can deliver here.

    python tools/check_grad_yardstick.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import diffsptk_amd as dsp  # noqa: E402
from diffsptk_amd import _lib  # noqa: E402
import torch_port  # noqa: E402  (test infrastructure: the host-side yardstick only)

dev = torch.device("cuda", 0)


def rowrel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().amax(-1) / b.abs().amax(-1).clamp_min(1e-300)).max().item()


for (nfft, M, alpha, n_iter, F, decay) in ((1024, 34, 0.0, 1, 130, 8.0), (1024, 31, 0.0, 5, 65, 8.0), (1024, 55, -0.2, 2, 130, 8.0), (2048, 49, 0.55, 10, 64, 4.0),
                                          (512, 24, 0.42, 10, 64, 8.0)):
    g = torch.Generator().manual_seed(5)
    X = (torch.rand(F, nfft // 2 + 1, generator=g) * 4 + 1e-3) ** 2 * torch.exp(-torch.linspace(0, decay, nfft // 2 + 1))
    wgt = torch.linspace(1.0, 0.3, M + 1)
    res = {}
    for name, dt, algo in (("tuned", torch.float32, None), ("generic32", torch.float32, _lib.ALGO_GENERIC), ("f64", torch.float64, None)):
        Xd = X.to(dev, dt).requires_grad_(True)
        mod = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=alpha, n_iter=n_iter, device=dev, dtype=dt)
        if algo is not None:
            mod.algo = algo
        mc = mod(Xd)
        (mc * wgt.to(dev, dt)).sum().backward()
        res[name] = (mc.detach(), Xd.grad.detach())
    for name, dt in (("reference ops float32 (host)", torch.float32), ("reference ops float64 (host)", torch.float64)):
        tab = torch_port.McepTables(nfft, M, alpha, dt)
        Xc = X.to(dt).clone().detach().requires_grad_(True)
        mc = torch_port.mcep(Xc, tab, n_iter)
        (mc * wgt.to(dt)).sum().backward()
        res[name] = (mc.detach(), Xc.grad.detach())
    ref = res["reference ops float64 (host)"]
    print(f"fft {nfft} / order {M}, alpha {alpha}, n_iter {n_iter}, {F} frames, spectral decay e^-{decay:g}:", flush=True)
    for name in ("tuned", "generic32", "reference ops float32 (host)", "f64"):
        print(f"  {name:30s} output error {rowrel(res[name][0], ref[0]):.2e}   gradient error {rowrel(res[name][1], ref[1]):.2e}  (of the row maximum, vs float64 of the reference's ops)", flush=True)
