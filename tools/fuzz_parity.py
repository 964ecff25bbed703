"""Randomised tuned-vs-generic parity sweep on the GPU (dev tool; the committed tests hold the fixed cases).
STFT forward/backward (all formats, centre, zmean, ragged T), mel-cepstral forward/backward (alpha, n_iter),
LPC forward/backward: the tuned float32 kernels against the generic float64 kernels on the same inputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import diffsptk_amd as dsp
from diffsptk_amd import _lib, ops
from diffsptk_amd.utils import tables

dev = "cuda"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = {}


def note(key, err):
    worst[key] = max(worst.get(key, 0.0), float(err))


def rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)


for case in range(n_cases):
    L = int(rng.choice([400, 512, 320, 257, 64, 401]))
    P = int(rng.choice([80, 160, 100, 37, 1 + L // 3]))
    B = int(rng.integers(1, 5))
    T = int(rng.integers(L, 9000))
    center = bool(rng.integers(0, 2))
    zmean = bool(rng.integers(0, 2))
    fmt = str(rng.choice(["power", "magnitude", "db", "log-magnitude", "complex"]))
    g = torch.Generator().manual_seed(case)
    x = torch.randn(B, T, generator=g)
    outs = {}
    for dt in (torch.float32, torch.float64):
        xd = x.to(dev, dt).requires_grad_(True)
        st = dsp.STFT(L, P, 512, center=center, zmean=zmean, out_format=fmt, device=dev, dtype=dt)
        y = st(xd)
        yr = torch.view_as_real(y) if y.is_complex() else y
        wgt = torch.linspace(0.5, 1.5, yr.size(-1), device=dev, dtype=dt)
        (yr * wgt).sum().backward()
        outs[dt] = (yr.detach(), xd.grad.detach())
    note(f"stft fwd {fmt}", rel(outs[torch.float32][0], outs[torch.float64][0]) if fmt not in ("db", "log-magnitude") else
         (outs[torch.float32][0].double() - outs[torch.float64][0]).abs().max().item())
    note(f"stft bwd {fmt}", rel(outs[torch.float32][1], outs[torch.float64][1]))

    # mel-cepstral analysis on the power spectrum
    alpha = float(rng.choice([0.0, 0.1, 0.42, 0.55, -0.3]))
    n_iter = int(rng.integers(0, 12))
    X = (torch.rand(int(rng.integers(1, 70)), 257, generator=g) * 4 + 1e-3) ** 2
    outs = {}
    for dt in (torch.float32, torch.float64):
        Xd = X.to(dev, dt).requires_grad_(True)
        mc = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=alpha, n_iter=n_iter, device=dev, dtype=dt)(Xd)
        (mc * torch.linspace(-1, 1, 25, device=dev, dtype=dt)).sum().backward()
        outs[dt] = (mc.detach(), Xd.grad.detach())
    note("mcep fwd", (outs[torch.float32][0].double() - outs[torch.float64][0]).abs().max().item())
    note("mcep bwd", rel(outs[torch.float32][1], outs[torch.float64][1]))

    # LPC on framed noise
    Lf = int(rng.choice([400, 25, 100, 512, 333]))
    fr = torch.randn(int(rng.integers(1, 300)), Lf, generator=g)
    outs = {}
    for dt in (torch.float32, torch.float64):
        fd = fr.to(dev, dt).requires_grad_(True)
        a = dsp.LPC(Lf, 24, eps=1e-5, device=dev, dtype=dt)(fd)
        (a * torch.linspace(-1, 1, 25, device=dev, dtype=dt)).sum().backward()
        outs[dt] = (a.detach(), fd.grad.detach())
    note("lpc fwd", (outs[torch.float32][0].double() - outs[torch.float64][0]).abs().max().item())
    note("lpc bwd", rel(outs[torch.float32][1], outs[torch.float64][1]))

for k in sorted(worst):
    print(f"{k:28s} worst {worst[k]:.3e}")
