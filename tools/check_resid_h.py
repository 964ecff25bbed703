"""dsa_mcep_newton_resid_h (binary16-split chains) against dsa_mcep_newton_resid (float32 matrix instructions) and float64, and the
untuned mel-cepstral analysis through either against the float64 oracle; 48 kHz set-ups.  Times per call (12 800 frames)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd import ops
from oracle import oracle as O
DEV = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for nfft, M, alpha in ((2048, 49, 0.55), (1024, 34, 0.55), (256, 12, 0.35)):
    K = nfft // 2 + 1
    g = torch.Generator().manual_seed(nfft)
    F = 12800 if nfft >= 1024 else 3000
    X = (torch.randn(F, K, generator=g).square() * (0.1 + torch.rand(F, 1, generator=g)) + 1e-4).to(DEV)
    m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=alpha, n_iter=10, device=DEV)
    logx = torch.log(X)
    mc = ops.rows_gemm(logx, m.G) if hasattr(ops, "rows_gemm") else None
    D, E = m.D, m.E
    img = ops.mcep_resid_images(D, E)
    r_h = ops.mcep_newton_resid_h(logx, mc, img)
    r_f = ops.mcep_newton_resid(logx, mc, D, E)
    e64 = torch.exp(logx.double() - 2 * mc.double() @ D.double()) @ E.double()
    sc = float(e64.abs().max())
    print(f"nfft {nfft} M {M}: resid vs float64 (of the max): binary16 split {float((r_h.double() - e64).abs().max()) / sc:.3e}   float32 matrix {float((r_f.double() - e64).abs().max()) / sc:.3e}")
    with torch.no_grad():
        y1 = m(X)
        t1 = timeit(lambda: m(X))
        os.environ["DSA_MCEP_RESID_H"] = "0"
        y0 = m(X)
        t0 = timeit(lambda: m(X))
        del os.environ["DSA_MCEP_RESID_H"]
    sel = slice(0, F, max(1, F // 64))
    ref = O.mcep(X[sel].double().cpu().numpy(), M, alpha, 10)
    e1 = np.abs(y1[sel].double().cpu().numpy() - ref).max()
    e0 = np.abs(y0[sel].double().cpu().numpy() - ref).max()
    print(f"   analysis: max |err| vs float64 oracle: split {e1:.3e}  float32 {e0:.3e}   ms per call: split {t1:.3f}  float32 {t0:.3f}")
