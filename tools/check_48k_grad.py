"""The 48 kHz mel-cepstral analysis WITH a gradient: the one-node path (McepNewtonStepsHFn: dsa_mcep_newton_update_bwd + dsa_mcep_newton_resid_h_bwd per
step) against the composed differentiable pieces (DSA_MCEP_GRAD_H=0) and against float64 autograd of the ATen port; times per forward + backward."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd import _lib
from oracle import torch_port as TP
dev = "cuda"
g = torch.Generator().manual_seed(3)
def grads(m, X, w, flag):
    os.environ["DSA_MCEP_GRAD_H"] = flag
    Xg = X.detach().clone().requires_grad_(True)
    y = m(Xg)
    (y * w).sum().backward()
    return y.detach(), Xg.grad
for nfft, M, alpha in ((2048, 49, 0.55), (1024, 34, 0.55), (2048, 40, 0.5), (2048, 54, 0.55), (2048, 32, 0.55), (2048, 47, 0.55), (2048, 48, 0.55)):
    K = nfft // 2 + 1
    for F, n_iter in ((70, 2), (333, 10)):
        X = (torch.randn(F, K, generator=g).square() + 0.05).to(dev)
        w = torch.randn(F, M + 1, generator=g).to(dev)
        m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=alpha, n_iter=n_iter, device=dev)
        y1, g1 = grads(m, X, w, "1")
        y0, g0 = grads(m, X, w, "0")
        tab = TP.McepTables(nfft, M, alpha, torch.float64)
        Xs = X.double().cpu().requires_grad_(True)
        yr = TP.mcep(Xs, tab, n_iter)
        (yr * w.double().cpu()).sum().backward()
        gr = Xs.grad
        def rel(a, b):   # per frame: max |err| / max |ref|
            a, b = a.double().cpu(), b.double().cpu()
            return float(((a - b).abs().amax(-1) / b.abs().amax(-1).clamp_min(1e-300)).max())
        print(f"nfft {nfft} M {M} F {F} n_iter {n_iter}: value vs f64 {rel(y1, yr.detach()):.2e} | grad one-node vs f64 {rel(g1, gr):.2e}  composed vs f64 {rel(g0, gr):.2e}  "
              f"one-node vs composed {rel(g1, g0):.2e} | finite {bool(torch.isfinite(g1).all())}")
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for nfft, M, fl, fp in ((2048, 49, 1200, 240), (1024, 34, 800, 200)):
    for B in (64, 512):
        x = torch.randn(B, 48000, generator=g).to(dev)
        with torch.no_grad():
            X = dsp.STFT(fl, fp, nfft, device=dev)(x).reshape(-1, nfft // 2 + 1)
        m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=dev)
        w = torch.randn(X.size(0), M + 1, device=dev)
        for flag in ("1", "0", "1", "0"):
            t = timeit(lambda: grads(m, X, w, flag))
            print(f"{nfft} / {M}: {X.size(0)} frames DSA_MCEP_GRAD_H={flag}: {t:.2f} ms per forward + backward, peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
            torch.cuda.reset_peak_memory_stats()
