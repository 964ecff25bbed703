"""The tuned forward kernels with / without the Newton history a gradient needs (iterates; iterates + rt rows): us per 204 800 frames."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(int(os.environ.get("B", "1024")), 16000, device=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
fused = dsp.fuse(stft, mcep)
with torch.no_grad():
    X = stft(x)
    for _ in range(30): mcep(X)   # clock ramp
    t0 = timeit(lambda: mcep(X)); f0 = timeit(lambda: fused(x))
Xg = X.detach().requires_grad_(True); xg = x.detach().requires_grad_(True)
t1 = timeit(lambda: mcep(Xg)); f1 = timeit(lambda: fused(xg))
os.environ["DSA_MCEP_HIST_RT"] = "0"
t2 = timeit(lambda: mcep(Xg)); f2 = timeit(lambda: fused(xg))
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: mcep fwd {t0:.1f} | + iterates {t2:.1f} | + iterates + rt {t1:.1f}   fused {f0:.1f} | {f2:.1f} | {f1:.1f}")
