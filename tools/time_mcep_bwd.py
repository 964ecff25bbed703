"""Mel-cepstral backward launch time (us per 204 800 frames, module API) for A/B runs of library builds."""
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
with torch.no_grad():
    X = stft(x)
Xd = X.detach().requires_grad_(True)
mc = mcep(Xd); g = torch.randn_like(mc)
t = timeit(lambda: torch.autograd.grad(mc, Xd, g, retain_graph=True))
with torch.no_grad():
    tf = timeit(lambda: mcep(X))
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: mcep bwd {t:.1f} | mcep fwd {tf:.1f}")
