"""STFT backward / inverse STFT launch times through the C-ABI-backed ops (us per 204 800 frames) for A/B runs."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = "cuda"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(int(os.environ.get("B", "1024")), 16000, device=dev)
st = dsp.STFT(400, 80, 512, device=dev)
xg = x.clone().requires_grad_(True); y = st(xg); g = torch.randn_like(y)
t_b = timeit(lambda: torch.autograd.grad(y, xg, g, retain_graph=True))
stc = dsp.STFT(400, 80, 512, out_format="complex", device=dev); ist = dsp.ISTFT(400, 80, 512, device=dev)
with torch.no_grad():
    Z = stc(x); t_i = timeit(lambda: ist(Z))
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: bwd {t_b:.1f} | istft {t_i:.1f}")
