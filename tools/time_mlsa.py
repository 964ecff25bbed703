"""MLSA filter modes (256 utterances x 1 s, order 24): wall time per call, for A/B runs."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = "cuda"
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
x = torch.randn(256, 16000, device=dev)
mc = 0.1 * torch.randn(256, 200, 25, device=dev)
out = []
with torch.no_grad():
    for mode in ("multi-stage", "single-stage"):
        f = dsp.PseudoMGLSADigitalFilter(24, 80, alpha=0.42, mode=mode, device=dev)
        out.append(f"{mode} {timeit(lambda: f(x, mc)):.2f} ms")
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: " + " | ".join(out))
