"""Per-module timings of the README LPC chain  lpc(window(frame(x)))  (dev tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
from diffsptk_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.randn(B, 16000, device="cuda")
fr, wn, lpc = dsp.Frame(400, 80), dsp.Window(400, device="cuda"), dsp.LPC(400, 24, eps=1e-5, device="cuda")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    f = fr(x); k1 = _lib.last_kernel()
    w = wn(f); k2 = _lib.last_kernel()
    a = lpc(w); k3 = _lib.last_kernel()
    t1, t2, t3 = timeit(lambda: fr(x)), timeit(lambda: wn(f)), timeit(lambda: lpc(w))
gb = f.numel() * 4 / 1e6
print(f"B={B}: frame {t1:.3f} ms [{k1}] ({(x.numel()*4/1e6+gb)/t1:.0f} GB/s) | window {t2:.3f} ms [{k2}] ({2*gb/t2:.0f} GB/s) | lpc {t3:.3f} ms [{k3}] ({gb/t3:.0f} GB/s)")
fg = f.clone().requires_grad_(True)
y = lpc(wn(fg))
g = torch.ones_like(y)
tb = timeit(lambda: torch.autograd.grad(y, fg, g, retain_graph=True))
print(f"   window+lpc backward {tb:.3f} ms")
