"""MLSA filter forward + backward (gradients with respect to the excitation and the mel-cepstra), 256 utterances x 1 s, float32."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
g = torch.Generator().manual_seed(0)
B = int(os.environ.get("B", 256))
x = torch.randn(B, 16000, generator=g).to(dev)
mc = (0.1 * torch.randn(B, 200, 25, generator=g)).to(dev)
def gpu_time(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mode, kw in (("multi-stage", {}), ("single-stage", {}), ("freq-domain", dict(frame_length=400, fft_length=512))):
    ml = dsp.MLSA(24, 80, alpha=0.42, mode=mode, device=dev, **kw)
    def fb():
        xg, mg = x.clone().requires_grad_(True), mc.clone().requires_grad_(True)
        ml(xg, mg).square().sum().backward()
    with torch.no_grad():
        t_f = gpu_time(lambda: ml(x, mc))
    print(f"{mode}: forward {t_f:.2f} ms, forward + backward {gpu_time(fb):.2f} ms")
