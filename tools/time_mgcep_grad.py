"""Mel-generalized cepstral analysis (gamma = -0.5) WITH a gradient: float32 (fused step forward / backward) against the float64
module (differentiable operator chain) on 512 frames, and the time of forward + backward at 51 200 frames."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
g = torch.Generator().manual_seed(0)
X = (torch.randn(51200, 257, generator=g).square() + 0.1).to(dev)
w = torch.randn(25, generator=g).to(dev)
for n_iter in (1, 3, 10):
    m32 = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=n_iter, device=dev)
    m64 = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=n_iter, device=dev, dtype=torch.float64)
    xs = X[:512].clone().requires_grad_(True)
    (m32(xs) * w).sum().backward()
    xd = X[:512].double().requires_grad_(True)
    (m64(xd) * w.double()).sum().backward()
    err = float((xs.grad.double() - xd.grad).abs().max() / xd.grad.abs().max())
    print(f"n_iter {n_iter}: max |grad32 - grad64| / max |grad64| = {err:.2e}")
def fb():
    xg = X.clone().requires_grad_(True)
    (m32(xg) * w).sum().backward()
for _ in range(2): fb()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): fb()
e1.record(); torch.cuda.synchronize()
print("forward + backward, 51 200 frames, n_iter 10: %.2f ms" % (e0.elapsed_time(e1) / 3))
