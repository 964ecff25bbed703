"""Forward + backward of the multi-stage MLSA filter at 256 x 1 s (for traces)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(256, 16000, generator=g).to(dev)
mc = (0.1 * torch.randn(256, 200, 25, generator=g)).to(dev)
ml = dsp.MLSA(24, 80, alpha=0.42, mode="multi-stage", device=dev)
for _ in range(3):
    xg, mg = x.clone().requires_grad_(True), mc.clone().requires_grad_(True)
    ml(xg, mg).square().sum().backward()
torch.cuda.synchronize()
