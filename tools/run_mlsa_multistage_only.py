import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
B, N, P, M = 256, 200, 80, 24
g = torch.Generator().manual_seed(0)
x = torch.randn(B, N * P, generator=g).to(dev)
mc = (0.1 * torch.randn(B, N, M + 1, generator=g)).to(dev)
with torch.no_grad():
    ml = dsp.MLSA(M, P, alpha=0.42, mode="multi-stage", device=dev)
    for _ in range(3): y = ml(x, mc)
    torch.cuda.synchronize()
