"""Forward + backward of the mel-generalized cepstral analysis (gamma = -0.5, 10 steps) at 51 200 frames (for traces)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
X = (torch.randn(51200, 257, generator=torch.Generator().manual_seed(0)).square() + 0.1).to(dev)
mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=10, device=dev)
for _ in range(3):
    xg = X.clone().requires_grad_(True)
    mg(xg).sum().backward()
torch.cuda.synchronize()
