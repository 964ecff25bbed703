"""The untuned analysis at the 48 kHz set-up (fft 2048 / order 49) with a gradient, 12 800 frames: for kernel traces."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
x = torch.randn(64, 48000, generator=torch.Generator().manual_seed(0)).to(dev)
stft = dsp.STFT(1200, 240, 2048, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=49, alpha=0.55, n_iter=10, device=dev)
with torch.no_grad():
    X = stft(x)
for _ in range(int(os.environ.get("N", "4"))):
    Xg = X.clone().requires_grad_(True)
    mcep(Xg).sum().backward()
torch.cuda.synchronize()
