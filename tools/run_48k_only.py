"""The untuned analysis at the 48 kHz set-up (frame 1200 / period 240 / fft 2048 / order 49), forward, 12 800 frames: for kernel traces."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
x = torch.randn(int(os.environ.get("B", "64")), 48000, generator=torch.Generator().manual_seed(0)).to(dev)
stft = dsp.STFT(1200, 240, 2048, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=49, alpha=0.55, n_iter=10, device=dev)
with torch.no_grad():
    X = stft(x)
    for _ in range(int(os.environ.get("N", "6"))):
        mcep(X)
torch.cuda.synchronize()
