"""A few forward passes of MelGeneralizedCepstralAnalysis (gamma = -0.5, 10 iterations) at 51 200 frames (for kernel traces)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
x = torch.randn(256, 16000, device=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=10, device=dev)
with torch.no_grad():
    X = stft(x)
    for _ in range(5):
        mc = mg(X)
torch.cuda.synchronize()
