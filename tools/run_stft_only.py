"""A few launches of the stand-alone packed STFT kernel at the bench size (for counter collection; DSA_STFT_RUN picks the pass order)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
x = torch.randn(1024, 16000, device="cuda")
stft = dsp.STFT(400, 80, 512, device="cuda")
with torch.no_grad():
    for _ in range(10):
        y = stft(x)
torch.cuda.synchronize()
