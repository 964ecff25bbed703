"""A few forward + backward steps of STFT -> mcep (for counter collection)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
x = torch.randn(1024, 16000, device=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
for _ in range(6):
    xg = x.clone().requires_grad_(True)
    mcep(stft(xg)).mean().backward()
torch.cuda.synchronize()
