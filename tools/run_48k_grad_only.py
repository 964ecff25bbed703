"""The analysis at a 48 kHz set-up (fft 2048 / order 49 by default) with a gradient, B utterances x 1 s: for kernel traces.
usage: python tools/run_48k_grad_only.py [B] [2048|1024]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nfft = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
fl, fp, M = (1200, 240, 49) if nfft == 2048 else (800, 200, 34)
x = torch.randn(B, 48000, generator=torch.Generator().manual_seed(0)).to(dev)
stft = dsp.STFT(fl, fp, nfft, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=dev)
with torch.no_grad():
    X = stft(x)
for _ in range(int(os.environ.get("N", "4"))):
    Xg = X.clone().requires_grad_(True)
    mcep(Xg).sum().backward()
torch.cuda.synchronize()
