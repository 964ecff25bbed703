"""A few launches of the packed STFT kernel at the 48 kHz geometries, 512 utterances x 1 s (for counter collection / traces)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
x = torch.randn(512, 48000, device="cuda")
with torch.no_grad():
    for fl, fp, nfft in ((1200, 240, 2048), (800, 200, 1024)):
        st = dsp.STFT(fl, fp, nfft, device="cuda")
        for _ in range(8):
            y = st(x)
torch.cuda.synchronize()
