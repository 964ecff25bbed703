"""A few launches of the fused STFT -> mel filter bank kernel at the bench size (for counter collection)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
x = torch.randn(1024, 16000, device=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, use_power=True, device=dev)
f = dsp.fuse(stft, fb)
with torch.no_grad():
    for _ in range(10):
        y = f(x)
assert f.last_path == "fused"
torch.cuda.synchronize()
