"""The fused step alternating between two streams with DSA_ALGO_OVERLAPPED_LAUNCHES (for kernel traces: do consecutive launches overlap?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
from diffsptk_amd import ops
x = torch.randn(1024, 16000, device="cuda")
stft = dsp.STFT(400, 80, 512, device="cuda")
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device="cuda")
fused = dsp.fuse(stft, mcep)
ss = [torch.cuda.Stream(), torch.cuda.Stream()]
flag = os.environ.get("OVERLAP", "1") == "1"
with torch.no_grad():
    for it in range(40):
        with torch.cuda.stream(ss[it % 2]), ops.overlapped_launches(flag):
            y = fused(x)
torch.cuda.synchronize()
