"""Griffin-Lim (20 iterations, zero initial phase) at 256 utterances x 1 s (for traces)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
x = torch.randn(256, 16000, generator=torch.Generator().manual_seed(0)).to(dev)
with torch.no_grad():
    X = dsp.STFT(400, 80, 512, device=dev)(x)
    gl = dsp.GriffinLim(400, 80, 512, n_iter=20, init_phase="zeros", device=dev)
    for _ in range(3):
        y = gl(X, out_length=16000)
torch.cuda.synchronize()
