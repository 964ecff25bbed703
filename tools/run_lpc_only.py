"""A dozen launches of the fused Frame -> Window -> LPC kernel (for rocprofv3 counter collection)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
from diffsptk_amd import ops
dev = "cuda"
x = torch.randn(1024, 16000, device=dev)
w = dsp.Window(400, device=dev).window
for _ in range(12):
    a = ops.frame_window_lpc(x, w, 400, 80, 24, 1e-5)
torch.cuda.synchronize()
