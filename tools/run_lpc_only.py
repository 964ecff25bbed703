"""A few launches of the fused Frame + Window + LPC kernel at the bench size (for counter collection)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
from diffsptk_amd import ops, _lib
dev = "cuda"
x = torch.randn(1024, 16000, device=dev)
w = dsp.Window(400, device=dev).window
with torch.no_grad():
    for _ in range(10):
        y = ops.frame_window_lpc(x, w, 400, 80, 24, 1e-5)
assert _lib.last_kernel() in ("frame_window_lpc24_fwd", "frame_window_lpc24_mfma_fwd"), _lib.last_kernel()
torch.cuda.synchronize()
