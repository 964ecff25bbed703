"""A few launches of the one-launch LPC backward (dsa_frame_window_lpc_bwd) at the bench size (for counter collection)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
from diffsptk_amd import ops, _lib
dev = "cuda"
x = torch.randn(1024, 16000, device=dev)
w = dsp.Window(400, device=dev).window
g = torch.randn(1024, 200, 25, device=dev)
gx = torch.empty_like(x)
for _ in range(10):
    ops._call("dsa_frame_window_lpc_bwd", g.data_ptr(), x.data_ptr(), 1024, 16000, 400, 80, w.data_ptr(), 1, 0, 24, 1e-5, _lib.F32, gx.data_ptr(), ops._stream())
assert _lib.last_kernel() == "frame_window_lpc24_bwd_mfma", _lib.last_kernel()
torch.cuda.synchronize()
