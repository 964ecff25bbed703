"""A/B of the tuned STFT forward variants (dev tool): DSA_STFT_VARIANT=<0|1> python tools/ab_stft.py [B].
Kernel time at B utterances x 1 s and deviation from the float64 generic path (parity tolerance of tests/)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp  # noqa: E402
from diffsptk_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.randn(B, 16000, generator=torch.Generator().manual_seed(0)).to("cuda")
x[1] *= 1e-6
x[2] *= 1e5
x[3, 8000:] *= 1e-4
stft = dsp.STFT(400, 80, 512).to("cuda")
y = stft(x)
torch.cuda.synchronize()
name = _lib.last_kernel()
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = stft(x)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
nb = min(B, 16)
ref = dsp.STFT(400, 80, 512).to("cuda").double()(x[:nb].double())
err = (y[:nb].double() - ref).abs()
tol = 1e-4 * ref + 2e-6 * ref.amax(-1, keepdim=True)
frames = y.shape[0] * y.shape[1]
print(f"{name}: median {ts[len(ts)//2]*1e3:.1f} us  min {ts[0]*1e3:.1f} us  {1348*frames/ts[len(ts)//2]/1e6:.0f} GB/s  "
      f"max err/tol {(err/tol).max().item():.4f}  max err/rowmax {(err/ref.amax(-1, keepdim=True)).max().item():.2e}  nan {torch.isnan(y).any().item()}")
