import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import diffsptk_amd as dsp
from diffsptk_amd import _lib, functional as F
x = torch.randn(3, 2500, generator=torch.Generator().manual_seed(5)).to("cuda")
for L, P in ((400, 80), (512, 64), (256, 100), (320, 40), (12, 4), (16, 8), (64, 16)):
    y = F.stft(x, frame_length=L, frame_period=P, fft_length=512)
    name = _lib.last_kernel()
    ref = F.stft(x.double(), frame_length=L, frame_period=P, fft_length=512)
    err = (y.double() - ref).abs() / ref.amax(-1, keepdim=True)
    bad = err > 1e-5
    print(L, P, name, "bad frac %.4f" % bad.float().mean().item(), "bad frames utt0:", bad[0].any(-1).nonzero().flatten().tolist()[:12],
          "bins:", bad[0].any(0).nonzero().flatten().tolist()[:12])
xx = torch.randn(1024, 16000, generator=torch.Generator().manual_seed(1)).to("cuda")
m = dsp.STFT(400, 80, 512).to("cuda")
y0 = m(xx).clone()
nd = 0
for i in range(10):
    y1 = m(xx)
    nd += int((y1 != y0).sum().item())
print("non-deterministic elements over 10 reruns:", nd)
