import os, sys, torch
sys.path.insert(0, "/root/repo")
import diffsptk_amd as dsp
xx = torch.randn(1024, 16000, generator=torch.Generator().manual_seed(1)).to("cuda")
m = dsp.STFT(400, 80, 512).to("cuda")
ref = m(xx).clone()
neg = 0; diff = 0
for i in range(40):
    y = m(xx)
    neg += int((y < 0).sum().item())
    d = (y != ref)
    diff += int(d.sum().item())
    if d.any() and diff <= 64:
        idx = d.nonzero()[:4].tolist()
        print("run", i, [(u, f, k, ref[u, f, k].item(), y[u, f, k].item()) for u, f, k in idx])
print("ref negatives:", int((ref < 0).sum().item()), " negatives over 40 runs:", neg, " differing elements:", diff)
