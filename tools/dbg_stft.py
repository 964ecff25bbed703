import os, sys, torch
sys.path.insert(0, "/root/repo")
import diffsptk_amd as dsp
x = torch.randn(2, 16000, generator=torch.Generator().manual_seed(0)).to("cuda")
y = dsp.STFT(400, 80, 512).to("cuda")(x)
ref = dsp.STFT(400, 80, 512).to("cuda").double()(x.double())
err = ((y.double() - ref).abs() / ref.amax(-1, keepdim=True))
bad = (err > 1e-5)
print("bad fraction", bad.float().mean().item())
print("bad frames (utt 0):", bad[0].any(-1).nonzero().flatten().tolist()[:40])
f = 50
print("bad bins frame", f, bad[0, f].nonzero().flatten().tolist())
print("bad bins frame 1", bad[0, 1].nonzero().flatten().tolist()[:80])
