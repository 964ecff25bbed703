"""Minimal script for traces / counter passes of the STFT backward and the inverse STFT (204 800 frames per launch)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = "cuda"
x = torch.randn(1024, 16000, device=dev)
st = dsp.STFT(400, 80, 512, device=dev)
xg = x.clone().requires_grad_(True)
y = st(xg); g = torch.randn_like(y)
stc = dsp.STFT(400, 80, 512, out_format="complex", device=dev); ist = dsp.ISTFT(400, 80, 512, device=dev)
with torch.no_grad():
    Z = stc(x)
x2 = x[:256].clone().requires_grad_(True); y2 = st(x2); g2 = torch.randn_like(y2)
for _ in range(int(os.environ.get("N", "20"))):
    torch.autograd.grad(y, xg, g, retain_graph=True)
    if os.environ.get("SMALL", "1") == "1":
        torch.autograd.grad(y2, x2, g2, retain_graph=True)
    with torch.no_grad():
        ist(Z)
torch.cuda.synchronize()
