"""A few forward + backward calls of the 48 kHz analysis (fft 2048 / order 49, F frames) for a kernel trace."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
F = int(os.environ.get("F", "102400"))
g = torch.Generator().manual_seed(0)
m = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=49, alpha=0.55, n_iter=10, device="cuda")
X = (torch.randn(F, 1025, generator=g).square() + 0.05).to("cuda").requires_grad_(True)
for _ in range(4):
    X.grad = None
    m(X).sum().backward()
torch.cuda.synchronize()
