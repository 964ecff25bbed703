import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
g = torch.Generator().manual_seed(0)
for F in (12800, 102400):
    X = (torch.randn(F, 1025, generator=g).square() + 0.05).cuda()
    m = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=49, alpha=0.55, n_iter=10, device="cuda")
    with torch.no_grad():
        y = m(X); y = m(X)
    names = ["top barrier + B operands + first staging", "stage loop", "(nothing)", "partial sums exchange", "records", "solve (call)"]
    for w in (0, 8):
        print(f"F={F} wave {'0 (even stages)' if w == 0 else '4 (odd stages)'}:", "  ".join(f"{names[i - 1]} {int(y[w, i])}" for i in range(1, 6)))
