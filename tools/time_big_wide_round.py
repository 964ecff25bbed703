"""One round of the 48 kHz one-launch Newton kernel (32 768 frames: wide tiles; 16 384: narrow tiles) -- the line tools/ab_libs.sh prints per library
variant (DSA_BIG_ABL builds: 1 no solve, 3 no solve and no products)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = "cuda"
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
os.environ["DSA_MCEP_BIG_TWIN"] = "0"
out = []
for nfft, M in ((2048, 49), (1024, 34)):
    m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=dev)
    for F, wide in ((16384, "0"), (32768, "1")):
        os.environ["DSA_MCEP_BIG_WIDE"] = wide
        X = (torch.randn(F, nfft // 2 + 1, generator=g).square() + 0.05).to(dev)
        with torch.no_grad():
            out.append(f"{nfft}/{M} {'wide' if wide == '1' else 'narrow'} {F}: {timeit(lambda: m(X)):.0f} us")
print(" | ".join(out))
