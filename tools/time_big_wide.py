"""One-launch 48 kHz analysis, wide tiles forced / planned, us per analysis (for A/B runs of library builds)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = "cuda"
def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
out = []
for nfft, M, fl, fp in ((2048, 49, 1200, 240), (1024, 34, 800, 200)):
    m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=dev)
    for B in (128, 512):
        x = torch.randn(B, 48000, generator=g).to(dev)
        with torch.no_grad():
            X = dsp.STFT(fl, fp, nfft, device=dev)(x)
            os.environ["DSA_MCEP_BIG_WIDE"] = "1"
            tw = timeit(lambda: m(X))
            os.environ.pop("DSA_MCEP_BIG_WIDE")
            tp = timeit(lambda: m(X))
        out.append(f"{nfft}/{M} {X.shape[0] * X.shape[1]}: wide {tw:.0f} planned {tp:.0f}")
print(" | ".join(out))
