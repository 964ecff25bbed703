"""dsa_mcep_newton_steps (all Newton steps of the 48 kHz analysis in one persistent launch) against the two-launch step it replaces:
bit for bit at every batch size / order / iteration count, against float64 through the module, and the call times."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd import _lib
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
bad = 0
for M in (49, 34, 32, 46):
    for F in (1, 15, 64, 65, 3217, 12800):
        X = (torch.randn(F, 1025, generator=g).square() + 0.05).to(dev)
        for n_iter in (1, 2, 10):
            m = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=M, alpha=0.55, n_iter=n_iter, device=dev)
            with torch.no_grad():
                os.environ["DSA_MCEP_BIG"] = "0"; a = m(X); ka = _lib.last_kernel()
                os.environ["DSA_MCEP_BIG"] = "2"; b = m(X); kb = _lib.last_kernel()
                b2 = m(X)
            eq = torch.equal(b, b2) and torch.equal(a, b) and kb == "mcep_big_newton"   # same bits on both paths
            if not eq:
                bad += 1
                print(f"M={M} F={F} n_iter={n_iter}: {ka} vs {kb}: max |diff| {float((a - b).abs().max()):.3e} (finite {bool(torch.isfinite(b).all())}, repeat equal {torch.equal(b, b2)})")
print("mismatching cases:", bad)
X = (torch.randn(2000, 1025, generator=g).square() + 0.05).to(dev)
m32 = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=49, alpha=0.55, n_iter=10, device=dev)
m64 = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=49, alpha=0.55, n_iter=10, device=dev, dtype=torch.float64)
with torch.no_grad():
    y64 = m64(X.double())
    for flag in ("0", "2"):
        os.environ["DSA_MCEP_BIG"] = flag
        y = m32(X)
        print(f"DSA_MCEP_BIG={flag}: {_lib.last_kernel()} max |f32 - f64| = {float((y.double() - y64).abs().max()):.3e}")
os.environ["DSA_MCEP_BIG"] = "2"
m34 = dsp.MelCepstralAnalysis(fft_length=1024, cep_order=34, alpha=0.55, n_iter=10, device=dev)
for B in (64, 512):
    x = torch.randn(B, 48000, generator=g).to(dev)
    with torch.no_grad():
        X = dsp.STFT(800, 200, 1024, device=dev)(x)
        for flag in ("0", "2", "0", "2"):
            os.environ["DSA_MCEP_BIG"] = flag
            t = timeit(lambda: m34(X), 5)
            print(f"1024 / 34: B={B} ({X.shape[0] * X.shape[1]} frames) DSA_MCEP_BIG={flag}: {t:.1f} us per analysis ({_lib.last_kernel()})")
for B in (64, 512):
    x = torch.randn(B, 48000, generator=g).to(dev)
    with torch.no_grad():
        X = dsp.STFT(1200, 240, 2048, device=dev)(x)
        for flag in ("0", "2", "0", "2"):
            os.environ["DSA_MCEP_BIG"] = flag
            t = timeit(lambda: m32(X), 5)
            print(f"B={B} ({X.shape[0] * X.shape[1]} frames) DSA_MCEP_BIG={flag}: {t:.1f} us per analysis ({_lib.last_kernel()})")
