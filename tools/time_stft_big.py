"""STFT at the 48 kHz geometries, us per launch and fraction of the HBM peak (algorithmic bytes: P * 4 in, (nfft / 2 + 1) * 4 out per
frame); DSA_STFT_BIG=0 in the environment times the generic kernel."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd import _lib
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
out = []
for B in (64, 512):
    x = torch.randn(B, 48000, device=dev)
    for fl, fp, nfft in ((1200, 240, 2048), (800, 200, 1024), (1024, 256, 1024)):
        st = dsp.STFT(fl, fp, nfft, device=dev)
        with torch.no_grad():
            y = st(x); k = _lib.last_kernel()
            t = timeit(lambda: st(x))
        fr = y.shape[0] * y.shape[1]
        by = fr * (fp * 4 + (nfft // 2 + 1) * 4)
        out.append(f"B={B} {fl}/{fp}/{nfft} {k}: {t:.1f} us, {by / t / 1e6:.2f} TB/s = {by / t / 1e6 / 8:.3f} of HBM")
print("\n".join(out))
