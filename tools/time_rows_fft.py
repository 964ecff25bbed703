"""Timings of the rows that run on the generic LDS FFT (51 200 frames): mgc2sp, MLSA single-stage / freq-domain, fftcep."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = "cuda"
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
x = torch.randn(256, 16000, device=dev)
mc = 0.1 * torch.randn(256, 200, 25, device=dev)
with torch.no_grad():
    t1 = timeit(lambda: dsp.MelGeneralizedCepstrumToSpectrum(24, 512, alpha=0.42, device=dev)(mc))
    f1 = dsp.PseudoMGLSADigitalFilter(24, 80, alpha=0.42, mode="single-stage", device=dev)
    t2 = timeit(lambda: f1(x, mc))
    st = dsp.STFT(1024, 256, 1024, device=dev)
    t3 = timeit(lambda: st(x))
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: mgc2sp {t1:.3f} ms | MLSA single-stage {t2:.2f} ms | STFT 1024/256 {t3:.3f} ms")
