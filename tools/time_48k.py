"""The analysis at a 48 kHz set-up (frame 1200 / period 240 / fft 2048 / order 49 / alpha 0.55): which kernels run and how long,
64 utterances x 1 s (12 800 frames), float32."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
from diffsptk_amd import _lib
dev = "cuda"
def gpu_time(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.randn(64, 48000, generator=torch.Generator().manual_seed(0)).to(dev)
for (fl, fp, nfft, M, a) in ((1200, 240, 2048, 49, 0.55), (1024, 256, 1024, 34, 0.55), (400, 80, 512, 24, 0.42)):
    stft = dsp.STFT(fl, fp, nfft, device=dev)
    mcep = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=a, n_iter=10, device=dev)
    with torch.no_grad():
        X = stft(x); k1 = _lib.last_kernel()
        t1 = gpu_time(lambda: stft(x))
        mc = mcep(X); k2 = _lib.last_kernel()
        t2 = gpu_time(lambda: mcep(X))
        os.environ["DSA_MCEP_COMPOSED"] = "0"
        mc_g = mcep(X); k3 = _lib.last_kernel()
        t3 = gpu_time(lambda: mcep(X), 1)
        os.environ["DSA_MCEP_COMPOSED"] = "1"
        ref = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=a, n_iter=10, device=dev, dtype=torch.float64)(X[:4].double())
        err = float((mc[:4].double() - ref).abs().max()), float((mc_g[:4].double() - ref).abs().max())
    fr = X.shape[0] * X.shape[1]
    print(f"fl {fl} fp {fp} nfft {nfft} M {M}: {fr} frames  STFT {t1:.3f} ms ({k1})  mcep {t2:.3f} ms ({k2}; one-workgroup-per-frame kernel {t3:.3f} ms)  -> {fr / (t1 + t2) / 1e3:.2f} Mframes/s; max |f32 - f64| {err[0]:.2e} (generic kernel {err[1]:.2e})")
# with a gradient: whole-batch composition + autograd against the generic kernel pair
for (fl, fp, nfft, M, a) in ((1200, 240, 2048, 49, 0.55), (1024, 256, 1024, 34, 0.55)):
    stft = dsp.STFT(fl, fp, nfft, device=dev)
    mcep = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=a, n_iter=10, device=dev)
    with torch.no_grad():
        X = stft(x)
    def fb():
        Xg = X.clone().requires_grad_(True)
        mcep(Xg).sum().backward()
    t_c = gpu_time(fb, 2)
    os.environ["DSA_MCEP_COMPOSED"] = "0"
    t_g = gpu_time(fb, 1)
    os.environ["DSA_MCEP_COMPOSED"] = "1"
    print(f"nfft {nfft} M {M}: forward + backward {t_c:.2f} ms (generic kernel pair {t_g:.2f} ms)")
# the mel-generalized analysis and mgc2sp at the 48 kHz set-up (row products of 1025 bins)
nfft, M, a = 2048, 49, 0.55
with torch.no_grad():
    X = dsp.STFT(1200, 240, nfft, device=dev)(x)
    for name in ("1", "0"):
        os.environ["DSA_FREQT_GEMM"] = name
        mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=a, gamma=-0.5, n_iter=5, device=dev)
        t = gpu_time(lambda: mg(X), 1)
        mc = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=a, n_iter=10, device=dev)(X)
        m2s = dsp.MelGeneralizedCepstrumToSpectrum(M, nfft, alpha=a, n_fft=4096, device=dev)
        t2 = gpu_time(lambda: m2s(mc), 2)
        print(f"DSA_FREQT_GEMM={name}: mgcep (gamma -0.5, 5 steps) {t:.2f} ms, mgc2sp {t2:.3f} ms")
os.environ["DSA_FREQT_GEMM"] = "1"
