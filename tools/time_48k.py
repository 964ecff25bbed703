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
