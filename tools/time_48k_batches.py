"""MelCepstralAnalysis at the 48 kHz set-ups, forward, at 64 and 512 utterances x 1 s (ms per call and us per 1000 frames)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
def gpu_time(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = []
for B in (64, 512):
    x = torch.randn(B, 48000, device=dev)
    for (fl, fp, nfft, M, a) in ((1200, 240, 2048, 49, 0.55), (1024, 256, 1024, 34, 0.55)):
        with torch.no_grad():
            X = dsp.STFT(fl, fp, nfft, device=dev)(x)
            mcep = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=a, n_iter=10, device=dev)
            t = gpu_time(lambda: mcep(X))
        fr = X.shape[0] * X.shape[1]
        out.append(f"B{B} {nfft}/{M}: {t:.3f} ms ({t * 1e6 / fr:.1f} us/1000 frames)")
print(" | ".join(out))
