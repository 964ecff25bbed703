"""Timing of the MLSA filter modes at the bench size of the f rows (256 utterances x 1 s), float32 against float64 on four utterances."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
from diffsptk_amd import _lib
dev = "cuda"
B, N, P, M = 256, 200, 80, 24
g = torch.Generator().manual_seed(0)
x = torch.randn(B, N * P, generator=g).to(dev)
stft = dsp.STFT(400, 80, 512).to(dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10).to(dev)
def gpu_time(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    mc = mcep(stft(torch.randn(B, N * P, generator=g).to(dev)))[:, :N]
    for mode in ("multi-stage", "single-stage"):
        ml = dsp.MLSA(M, P, alpha=0.42, mode=mode, device=dev)
        y = ml(x, mc)
        print(mode, "ms", round(gpu_time(lambda: ml(x, mc)), 3), _lib.last_kernel(), float(y.abs().max()))
        yd = dsp.MLSA(M, P, alpha=0.42, mode=mode, device=dev, dtype=torch.float64)(x[:4].double(), mc[:4].double())
        print("   max |f32 - f64| / max|y| on 4 utterances:", float((y[:4].double() - yd).abs().max() / yd.abs().max()))
