"""Timing of mgc2sp and the frequency-domain MLSA filter at the bench size of the f rows (256 x 1 s)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
from diffsptk_amd import _lib
dev = "cuda"
g = torch.Generator().manual_seed(0)
mc = (0.1 * torch.randn(256, 200, 25, generator=g)).to(dev)
x = torch.randn(256, 16000, generator=g).to(dev)
def gpu_time(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    m = dsp.MelGeneralizedCepstrumToSpectrum(24, 512, alpha=0.42, device=dev)
    print("mgc2sp power ms", round(gpu_time(lambda: m(mc)), 4), _lib.last_kernel())
    mc_ = dsp.MelGeneralizedCepstrumToSpectrum(24, 512, alpha=0.42, out_format="complex", device=dev)
    print("mgc2sp complex ms", round(gpu_time(lambda: mc_(mc)), 4))
    ml = dsp.MLSA(24, 80, alpha=0.42, mode="freq-domain", frame_length=400, fft_length=512, device=dev)
    print("MLSA freq-domain ms", round(gpu_time(lambda: ml(x, mc)), 4))
