#!/usr/bin/env python3
"""BASELINE configs[2]/[3]: STFT + mcep forward+backward and the LPC branch, timed with HIP events.
usage: python tools/bench_fwdbwd.py [B]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
from diffsptk_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = "cuda"
x = torch.randn(B, 16000, device=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)

def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

# clock ramp: the first ~20 ms after idle run 10 % slower (see bench.py)
_t0 = time.perf_counter()
while time.perf_counter() - _t0 < 0.3:
    with torch.no_grad():
        for _ in range(10): mcep(stft(x))
    torch.cuda.synchronize()

def fwd():
    with torch.no_grad(): return mcep(stft(x))
def fwdbwd():
    xg = x.clone().requires_grad_(True)
    mcep(stft(xg)).mean().backward()
    return xg.grad
frames = B * 200
t_f, t_fb = timeit(fwd), timeit(fwdbwd)
print(f"B={B}: stft+mcep fwd {t_f:.3f} ms ({frames/t_f*1e3:.3e} frames/s) | fwd+bwd {t_fb:.3f} ms ({frames/t_fb*1e3:.3e} frames/s)")
# pieces of the backward
xg = x.clone().requires_grad_(True)
X = stft(xg); Xd = X.detach().requires_grad_(True)
mc = mcep(Xd)
g = torch.ones_like(mc) / mc.numel()
t_mb = timeit(lambda: torch.autograd.grad(mc, Xd, g, retain_graph=True))
gX = torch.autograd.grad(mc, Xd, g, retain_graph=True)[0]
t_sb = timeit(lambda: torch.autograd.grad(X, xg, gX, retain_graph=True))
print(f"   mcep bwd {t_mb:.3f} ms | stft bwd {t_sb:.3f} ms")
w = dsp.Window(400, device=dev).window
t_l = timeit(lambda: ops.frame_window_lpc(x, w, 400, 80, 24, 1e-5))
fr, wn, lpc = dsp.Frame(400, 80), dsp.Window(400, device=dev), dsp.LPC(400, 24, eps=1e-5, device=dev)
t_lm = timeit(lambda: lpc(wn(fr(x))))
def lpc_fb():
    xg = x.clone().requires_grad_(True); lpc(wn(fr(xg))).mean().backward()
t_lfb = timeit(lpc_fb)
print(f"   LPC fused fwd {t_l:.3f} ms ({frames/t_l*1e3:.3e} frames/s) | module chain fwd {t_lm:.3f} ms | fwd+bwd {t_lfb:.3f} ms")
# filter bank / MFCC on the STFT power spectrum
fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, use_power=True, device=dev)
mf = dsp.MFCC(fft_length=512, mfcc_order=12, n_channel=40, sample_rate=16000, lifter=22, device=dev)
with torch.no_grad():
    Xs = stft(x)
    t_fb = timeit(lambda: fb(Xs))
    t_mf = timeit(lambda: mf(Xs))
    t_sfb = timeit(lambda: mf(stft(x)))
Xr = Xs.clone().requires_grad_(True)
t_mfb = timeit(lambda: torch.autograd.grad(mf(Xr).sum(), Xr))
print(f"   fbank(40) fwd {t_fb:.3f} ms ({1188*frames/t_fb/1e6:.0f} GB/s) | MFCC(12) fwd {t_mf:.3f} ms | STFT->MFCC {t_sfb:.3f} ms "
      f"({frames/t_sfb*1e3:.3e} frames/s) | MFCC fwd+bwd {t_mfb:.3f} ms")
# analysis -> synthesis round trip
stc = dsp.STFT(400, 80, 512, out_format="complex", device=dev)
ist = dsp.ISTFT(400, 80, 512, device=dev)
with torch.no_grad():
    Z = stc(x)
    t_sc = timeit(lambda: stc(x))
    t_is = timeit(lambda: ist(Z))
    err = (ist(Z) - x).abs().max().item()
print(f"   STFT complex fwd {t_sc:.3f} ms | ISTFT {t_is:.3f} ms ({frames/t_is*1e3:.3e} frames/s) | round-trip max error {err:.2e}")
gl = dsp.GriffinLim(400, 80, 512, n_iter=8, init_phase="zeros", device=dev)
with torch.no_grad():
    Xp = stft(x)
    t_gl = timeit(lambda: gl(Xp, out_length=x.size(-1)))
print(f"   Griffin-Lim 8 iterations {t_gl:.3f} ms ({t_gl/8:.3f} ms per iteration: ISTFT + complex STFT + update)")
# cepstral analysis of the STFT output
fc0 = dsp.CepstralAnalysis(fft_length=512, cep_order=24, device=dev)
fc2 = dsp.CepstralAnalysis(fft_length=512, cep_order=24, n_iter=2, device=dev)
with torch.no_grad():
    t_fc0 = timeit(lambda: fc0(Xp))
    t_fc2 = timeit(lambda: fc2(Xp))
Xq = Xp.clone().requires_grad_(True)
t_fcb = timeit(lambda: torch.autograd.grad(fc0(Xq).sum(), Xq))
print(f"   fftcep(24) fwd {t_fc0:.3f} ms ({frames/t_fc0*1e3:.3e} frames/s) | n_iter=2 fwd {t_fc2:.3f} ms | n_iter=0 fwd+bwd {t_fcb:.3f} ms")
Zg = Z.clone().requires_grad_(True)
t_isb = timeit(lambda: torch.autograd.grad(ist(Zg).sum(), Zg))
print(f"   ISTFT fwd+bwd {t_isb:.3f} ms")
