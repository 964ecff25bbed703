"""The packed STFT backward for fft_length 1024 / 2048 (csrc/stft_bwd_pk_big.h) against float64 autograd of torch.stft-style math (the ATen port) and,
run with DSA_STFT_BIG_BWD=0, the generic backward; times.  usage: python tools/check_stft_big_bwd.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd import _lib
dev = "cuda"
g = torch.Generator().manual_seed(2)
def ref_grad(x, w64, fl, fp, nfft, gy):
    # float64: frames (center, constant padding), window, rfft, power
    x64 = x.double().cpu().requires_grad_(True)
    left = fl // 2
    T = x64.size(-1)
    N = (T - 1) // fp + 1
    xp = torch.nn.functional.pad(x64, (left, (N - 1) * fp + fl - left - T if (N - 1) * fp + fl - left > T else 0))
    fr = xp.unfold(-1, fl, fp)[:, :N]
    X = torch.fft.rfft(fr * w64, n=nfft)
    y = X.real ** 2 + X.imag ** 2
    (y * gy.double().cpu()).sum().backward()
    return y.detach(), x64.grad
for (fl, fp, nfft) in ((1200, 240, 2048), (800, 200, 1024), (1024, 256, 1024), (2048, 512, 2048), (882, 220, 1024), (600, 150, 2048)):
    for (B, T) in ((3, 48000), (2, 4410), (5, 2000)):
        x = torch.randn(B, T, generator=g).to(dev)
        stft = dsp.STFT(fl, fp, nfft, device=dev)
        xg = x.clone().requires_grad_(True)
        y = stft(xg)
        gy = torch.randn(y.shape, generator=g).to(dev)
        (y * gy).sum().backward()
        kern = _lib.last_kernel()
        yr, gr = ref_grad(x, stft.window.double().cpu(), fl, fp, nfft, gy)
        err = float((xg.grad.double().cpu() - gr).abs().max() / gr.abs().max())
        print(f"fl {fl} fp {fp} nfft {nfft} B {B} T {T}: kernel {kern}, max |gx - gx64| / max |gx64| = {err:.2e}, finite {bool(torch.isfinite(xg.grad).all())}")
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (fl, fp, nfft) in ((1200, 240, 2048), (800, 200, 1024)):
    for B in (64, 512):
        x = torch.randn(B, 48000, generator=g).to(dev)
        stft = dsp.STFT(fl, fp, nfft, device=dev)
        xg = x.clone().requires_grad_(True)
        y = stft(xg)
        gy = torch.randn_like(y)
        t = timeit(lambda: torch.autograd.grad(y, xg, gy, retain_graph=True))
        print(f"fl {fl} fp {fp} nfft {nfft} B {B}: backward {t:.3f} ms ({_lib.last_kernel()})")
