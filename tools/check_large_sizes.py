"""Sizes beyond 2^31 ELEMENTS per tensor ("bounded by the 32-bit grid, not memory" -- checked here, on the GPU box).

Every op of the path is per utterance, and the kernels are batch-invariant bit for bit, so the check is exact: a few utterances of a
batch whose spectrogram has more than 2^31 floats (first, last, the two around the 2^31-element boundary) must equal, BITWISE, the
same utterances computed alone in a small batch.  Forward of STFT (packed 512 kernel), mel-cepstral analysis (two kernels and the
one-launch fuse(stft, mcep)), fuse(frame, window, lpc); the gradient of mcep(stft(x)); the 48 kHz set-up (fft 2048 / order 49).

    python tools/check_large_sizes.py [utterances_16k (default 42000)] [utterances_48k (default 10600)]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp  # noqa: E402

dev = torch.device("cuda", 0)
B16, B48 = 42000, 10600
bad = 0


def pick(B, N, K):
    """utterance indices: ends + the pair around the 2^31-element boundary of a (B, N, K) tensor"""
    edge = (2 ** 31) // (N * K)
    idx = sorted({0, 1, B // 2, B - 2, B - 1} | ({edge - 1, edge, edge + 1} if 1 <= edge < B - 1 else set()))
    return idx, edge


def same(tag, big, small):
    global bad
    eq = torch.equal(big, small)
    fin = bool(torch.isfinite(big).all())
    print(f"  {tag}: bitwise equal to the small batch: {eq}, finite: {fin}", flush=True)
    bad += (not eq) or (not fin)


def section_16k():
    FL, FP, NFFT, M = 400, 80, 512, 24
    T = 16000
    N = (T - 1) // FP + 1
    idx, edge = pick(B16, N, NFFT // 2 + 1)
    print(f"16 kHz: {B16} utterances x 1 s = {B16 * N} frames; spectrogram {B16 * N * 257 / 2 ** 31:.3f} x 2^31 floats "
          f"({B16 * N * 257 * 4 / 2 ** 30:.2f} GiB); boundary inside utterance {edge}; checked {idx}", flush=True)
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(B16, T, device=dev, generator=g)
    xs = x[idx].clone()
    stft = dsp.STFT(FL, FP, NFFT, device=dev)
    mcep = dsp.MelCepstralAnalysis(fft_length=NFFT, cep_order=M, alpha=0.42, n_iter=10, device=dev)
    with torch.no_grad():
        t0 = time.perf_counter()
        X = stft(x)
        torch.cuda.synchronize()
        print(f"  stft: {tuple(X.shape)} in {1e3 * (time.perf_counter() - t0):.1f} ms ({dsp._lib.last_kernel()})", flush=True)
        same("STFT power spectrogram", X[idx], stft(xs))
        mc = mcep(X)
        torch.cuda.synchronize()
        print(f"  mcep: {tuple(mc.shape)} ({dsp._lib.last_kernel()})", flush=True)
        mcs = mcep(stft(xs))
        same("mcep(stft(x))", mc[idx], mcs)
        del X
        fused = dsp.fuse(stft, mcep)
        mcf = fused(x)
        torch.cuda.synchronize()
        print(f"  fuse(stft, mcep): ({dsp._lib.last_kernel()})", flush=True)
        same("fuse(stft, mcep)(x)", mcf[idx], fused(xs))
        same("fuse(stft, mcep)(x) vs mcep(stft(x))", mcf[idx], mcs)
        del mc, mcf
        frame, window = dsp.Frame(FL, FP), dsp.Window(FL, device=dev)
        lpc = dsp.LPC(FL, M, device=dev)
        fl = dsp.fuse(frame, window, lpc)
        a = fl(x)
        torch.cuda.synchronize()
        same("fuse(frame, window, lpc)(x)", a[idx], fl(xs))
        del a
        # ONE utterance of B16 x T samples (8.4 M frames): the frames around the 2^31-element boundary of its spectrogram against a
        # 0.5 s cut of the waveform there (interior frames: the cut's zero padding reaches 200 samples in)
        xl = x.reshape(1, B16 * T)
        Nl = (B16 * T - 1) // FP + 1
        m0 = ((2 ** 31) // 257 - 40) // 16 * 16
        cut = xl[:, m0 * FP: m0 * FP + 8000].clone()
        Xl = stft(xl)
        torch.cuda.synchronize()
        print(f"  one utterance of {B16 * T} samples: stft {tuple(Xl.shape)} ({dsp._lib.last_kernel()})", flush=True)
        Xc = stft(cut)
        same("STFT of the long utterance, frames around the boundary", Xl[0, m0 + 3: m0 + 96], Xc[0, 3:96])
        same("STFT of the long utterance, last frames", Xl[0, Nl - 64:], stft(xl[:, (Nl - 96) * FP:].clone())[0, 32:])
        del Xl
        ml = fused(xl)
        torch.cuda.synchronize()
        same("fuse(stft, mcep) of the long utterance", ml[0, m0 + 3: m0 + 96], fused(cut)[0, 3:96])
        al = fl(xl)
        same("fuse(frame, window, lpc) of the long utterance", al[0, m0 + 3: m0 + 96], fl(cut)[0, 3:96])
        del ml, al, xl
    # gradient of mcep(stft(x)) at the large size: a fixed random cotangent per utterance
    gsel = torch.randn(len(idx), N, M + 1, device=dev, generator=g)
    xg = x.requires_grad_(True)
    mc = mcep(stft(xg))
    w = torch.zeros(B16, N, M + 1, device=dev)
    w[idx] = gsel
    (mc * w).sum().backward()
    torch.cuda.synchronize()
    gx_big = xg.grad[idx].clone()
    kb = dsp._lib.last_kernel()
    print(f"  backward of mcep(stft(x)) done ({kb}); cotangent non-zero on the checked utterances only", flush=True)
    rest = xg.grad.clone()
    rest[idx] = 0
    others = float(rest.abs().max())
    del rest
    xsg = xs.clone().requires_grad_(True)
    (mcep(stft(xsg)) * gsel).sum().backward()
    # above ops.MCEP_HIST_RT_MAX_BYTES the forward keeps no rt rows and the backward recomputes them (another kernel, other rounding):
    # bitwise equality is then not expected; the error against the small batch's gradient is what counts (tests: 3e-6 of the row maximum)
    eq = torch.equal(gx_big, xsg.grad)
    err = float((gx_big - xsg.grad).abs().max() / xsg.grad.abs().max())
    print(f"  d mcep(stft(x)) / dx: bitwise equal to the small batch: {eq}; max |difference| / max |gradient| = {err:.3e}; "
          f"finite: {bool(torch.isfinite(gx_big).all())}", flush=True)
    print(f"  largest gradient magnitude on the other utterances (must be 0): {others:.3e}", flush=True)
    global bad
    bad += (err > 3e-6) + (others != 0.0)


def section_48k():
    FL, FP, NFFT, M = 1200, 240, 2048, 49
    T = 48000
    N = (T - 1) // FP + 1
    idx, edge = pick(B48, N, NFFT // 2 + 1)
    print(f"48 kHz: {B48} utterances x 1 s = {B48 * N} frames; spectrogram {B48 * N * 1025 / 2 ** 31:.3f} x 2^31 floats "
          f"({B48 * N * 1025 * 4 / 2 ** 30:.2f} GiB); boundary inside utterance {edge}; checked {idx}", flush=True)
    g = torch.Generator(device=dev).manual_seed(8)
    x = torch.randn(B48, T, device=dev, generator=g)
    xs = x[idx].clone()
    stft = dsp.STFT(FL, FP, NFFT, device=dev)
    mcep = dsp.MelCepstralAnalysis(fft_length=NFFT, cep_order=M, alpha=0.55, n_iter=10, device=dev)
    with torch.no_grad():
        X = stft(x)
        torch.cuda.synchronize()
        print(f"  stft: {tuple(X.shape)} ({dsp._lib.last_kernel()})", flush=True)
        Xs = stft(xs)
        same("STFT 2048 power spectrogram", X[idx], Xs)
        t0 = time.perf_counter()
        mc = mcep(X)
        torch.cuda.synchronize()
        print(f"  mcep 2048 / 49: {tuple(mc.shape)} in {1e3 * (time.perf_counter() - t0):.1f} ms ({dsp._lib.last_kernel()})", flush=True)
        same("mcep(stft(x)) at 2048 / 49", mc[idx], mcep(Xs))
        del X, mc
    # the 48 kHz gradient (one autograd node per analysis: residual-adjoint sweep + the glogx pass) beyond 2^31 elements of (F, K)
    gsel = torch.randn(len(idx), N, M + 1, device=dev, generator=g)
    xg = x.requires_grad_(True)
    mcg = mcep(stft(xg))
    w = torch.zeros(B48, N, M + 1, device=dev)
    w[idx] = gsel
    (mcg * w).sum().backward()
    torch.cuda.synchronize()
    gx_big = xg.grad[idx].clone()
    rest = xg.grad.clone()
    rest[idx] = 0
    others = float(rest.abs().max())
    del rest, mcg, w
    xsg = xs.clone().requires_grad_(True)
    (mcep(stft(xsg)) * gsel).sum().backward()
    eq = torch.equal(gx_big, xsg.grad)
    err = float((gx_big - xsg.grad).abs().max() / xsg.grad.abs().max())
    print(f"  d mcep(stft(x)) / dx at 2048 / 49: bitwise equal to the small batch: {eq}; max |difference| / max |gradient| = {err:.3e}; "
          f"finite: {bool(torch.isfinite(gx_big).all())}; peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB", flush=True)
    print(f"  largest gradient magnitude on the other utterances (must be 0): {others:.3e}", flush=True)
    global bad
    bad += (err > 3e-6) + (others != 0.0)


def run_all(b16=42000, b48=10600):
    """both sections; returns the number of mismatching / failing checks (tests/test_gpu_edge.py calls this)"""
    global B16, B48, bad
    B16, B48, bad = b16, b48, 0
    for sec in (section_16k, section_48k):
        try:
            sec()
        except Exception as e:   # noqa: BLE001 -- a report, not a test: say what failed and go on
            bad += 1
            print(f"  FAILED with {type(e).__name__}: {e}", flush=True)
        torch.cuda.empty_cache()
    print("mismatching / failing checks:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run_all(*(int(a) for a in sys.argv[1:3])) else 0)
