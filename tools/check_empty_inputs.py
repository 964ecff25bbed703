"""Empty batches and shortest inputs through the module API (the reference's ATen ops accept zero-size leading dims and T >= 1):
every call must return the reference's shape, launch nothing out of bounds, and give an (empty / finite) gradient.

    python tools/check_empty_inputs.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp  # noqa: E402

dev = torch.device("cuda", 0)


def run_all():
    bad = 0

    def check(tag, fn, shape):
        nonlocal bad
        try:
            y = fn()
            torch.cuda.synchronize()
            ok = tuple(y.shape) == tuple(shape) and bool(torch.isfinite(y).all())
            print(f"  {tag}: shape {tuple(y.shape)} (expected {tuple(shape)}) {'ok' if ok else 'MISMATCH'}", flush=True)
            bad += not ok
            return y
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"  {tag}: FAILED with {type(e).__name__}: {e}", flush=True)
            return None

    for (FL, FP, NFFT, M, alpha, T) in ((400, 80, 512, 24, 0.42, 16000), (1200, 240, 2048, 49, 0.55, 48000)):
        N = (T - 1) // FP + 1
        K = NFFT // 2 + 1
        print(f"fft {NFFT} / order {M}:", flush=True)
        stft = dsp.STFT(FL, FP, NFFT, device=dev)
        mcep = dsp.MelCepstralAnalysis(fft_length=NFFT, cep_order=M, alpha=alpha, n_iter=10, device=dev)
        fused = dsp.fuse(stft, mcep)
        for B in (0,):
            x = torch.zeros(B, T, device=dev, requires_grad=True)
            check(f"stft, batch {B}", lambda: stft(x), (B, N, K))
            check(f"mcep(stft), batch {B}", lambda: mcep(stft(x)), (B, N, M + 1))
            check(f"fuse(stft, mcep), batch {B}", lambda: fused(x), (B, N, M + 1))
            y = check(f"mcep on an empty spectrogram", lambda: mcep(torch.ones(B, N, K, device=dev)), (B, N, M + 1))
            try:
                mcep(stft(x)).sum().backward()
                fused(x).sum().backward()
                torch.cuda.synchronize()
                okg = x.grad is not None and tuple(x.grad.shape) == (B, T)
                print(f"  backward, batch {B}: grad shape {None if x.grad is None else tuple(x.grad.shape)} {'ok' if okg else 'MISMATCH'}", flush=True)
                bad += not okg
            except Exception as e:   # noqa: BLE001
                bad += 1
                print(f"  backward, batch {B}: FAILED with {type(e).__name__}: {e}", flush=True)
        # shortest inputs: one sample, fewer samples than a frame, one more than a period
        for Ts in (1, FP, FP + 1, FL - 1):
            Ns = (Ts - 1) // FP + 1
            xs = torch.randn(3, Ts, device=dev, requires_grad=True)
            check(f"stft, T = {Ts}", lambda: stft(xs), (3, Ns, K))
            check(f"mcep(stft), T = {Ts}", lambda: mcep(stft(xs)), (3, Ns, M + 1))
            ya = check(f"fuse(stft, mcep), T = {Ts}", lambda: fused(xs), (3, Ns, M + 1))
            yb = mcep(stft(xs))
            if ya is not None and not torch.equal(ya, yb):
                bad += 1
                print(f"  fuse(stft, mcep) != mcep(stft), T = {Ts}: max diff {float((ya - yb).abs().max()):.3e}", flush=True)
            try:
                fused(xs).sum().backward()
                torch.cuda.synchronize()
                okg = bool(torch.isfinite(xs.grad).all())
                bad += not okg
                print(f"  backward, T = {Ts}: finite {okg}", flush=True)
            except Exception as e:   # noqa: BLE001
                bad += 1
                print(f"  backward, T = {Ts}: FAILED with {type(e).__name__}: {e}", flush=True)
    print("16 kHz consumers:", flush=True)
    stft = dsp.STFT(400, 80, 512, device=dev)
    x0 = torch.zeros(0, 16000, device=dev)
    fbank = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, device=dev)
    check("fbank(stft), batch 0", lambda: fbank(stft(x0)), (0, 200, 40))
    check("fuse(stft, fbank), batch 0", lambda: dsp.fuse(stft, fbank)(x0), (0, 200, 40))
    frame, window, lpc = dsp.Frame(400, 80), dsp.Window(400, device=dev), dsp.LPC(400, 24, device=dev)
    check("lpc(window(frame)), batch 0", lambda: lpc(window(frame(x0))), (0, 200, 25))
    check("fuse(frame, window, lpc), batch 0", lambda: dsp.fuse(frame, window, lpc)(x0), (0, 200, 25))
    istft = dsp.ISTFT(400, 80, 512, device=dev)
    stc = dsp.STFT(400, 80, 512, out_format="complex", device=dev)
    check("istft(stft complex), batch 0", lambda: istft(stc(x0), out_length=16000), (0, 16000))
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=5, device=dev)
    check("mgcep, batch 0", lambda: mg(stft(x0)), (0, 200, 25))
    fc = dsp.CepstralAnalysis(fft_length=512, cep_order=24, n_iter=2, device=dev)
    check("fftcep, batch 0", lambda: fc(stft(x0)), (0, 200, 25))
    # gradients of the consumers at batch 0 (empty gradients of the right shape), the synthesis side
    def check_grad(tag, fn):
        nonlocal bad
        try:
            xz = torch.zeros(0, 16000, device=dev, requires_grad=True)
            fn(xz).sum().backward()
            torch.cuda.synchronize()
            ok = xz.grad is not None and tuple(xz.grad.shape) == (0, 16000)
            print(f"  backward of {tag}, batch 0: {'ok' if ok else 'MISMATCH'}", flush=True)
            bad += not ok
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"  backward of {tag}, batch 0: FAILED with {type(e).__name__}: {e}", flush=True)

    check_grad("stft", stft)
    check_grad("fbank(stft)", lambda z: fbank(stft(z)))
    check_grad("fuse(stft, fbank)", dsp.fuse(stft, fbank))
    mfcc = dsp.MFCC(fft_length=512, mfcc_order=12, n_channel=40, sample_rate=16000, device=dev)
    check("mfcc(stft), batch 0", lambda: mfcc(stft(x0)), (0, 200, 12))
    check_grad("fuse(stft, mfcc)", dsp.fuse(stft, mfcc))
    check_grad("lpc(window(frame))", lambda z: lpc(window(frame(z))))
    check_grad("fuse(frame, window, lpc)", dsp.fuse(frame, window, lpc))
    check_grad("istft(stft complex)", lambda z: istft(stc(z), out_length=16000))
    check_grad("mgcep(stft)", lambda z: mg(stft(z)))
    check_grad("fftcep(stft)", lambda z: fc(stft(z)))
    for opts in (dict(zmean=True), dict(mode="reflect"), dict(relative_floor=-80), dict(out_format="db"), dict(out_format="magnitude")):
        so = dsp.STFT(400, 80, 512, device=dev, **opts)
        check(f"stft {opts}, batch 0", lambda: so(x0), (0, 200, 257))
        check_grad(f"stft {opts}", so)
    mcep16 = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
    so = dsp.STFT(400, 80, 512, device=dev, zmean=True, mode="reflect")
    check("fuse(stft zmean reflect, mcep), batch 0", lambda: dsp.fuse(so, mcep16)(x0), (0, 200, 25))
    mc0 = torch.zeros(0, 200, 25, device=dev)
    sp = dsp.MelGeneralizedCepstrumToSpectrum(24, 512, alpha=0.42, device=dev)
    check("mgc2sp, batch 0", lambda: sp(mc0), (0, 200, 257))
    m2b = dsp.MelCepstrumToMLSADigitalFilterCoefficients(24, alpha=0.42, device=dev)
    check("mc2b, batch 0", lambda: m2b(mc0), (0, 200, 25))
    for mode, kw in (("multi-stage", dict(taylor_order=20, cep_order=199)), ("single-stage", dict(ir_length=400, n_fft=512)),
                     ("freq-domain", dict(frame_length=400, fft_length=512))):
        try:
            ml = dsp.PseudoMGLSADigitalFilter(24, frame_period=80, alpha=0.42, mode=mode, device=dev, **kw)
            check(f"MLSA {mode}, batch 0", lambda: ml(x0, mc0), (0, 16000))
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"  MLSA {mode}: FAILED with {type(e).__name__}: {e}", flush=True)
    try:
        gl = dsp.GriffinLim(400, 80, 512, n_iter=2, device=dev)
        check("griffin-lim, batch 0", lambda: gl(torch.zeros(0, 200, 257, device=dev), out_length=16000), (0, 16000))
    except Exception as e:   # noqa: BLE001
        bad += 1
        print(f"  griffin-lim: FAILED with {type(e).__name__}: {e}", flush=True)
    # 48 kHz gradient at batch 0
    st48 = dsp.STFT(1200, 240, 2048, device=dev)
    mc48 = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=49, alpha=0.55, n_iter=10, device=dev)
    try:
        xz = torch.zeros(0, 48000, device=dev, requires_grad=True)
        mc48(st48(xz)).sum().backward()
        torch.cuda.synchronize()
        ok = xz.grad is not None and tuple(xz.grad.shape) == (0, 48000)
        print(f"  backward of mcep(stft) at 2048 / 49, batch 0: {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    except Exception as e:   # noqa: BLE001
        bad += 1
        print(f"  backward of mcep(stft) at 2048 / 49, batch 0: FAILED with {type(e).__name__}: {e}", flush=True)
    print("mismatching / failing checks:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run_all() else 0)
