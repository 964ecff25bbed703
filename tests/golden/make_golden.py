#!/usr/bin/env python3
"""Generate golden fixtures by importing the REFERENCE (sp-nitech/diffsptk v4.0.0).

Runs ONLY in the build container, where the reference is mounted read-only at
/root/reference.  Nothing from the reference travels: this script writes plain
input/expected-output arrays (npz) that the parity tests load.  Two empty stub
modules (torchaudio, soundfile) are injected because the reference imports them at
module top although the hot path never touches them.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

torch 2.10.0+rocm7.0, CPU path (MKL / pocketfft / LAPACK), default 8 intra-op threads.
"""
import json
import os
import sys
import types
import wave

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    for name in ("torchaudio", "soundfile"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    import diffsptk  # noqa: E402

    return diffsptk


def read_wav_int16(path):
    with wave.open(path, "rb") as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2
        sr = w.getframerate()
        raw = w.readframes(w.getnframes())
    return np.frombuffer(raw, dtype="<i2").copy(), sr


def npy(t):
    return t.detach().cpu().numpy()


def main():
    torch.manual_seed(0)
    d = import_reference()
    F = d.functional
    f32, f64 = torch.float32, torch.float64
    DT = {"f32": f32, "f64": f64}

    # ------------------------------------------------------------------ tables
    tab = {}
    for name, dt in DT.items():
        win = d.Window(400, 512, dtype=dt).window
        tab[f"blackman400_power_{name}"] = npy(win)
    mc = d.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, dtype=f64)
    tab["freqt_A"] = npy(mc.freqt.A)
    tab["ifreqt_A"] = npy(mc.ifreqt.A)
    tab["rfreqt_A"] = npy(mc.rfreqt.A)
    tab["alpha_vector"] = npy(mc.alpha_vector)
    # the reference test grid for windows (tests/test_window.py): all types x norms x sym
    for w in list(range(7)) + ["povey", "sine", "vorbis", "kbd"]:
        for norm in (0, 1, 2):
            for sym in (True, False):
                if w == "kbd" and not sym:
                    continue
                for L in (8, 10, 400):
                    key = f"win_{w}_{norm}_{int(sym)}_{L}"
                    tab[key] = npy(d.Window(L, window=w, norm=norm, symmetric=sym, dtype=f64).window)
    fq = d.FrequencyTransform(19, 29, 0.1, dtype=f64)
    tab["freqt_19_29_0.1"] = npy(fq.A)
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **tab)

    # ---------------------------------------------------------------- data.wav
    pcm, sr = read_wav_int16(os.path.join(REF, "assets", "data.wav"))
    assert sr == 16000 and pcm.shape == (19200,)
    g1 = {"pcm": pcm, "sample_rate": np.int64(sr)}
    trace_frames = np.array([0, 50, 100, 150, 239])
    for name, dt in DT.items():
        x = torch.tensor(pcm.astype(np.float64) / 32768.0, dtype=dt)
        stft = d.STFT(400, 80, 512, dtype=dt)
        X = stft(x)
        g1[f"stft_power_{name}"] = npy(X)
        mcep = d.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, dtype=dt)
        g1[f"mcep_{name}"] = npy(mcep(X))
        lpc = d.LPC(400, 24, eps=1e-5, dtype=dt)
        xw = d.Window(400, dtype=dt)(d.Frame(400, 80)(x))
        g1[f"acorr_{name}"] = npy(d.Autocorrelation(400, 24)(xw))
        g1[f"lpc_{name}"] = npy(lpc(xw))
        # Newton trace mc_k, k = 0..10 for a handful of frames
        tr = []
        for k in range(11):
            m = d.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=k, dtype=dt)
            tr.append(npy(m(X[trace_frames])))
        g1[f"mcep_trace_{name}"] = np.stack(tr)  # (11, 5, 25)
    g1["trace_frames"] = trace_frames
    np.savez_compressed(os.path.join(HERE, "datawav.npz"), **g1)

    # ------------------------------------------------------------- randn(2,16000)
    g2 = {}
    x64 = torch.randn(2, 16000, dtype=f64)
    x64 = x64.to(f32).to(f64)  # exactly representable in both dtypes
    g2["x"] = npy(x64.to(f32))
    for name, dt in DT.items():
        x = x64.to(dt).clone().requires_grad_(True)
        stft = d.STFT(400, 80, 512, dtype=dt)
        mcep = d.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, dtype=dt)
        X = stft(x)
        g2[f"stft_power_{name}"] = npy(X)
        g2[f"stft_complex_{name}"] = npy(d.STFT(400, 80, 512, out_format="complex", dtype=dt)(x.detach()))
        mcv = mcep(X)
        g2[f"mcep_{name}"] = npy(mcv)
        # gradient goldens: d mean(mcep(stft(x))) / dx, and through STFT alone with a fixed cotangent
        (gx,) = torch.autograd.grad(mcv.mean(), x, retain_graph=True)
        g2[f"grad_mcep_mean_{name}"] = npy(gx)
        (gx2,) = torch.autograd.grad(torch.log(X).mean(), x)
        g2[f"grad_logstft_mean_{name}"] = npy(gx2)
        # mcep alone: gradient wrt the power spectrum for a weighted loss
        Xl = X.detach().clone().requires_grad_(True)
        wts = torch.linspace(-1, 1, 25, dtype=dt)
        (gX,) = torch.autograd.grad((mcep(Xl) * wts).sum(), Xl)
        g2[f"grad_mcep_wsum_wrt_X_{name}"] = npy(gX)
        # LPC branch
        xl = x64.to(dt).clone().requires_grad_(True)
        fr = d.Frame(400, 80)
        wn = d.Window(400, dtype=dt)
        lpc = d.LPC(400, 24, eps=1e-5, dtype=dt)
        a = lpc(wn(fr(xl)))
        g2[f"lpc_{name}"] = npy(a)
        (gl,) = torch.autograd.grad((a * wts).sum(), xl)
        g2[f"grad_lpc_wsum_{name}"] = npy(gl)
    np.savez_compressed(os.path.join(HERE, "randn.npz"), **g2)

    # ------------------------------------------- the reference tests' small grids
    g6 = {}
    # Frame: tests/test_frame.py  fl,fp in 1..5 x center x zmean on ramp(19)-like input; + pad modes
    xr = torch.arange(20, dtype=f64)
    g6["frame_x"] = npy(xr)
    for fl in range(1, 6):
        for fp in range(1, 6):
            for center in (True, False):
                for zmean in (True, False):
                    y = d.Frame(fl, fp, center=center, zmean=zmean)(xr)
                    g6[f"frame_{fl}_{fp}_{int(center)}_{int(zmean)}"] = npy(y)
    xm = torch.randn(3, 37, dtype=f64)
    g6["frame_modes_x"] = npy(xm)
    for mode in ("constant", "reflect", "replicate", "circular"):
        for center in (True, False):
            g6[f"frame_mode_{mode}_{int(center)}"] = npy(d.Frame(12, 5, center=center, mode=mode)(xm))
    # fftr: tests/test_fftr.py  L=16, M=12, 5 formats
    xf = torch.randn(2, 13, dtype=f64)
    g6["fftr_x"] = npy(xf)
    for o in range(5):
        y = d.RealValuedFastFourierTransform(16, out_format=o, dtype=f64)(xf)
        if o == 0:
            y = torch.view_as_real(y)
        g6[f"fftr_{o}"] = npy(y)
    # spec: tests/test_spec.py  L=16, eps=0.01, 4 formats x relative floor, b / a / both branches
    sb = torch.randn(2, 16, dtype=f64)
    sa = torch.randn(2, 16, dtype=f64)
    g6["spec_b"], g6["spec_a"] = npy(sb), npy(sa)
    for o in range(4):
        for rf in (None, -40):
            sp = d.Spectrum(16, eps=0.01, relative_floor=rf, out_format=o)
            g6[f"spec_ba_{o}_{rf}"] = npy(sp(sb, sa))
            g6[f"spec_b_{o}_{rf}"] = npy(sp(sb))
    g6["spec_b_only"] = npy(d.Spectrum(16)(sb))
    g6["spec_a_only"] = npy(d.Spectrum(16)(None, sa))
    # stft: tests/test_stft.py  T=100 P=10 L1=12 L2=16 hamming(w=1) power-norm(n=1) eps=1e-6
    xs = torch.randn(100, dtype=f64)
    g6["stft_x"] = npy(xs)
    kw = dict(frame_length=12, frame_period=10, fft_length=16, window=1, norm=1, eps=1e-6)
    g6["stft_power"] = npy(d.STFT(**kw, dtype=f64)(xs))
    g6["stft_complex"] = npy(torch.view_as_real(d.STFT(**kw, out_format="complex", dtype=f64)(xs)))
    for o in ("db", "log-magnitude", "magnitude"):
        g6[f"stft_{o}"] = npy(d.STFT(**kw, out_format=o, dtype=f64)(xs))
    g6["stft_relfloor"] = npy(d.STFT(**kw, relative_floor=-20, dtype=f64)(xs))
    g6["stft_zmean_nocenter_reflect"] = npy(
        d.STFT(**kw, center=False, zmean=True, mode="reflect", dtype=f64)(xs)
    )
    # odd frame length / non power-of-two fft / ragged T
    xo = torch.randn(3, 61, dtype=f64)
    g6["stft_odd_x"] = npy(xo)
    g6["stft_odd"] = npy(d.STFT(9, 4, 20, dtype=f64)(xo))
    # freqt: tests/test_freqt.py  m=19 -> M=29 alpha=0.1
    cq = torch.randn(2, 20, dtype=f64)
    g6["freqt_c"] = npy(cq)
    g6["freqt_out"] = npy(d.FrequencyTransform(19, 29, 0.1, dtype=f64)(cq))
    # mcep: tests/test_mcep.py  L=32 M in {0,7,8} n_iter in {0,3} alpha=0.1, Spectrum(L, eps=0) of nrand
    xn = torch.randn(2, 32, dtype=f64)
    g6["mcep_x"] = npy(xn)
    S = d.Spectrum(32, eps=0)(xn)
    g6["mcep_S"] = npy(S)
    for M in (0, 7, 8, 16):
        for n_iter in (0, 3):
            m = d.MelCepstralAnalysis(fft_length=32, cep_order=M, alpha=0.1, n_iter=n_iter, dtype=f64)
            g6[f"mcep_{M}_{n_iter}"] = npy(m(S))
    # acorr: tests/test_acorr.py  M in {12,13}?? L=14, 4 formats
    xa = torch.randn(2, 14, dtype=f64)
    g6["acorr_x"] = npy(xa)
    for M in (12, 13):
        for o in range(4):
            g6[f"acorr_{M}_{o}"] = npy(d.Autocorrelation(14, M, out_format=o)(xa))
    # levdur: tests/test_levdur.py M=30 L=52 ; lpc: tests/test_lpc.py M=14 L=30
    xl = torch.randn(2, 52, dtype=f64)
    g6["levdur_x"] = npy(xl)
    r = d.Autocorrelation(52, 30)(xl)
    g6["levdur_r"] = npy(r)
    g6["levdur_out_eps0"] = npy(d.LevinsonDurbin(30, eps=0, dtype=f64)(r))
    g6["levdur_out_eps1e-5"] = npy(d.LevinsonDurbin(30, eps=1e-5, dtype=f64)(r))
    xp = torch.randn(2, 30, dtype=f64)
    g6["lpc_x"] = npy(xp)
    g6["lpc_out"] = npy(d.LPC(30, 14, eps=0, dtype=f64)(xp))
    np.savez_compressed(os.path.join(HERE, "grids.npz"), **g6)

    # ------------------------------------------------------------------ mel filter bank / MFCC
    g7 = {}
    X64 = torch.from_numpy(g1["stft_power_f64"]).clone()      # data.wav power spectrum (240, 257)
    for name, dt in DT.items():
        Xd = X64.to(dt)
        fb = d.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, out_format="y,E", dtype=dt)
        y, E = fb(Xd)
        g7[f"fbank_y_{name}"], g7[f"fbank_E_{name}"] = npy(y), npy(E)
        fbp = d.MelFilterBankAnalysis(fft_length=512, n_channel=80, sample_rate=16000, f_min=50, f_max=7600, floor=1e-3,
                                      gamma=-0.5, scale="mel", use_power=True, out_format="yE", dtype=dt)
        g7[f"fbank_pow_yE_{name}"] = npy(fbp(Xd))
        mf = d.MFCC(fft_length=512, mfcc_order=12, n_channel=40, sample_rate=16000, lifter=22, out_format="ycE", dtype=dt)
        g7[f"mfcc_ycE_{name}"] = npy(mf(Xd))
        Xg = Xd.clone().requires_grad_(True)
        out = mf(Xg)
        (out * torch.linspace(-1, 1, out.size(-1), dtype=dt)).sum().backward()
        g7[f"grad_mfcc_wsum_{name}"] = npy(Xg.grad)
        Xg = Xd.clone().requires_grad_(True)
        out = fbp(Xg)
        (out * torch.linspace(1, 2, out.size(-1), dtype=dt)).sum().backward()
        g7[f"grad_fbank_pow_wsum_{name}"] = npy(Xg.grad)
    g7["H_htk40"] = npy(d.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, dtype=f64).H)
    g7["H_mel80"] = npy(fbp.H.double()) if False else npy(d.MelFilterBankAnalysis(
        fft_length=512, n_channel=80, sample_rate=16000, f_min=50, f_max=7600, scale="mel", dtype=f64).H)
    g7["H_bark_erb"] = npy(d.MelFilterBankAnalysis(fft_length=2048, n_channel=40, sample_rate=8000, scale="bark",
                                                   erb_factor=0.5, dtype=f64).H)
    g7["dct2_40"] = npy(d.DCT(40, 2, dtype=f64).W)
    # doctest known answers (fbank.py:142-153, mfcc.py:132-143) with their input spectrum
    st = d.STFT(frame_length=10, frame_period=10, fft_length=32)
    xs = st(d.ramp(19))
    g7["doc_spec"] = npy(xs)
    g7["doc_fbank"] = npy(d.MelFilterBankAnalysis(fft_length=32, n_channel=4, sample_rate=8000)(xs))
    g7["doc_mfcc"] = npy(d.MFCC(fft_length=32, mfcc_order=4, n_channel=8, sample_rate=8000)(xs))
    # the reference test's small configuration (tests/test_fbank.py:24-36): floor = 1
    xr = torch.rand(2, 17, dtype=f64) * 4
    g7["grid_x"] = npy(xr)
    g7["grid_fbank_yE"] = npy(d.MelFilterBankAnalysis(fft_length=32, n_channel=10, sample_rate=8000, f_min=300, f_max=3400,
                                                      floor=1, out_format=1, dtype=f64)(xr))
    np.savez_compressed(os.path.join(HERE, "fbank.npz"), **g7)

    # ------------------------------------------------------------------ inverse path
    g8 = {}
    yc = torch.randn(3, 9, dtype=torch.complex128)
    g8["ifftr_y"] = npy(yc)
    g8["ifftr_full"] = npy(d.RealValuedInverseFastFourierTransform(16, dtype=f64)(yc))
    g8["ifftr_5"] = npy(d.RealValuedInverseFastFourierTransform(16, 5, dtype=f64)(yc))
    fr = torch.randn(2, 7, 12, dtype=f64)
    g8["unframe_y"] = npy(fr)
    for tag, kw, ol in (("default", dict(), None), ("nocenter", dict(center=False), None),
                        ("blackman_20", dict(window="blackman", norm="power"), 20),
                        ("hanning_long", dict(window="hanning"), 40), ("nocenter_9", dict(center=False), 9)):
        g8[f"unframe_{tag}"] = npy(d.Unframe(12, 4, dtype=f64, **kw)(fr, out_length=ol))
    g8["unframe_doc"] = npy(d.Unframe(5, 2)(d.Frame(5, 2)(d.ramp(1, 9))))
    ys = torch.randn(2, 10, 33, dtype=torch.complex128)
    g8["istft_y"] = npy(ys)
    g8["istft_default"] = npy(d.ISTFT(40, 8, 64, dtype=f64)(ys))
    g8["istft_len70"] = npy(d.ISTFT(40, 8, 64, dtype=f64)(ys, out_length=70))
    g8["istft_nocenter_hamming"] = npy(d.ISTFT(40, 8, 64, center=False, window="hamming", norm="none", dtype=f64)(ys))
    ysg = ys.clone().requires_grad_(True)
    out = d.ISTFT(40, 8, 64, dtype=f64)(ysg)
    (out * torch.linspace(-1, 1, out.size(-1), dtype=f64)).sum().backward()
    g8["grad_istft_wsum"] = npy(ysg.grad)
    # analysis -> synthesis round trip at the bench configuration on data.wav (tests/test_istft.py:26-58)
    xw = torch.from_numpy(pcm.astype(np.float64) / 32768.0)
    for name, dt in DT.items():
        st = d.STFT(400, 80, 512, out_format="complex", dtype=dt)
        ist = d.ISTFT(400, 80, 512, dtype=dt)
        g8[f"roundtrip_{name}"] = npy(ist(st(xw.to(dt)), out_length=xw.numel()))
    np.savez_compressed(os.path.join(HERE, "inverse.npz"), **g8)

    # ------------------------------------------------------------------ Griffin-Lim (griffin.py)
    g9 = {}
    sp = dict(frame_length=3, frame_period=1, fft_length=8)
    xd = d.ramp(1, 3)
    g9["doc_y"] = npy(d.GriffinLim(**sp, n_iter=10, init_phase="zeros")(d.STFT(**sp, out_format="power")(xd), out_length=3))
    xs = xw[2000:6000]
    for name, dt in DT.items():
        X = d.STFT(400, 80, 512, out_format="power", dtype=dt)(xs.to(dt))
        if name == "f64":
            g9["seg_power"] = npy(X)
        for it in (0, 1, 5):
            g9[f"seg_iter{it}_{name}"] = npy(d.GriffinLim(400, 80, 512, n_iter=it, init_phase="zeros", dtype=dt)(X, out_length=xs.numel()))
    Xb = d.STFT(64, 16, 64, window="hanning", norm="none", out_format="power", dtype=f64)(torch.randn(3, 700, dtype=f64))
    g9["rand_power"] = npy(Xb)
    g9["rand_iter4"] = npy(d.GriffinLim(64, 16, 64, window="hanning", norm="none", n_iter=4, alpha=0.5, beta=0.2, gamma=1.3,
                                        init_phase="zeros", dtype=f64)(Xb))
    np.savez_compressed(os.path.join(HERE, "griffin.npz"), **g9)

    # ------------------------------------------------------------------ cepstral analysis (fftcep.py)
    g10 = {}
    g10["doc_c"] = npy(d.CepstralAnalysis(fft_length=16, cep_order=3)(d.STFT(frame_length=10, frame_period=10, fft_length=16)(d.ramp(19))))
    Xr = torch.distributions.Gamma(2.0, 1.0).sample((6, 17)).to(f64) + 1e-3
    g10["rand_x"] = npy(Xr)
    for tag, kw in (("i0", dict()), ("i3", dict(n_iter=3)), ("i2a", dict(n_iter=2, accel=0.5))):
        g10[f"rand_{tag}"] = npy(d.CepstralAnalysis(fft_length=32, cep_order=5, **kw)(Xr))
    Xe = torch.distributions.Gamma(2.0, 1.0).sample((4, 9)).to(f64) + 1e-3
    g10["edge_x"] = npy(Xe)
    g10["edge_i2"] = npy(d.CepstralAnalysis(fft_length=16, cep_order=8, n_iter=2)(Xe))      # H == N: both ends halved
    Xw = torch.from_numpy(np.load(os.path.join(HERE, "datawav.npz"))["stft_power_f64"])
    for name, dt in DT.items():
        for tag, kw in (("i0", dict()), ("i3a", dict(n_iter=3, accel=0.2))):
            g10[f"wav_{tag}_{name}"] = npy(d.CepstralAnalysis(fft_length=512, cep_order=24, **kw)(Xw.to(dt)))
    for tag, kw in (("i0", dict()), ("i3a", dict(n_iter=3, accel=0.2))):
        Xg = Xw.clone().requires_grad_(True)
        out = d.CepstralAnalysis(fft_length=512, cep_order=24, **kw)(Xg)
        (out * torch.linspace(-1, 1, 25, dtype=f64)).sum().backward()
        g10[f"grad_wav_{tag}"] = npy(Xg.grad)
    np.savez_compressed(os.path.join(HERE, "fftcep.npz"), **g10)

    meta = {
        "reference": "sp-nitech/diffsptk 4.0.0 (/root/reference)",
        "torch": torch.__version__,
        "numpy": np.__version__,
        "threads": torch.get_num_threads(),
        "seed": 0,
    }
    with open(os.path.join(HERE, "META.json"), "w") as f:
        json.dump(meta, f, indent=1)
    for fn in ("tables.npz", "datawav.npz", "randn.npz", "grids.npz", "fbank.npz", "inverse.npz", "griffin.npz", "fftcep.npz"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)) // 1024, "KiB")


if __name__ == "__main__":
    main()
