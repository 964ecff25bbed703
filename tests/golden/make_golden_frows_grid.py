#!/usr/bin/env python3
"""SURVEY 8(f) rows at the BASELINE geometry through the reference in float64 (outputs + input gradients of a cosine-weighted sum): ISTFT of a
complex STFT, mel-generalized cepstral analysis (gamma -0.5 / -1 / c = 3), cepstral analysis, mgc2sp in 4 formats, mc2b / b2mc, the MLSA filter in
its three torchlpc-free modes (minimum / maximum / zero phase).  Build container only; writes tests/golden/frows_grid.npz (+ .json)."""
import json
import os
import sys
import types

import numpy as np
import torch

for name in ("torchaudio", "soundfile"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, "/root/reference")
import diffsptk as d  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
f64 = torch.float64
FL, FP, NFFT, M = 400, 80, 512, 24


def main():
    g = torch.Generator().manual_seed(0)
    T = 1200
    env = 0.2 * (1.0 + 0.9 * torch.sin(torch.linspace(0, 7, T, dtype=f64))[None])
    x = env * torch.randn(1, T, generator=g, dtype=f64) + 0.01
    stft = d.STFT(FL, FP, NFFT, dtype=f64)
    X = stft(x).detach()
    mc = d.MelCepstralAnalysis(fft_length=NFFT, cep_order=M, alpha=0.42, n_iter=10, dtype=f64)(X).detach()
    out, meta = {"x": x.numpy(), "X": X.numpy(), "mc": mc.numpy()}, []

    def record(tag, fn, inp, spec):
        a = inp.clone().requires_grad_(True)
        y = fn(a)
        yr = torch.view_as_real(y) if y.is_complex() else y
        i = len(meta)
        w = torch.cos(0.37 * torch.arange(yr.numel(), dtype=f64) + i).reshape(yr.shape)
        (ga,) = torch.autograd.grad((yr * w).sum(), a)
        out[f"y{i}"], out[f"g{i}"] = yr.detach().numpy().astype(np.float32), ga.numpy().astype(np.float32)
        meta.append({"tag": tag, **spec})

    stc = d.STFT(FL, FP, NFFT, out_format="complex", dtype=f64)
    ist = d.ISTFT(FL, FP, NFFT, dtype=f64)
    record("istft(stft complex)", lambda z: ist(stc(z), out_length=T), x, {"kind": "istft", "input": "x"})
    for kw in ({"gamma": -0.5}, {"gamma": -1.0}, {"c": 3}):
        m = d.MelGeneralizedCepstralAnalysis(fft_length=NFFT, cep_order=M, alpha=0.42, n_iter=10, dtype=f64, **kw)
        record(f"mgcep {kw}", m, X, {"kind": "mgcep", "kwargs": kw, "input": "X"})
    for kw in ({"n_iter": 0}, {"n_iter": 3}, {"n_iter": 3, "accel": 0.5}):
        m = d.CepstralAnalysis(fft_length=NFFT, cep_order=M, **kw)
        record(f"fftcep {kw}", m, X, {"kind": "fftcep", "kwargs": kw, "input": "X"})
    for fmt in ("power", "db", "log-magnitude", "complex"):
        m = d.MelGeneralizedCepstrumToSpectrum(M, NFFT, alpha=0.42, out_format=fmt, dtype=f64)
        record(f"mgc2sp {fmt}", m, mc, {"kind": "mgc2sp", "kwargs": {"out_format": fmt}, "input": "mc"})
    record("mc2b", d.MelCepstrumToMLSADigitalFilterCoefficients(M, alpha=0.42, dtype=f64), mc, {"kind": "mc2b", "input": "mc"})
    record("b2mc", d.MLSADigitalFilterCoefficientsToMelCepstrum(M, alpha=0.42, dtype=f64), mc, {"kind": "b2mc", "input": "mc"})
    for mode, kw in (("multi-stage", {"taylor_order": 20, "cep_order": 99}), ("single-stage", {"ir_length": 400, "n_fft": 512}),
                     ("freq-domain", {"frame_length": FL, "fft_length": NFFT})):
        for phase in ("minimum", "maximum", "zero"):
            m = d.MLSA(M, FP, alpha=0.42, mode=mode, phase=phase, dtype=f64, **kw)
            record(f"mlsa {mode} {phase}: d/dx", lambda z, m=m: m(z, mc), x, {"kind": "mlsa", "mode": mode, "phase": phase, "kwargs": kw, "input": "x", "wrt": "x"})
            if phase == "minimum":
                record(f"mlsa {mode} {phase}: d/dmc", lambda c, m=m: m(x, c), mc, {"kind": "mlsa", "mode": mode, "phase": phase, "kwargs": kw, "input": "mc", "wrt": "mc"})
    np.savez_compressed(os.path.join(HERE, "frows_grid.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "frows_grid.json"), "w"), separators=(",", ":"))
    print("wrote", len(meta), "cases")


if __name__ == "__main__":
    main()
