#!/usr/bin/env python3
"""The BASELINE geometry (frame 400, period 80, fft 512, order 24) under every STFT / Frame option, through the reference in float64:
STFT alone (5 formats x 8 option sets) and the chains STFT -> mcep / fbank / mfcc and Frame -> Window -> LPC (5 option sets each),
with the output and the input gradient of a seeded linear functional.  The GPU test runs the same cases in float32 through the
TUNED kernels and the fuse(...) launches.  Build container only; writes tests/golden/tuned_grid.npz (+ .json: the cases)."""
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
OPTS = [{}, {"zmean": True}, {"mode": "reflect"}, {"mode": "replicate"}, {"mode": "circular"}, {"relative_floor": -60}, {"center": False},
        {"window": "hanning", "norm": "magnitude", "eps": 1e-6}]


def main():
    g = torch.Generator().manual_seed(0)
    # speech-like dynamics: noise with a slowly varying envelope and an offset (zmean matters)
    T = 1200
    env = (0.05 + torch.rand(1, 1, generator=g, dtype=f64)) * (1.0 + 0.9 * torch.sin(torch.linspace(0, 9, T, dtype=f64))[None])
    x = env * torch.randn(1, T, generator=g, dtype=f64) + 0.02
    out, meta = {"x": x.numpy()}, []

    def record(tag, fn, spec):
        xg = x.clone().requires_grad_(True)
        y = fn(xg)
        yr = torch.view_as_real(y) if y.is_complex() else y
        w = torch.cos(0.37 * torch.arange(yr.numel(), dtype=f64) + len(meta)).reshape(yr.shape)   # (the test rebuilds it: nothing to store)
        (gx,) = torch.autograd.grad((yr * w).sum(), xg)
        i = len(meta)
        out[f"y{i}"], out[f"gx{i}"] = yr.detach().numpy().astype(np.float32), gx.numpy().astype(np.float32)
        meta.append({"tag": tag, **spec})

    for fmt in ("power", "magnitude", "db", "log-magnitude", "complex"):
        for o in OPTS:
            if fmt == "complex" and ("relative_floor" in o or "eps" in o):
                continue
            kw = {"out_format": fmt, **o}
            m = d.STFT(FL, FP, NFFT, dtype=f64, **kw)
            record(f"stft {kw}", m, {"kind": "stft", "stft": kw})
    for o in OPTS[:7]:
        st = d.STFT(FL, FP, NFFT, dtype=f64, **o)
        mc = d.MelCepstralAnalysis(fft_length=NFFT, cep_order=M, alpha=0.42, n_iter=10, dtype=f64)
        record(f"mcep(stft {o})", lambda z, st=st, mc=mc: mc(st(z)), {"kind": "mcep", "stft": o})
    for o in OPTS[:5]:
        for use_power, fmt in ((True, "power"), (False, "magnitude")):
            st = d.STFT(FL, FP, NFFT, dtype=f64, out_format=fmt, **o)
            fb = d.FBANK(fft_length=NFFT, n_channel=40, sample_rate=16000, use_power=use_power, dtype=f64)
            record(f"fbank(stft {fmt} {o})", lambda z, st=st, fb=fb: fb(st(z)), {"kind": "fbank", "stft": {"out_format": fmt, **o}, "use_power": use_power})
        st = d.STFT(FL, FP, NFFT, dtype=f64, **o)
        mf = d.MFCC(fft_length=NFFT, mfcc_order=12, n_channel=40, sample_rate=16000, dtype=f64)
        record(f"mfcc(stft {o})", lambda z, st=st, mf=mf: mf(st(z)), {"kind": "mfcc", "stft": o})
    for o in ({}, {"zmean": True}, {"mode": "reflect"}, {"center": False}, {"mode": "circular", "zmean": True}):
        fr = d.Frame(FL, FP, **o)
        wi = d.Window(FL, dtype=f64)
        lp = d.LPC(FL, M, eps=1e-5, dtype=f64)
        record(f"lpc(window(frame {o}))", lambda z, fr=fr, wi=wi, lp=lp: lp(wi(fr(z))), {"kind": "lpc", "frame": o})
    np.savez_compressed(os.path.join(HERE, "tuned_grid.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "tuned_grid.json"), "w"), separators=(",", ":"))
    print("wrote", len(meta), "cases")


if __name__ == "__main__":
    main()
