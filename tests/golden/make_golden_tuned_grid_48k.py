#!/usr/bin/env python3
"""The 44.1 / 48 kHz set-ups (frame 1200 / period 240 / fft 2048 / order 49, alpha 0.55; 800 / 200 / 1024 / 34, alpha 0.53) through the
reference in float64: STFT under 7 option sets and mcep(stft(x)) under 3, output + input gradient of a cosine-weighted sum (as
make_golden_tuned_grid.py).  Build container only; writes tests/golden/tuned_grid_48k.npz (+ .json)."""
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
GEOS = [(1200, 240, 2048, 49, 0.55), (800, 200, 1024, 34, 0.53)]
STFT_OPTS = [{}, {"zmean": True}, {"mode": "reflect"}, {"center": False}, {"out_format": "magnitude"}, {"out_format": "db"}, {"out_format": "complex"}]
MCEP_OPTS = [{}, {"zmean": True}, {"mode": "replicate", "relative_floor": -70}]


def main():
    g = torch.Generator().manual_seed(0)
    out, meta = {}, []
    for gi, (FL, FP, NFFT, M, alpha) in enumerate(GEOS):
        T = 12 * FP
        env = 0.2 * (1.0 + 0.9 * torch.sin(torch.linspace(0, 7, T, dtype=f64))[None])
        x = env * torch.randn(1, T, generator=g, dtype=f64) + 0.01
        out[f"x{gi}"] = x.numpy()

        def record(tag, fn, spec):
            xg = x.clone().requires_grad_(True)
            y = fn(xg)
            yr = torch.view_as_real(y) if y.is_complex() else y
            i = len(meta)
            w = torch.cos(0.37 * torch.arange(yr.numel(), dtype=f64) + i).reshape(yr.shape)
            (gx,) = torch.autograd.grad((yr * w).sum(), xg)
            out[f"y{i}"], out[f"gx{i}"] = yr.detach().numpy().astype(np.float32), gx.numpy().astype(np.float32)
            meta.append({"tag": tag, "geo": gi, "geometry": [FL, FP, NFFT, M, alpha], **spec})

        for o in STFT_OPTS:
            record(f"stft {NFFT} {o}", d.STFT(FL, FP, NFFT, dtype=f64, **o), {"kind": "stft", "stft": o})
        for o in MCEP_OPTS:
            st = d.STFT(FL, FP, NFFT, dtype=f64, **o)
            mc = d.MelCepstralAnalysis(fft_length=NFFT, cep_order=M, alpha=alpha, n_iter=10, dtype=f64)
            record(f"mcep {NFFT}/{M} (stft {o})", lambda z, st=st, mc=mc: mc(st(z)), {"kind": "mcep", "stft": o})
    np.savez_compressed(os.path.join(HERE, "tuned_grid_48k.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "tuned_grid_48k.json"), "w"), separators=(",", ":"))
    print("wrote", len(meta), "cases")


if __name__ == "__main__":
    main()
