#!/usr/bin/env python3
"""An OPTION GRID through the reference's modules of the path and its consumers: for every case the input (seeded), the reference's
output and the gradient of a seeded linear functional of it w.r.t. the input (float64) -- a crash / shape / value smoke of rarely used option paths.  Build container only
(imports /root/reference); writes tests/golden/option_grid.npz + option_grid.json (the cases: module, args, kwargs, input recipe).
"""
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


def cases():
    c = []
    # (module, args, kwargs, input recipe): recipes: "wave:T", "frames:N:L", "spec:N:K" (positive), "cep:N:M1", "cplx:N:K"
    for w in ("blackman", "hamming", "hanning", "bartlett", "trapezoidal", "rectangular", "nuttall"):
        for norm in ("none", "power", "magnitude"):
            c.append(("Window", [30, 32], {"window": w, "norm": norm, "symmetric": w != "hamming"}, "frames:5:30"))
    for mode in ("constant", "reflect", "replicate", "circular"):
        for center in (True, False):
            c.append(("Frame", [30, 7], {"center": center, "zmean": mode == "reflect", "mode": mode}, "wave:200"))
    for fmt in ("db", "log-magnitude", "magnitude", "power", "complex"):
        for kw in ({}, {"zmean": True}, {"mode": "reflect"}, {"relative_floor": -60}, {"window": "hanning", "norm": "magnitude"}, {"center": False}):
            if fmt == "complex" and "relative_floor" in kw:
                continue
            c.append(("STFT", [30, 7, 32], {"out_format": fmt, **kw}, "wave:200"))
    for fmt in ("naive", "normalized", "biased", "unbiased"):
        c.append(("Autocorrelation", [30, 9], {"out_format": fmt}, "frames:5:30"))
    for eps in (None, 0.0, 1e-3):
        c.append(("LPC", [30, 9], {"eps": eps}, "frames:5:30"))
    for M, n_iter, alpha in ((0, 2, 0.1), (5, 0, 0.3), (12, 4, 0.42), (16, 3, -0.2)):
        c.append(("MelCepstralAnalysis", [], {"fft_length": 32, "cep_order": M, "alpha": alpha, "n_iter": n_iter}, "spec:5:17"))
    for gamma, cc in ((-0.5, None), (-1.0, None), (0.0, None), (0, 3)):
        kw = {"fft_length": 32, "cep_order": 8, "alpha": 0.2, "n_iter": 3}
        kw.update({"c": cc} if cc else {"gamma": gamma})
        c.append(("MelGeneralizedCepstralAnalysis", [], kw, "spec:5:17"))
    for accel, n_iter in ((0.0, 0), (0.0, 3), (0.5, 2)):
        c.append(("CepstralAnalysis", [], {"fft_length": 32, "cep_order": 8, "accel": accel, "n_iter": n_iter}, "spec:5:17"))
    for scale in ("htk", "mel", "inverted-mel", "bark", "linear"):
        for fmt in ("y", "yE", "y,E"):
            c.append(("MelFilterBankAnalysis", [], {"fft_length": 32, "n_channel": 6, "sample_rate": 8000, "scale": scale, "out_format": fmt,
                                                    "gamma": -0.5 if scale == "bark" else 0, "use_power": scale == "mel"}, "spec:5:17"))
    for fmt in ("y", "yE", "yc", "ycE"):
        c.append(("MFCC", [], {"fft_length": 32, "mfcc_order": 4, "n_channel": 6, "sample_rate": 8000, "lifter": 3, "out_format": fmt}, "spec:5:17"))
    for t in (1, 2, 3, 4):
        c.append(("DCT", [8], {"dct_type": t}, "frames:5:8"))
    for kw in ({}, {"center": False}, {"window": "hanning", "norm": "none"}):
        c.append(("ISTFT", [30, 7, 32], kw, "cplx:29:17"))
        c.append(("Unframe", [30, 7], {k: v for k, v in kw.items()}, "frames:29:30"))
    for fmt in ("db", "log-magnitude", "magnitude", "power", "complex", "cycle", "radian", "degree"):
        c.append(("MelGeneralizedCepstrumToSpectrum", [8, 32], {"alpha": 0.2, "gamma": -0.5 if fmt == "db" else 0, "out_format": fmt, "norm": fmt == "power", "mul": False}, "cep:5:9"))
    c.append(("MelCepstrumToMLSADigitalFilterCoefficients", [8], {"alpha": 0.3}, "cep:5:9"))
    c.append(("MLSADigitalFilterCoefficientsToMelCepstrum", [8], {"alpha": 0.3}, "cep:5:9"))
    for fmt in ("complex", "real", "imaginary", "amplitude", "power"):
        c.append(("RealValuedFastFourierTransform", [32], {"out_format": fmt}, "frames:5:20"))
    for fmt in ("db", "log-magnitude", "magnitude", "power"):
        c.append(("Spectrum", [32], {"out_format": fmt, "eps": 1e-6, "relative_floor": -40 if fmt == "db" else None}, "frames:5:9"))
    c.append(("FrequencyTransform", [8, 11, 0.3], {}, "cep:5:9"))
    c.append(("GeneralizedCepstrumGainNormalization", [8], {"gamma": -0.5}, "cep:5:9"))
    c.append(("GeneralizedCepstrumInverseGainNormalization", [8], {"c": 2}, "cep:5:9"))
    c.append(("MelGeneralizedCepstrumToMelGeneralizedCepstrum", [8, 10], {"in_alpha": 0.1, "out_alpha": 0.3, "in_gamma": -0.5, "out_gamma": -1.0, "in_norm": False, "out_mul": True, "n_fft": 64}, "cep:5:9"))
    return c


def make_input(recipe, g):
    kind, *dims = recipe.split(":")
    dims = [int(v) for v in dims]
    if kind == "wave":
        return torch.randn(2, dims[0], generator=g, dtype=f64)
    if kind == "frames":
        return torch.randn(2, dims[0], dims[1], generator=g, dtype=f64)
    if kind == "spec":
        return torch.rand(2, dims[0], dims[1], generator=g, dtype=f64) + 0.05
    if kind == "cep":
        return 0.3 * torch.randn(2, dims[0], dims[1], generator=g, dtype=f64)
    if kind == "cplx":
        return torch.complex(torch.randn(2, dims[0], dims[1], generator=g, dtype=f64), torch.randn(2, dims[0], dims[1], generator=g, dtype=f64))
    raise ValueError(recipe)


def main():
    out, meta = {}, []
    g = torch.Generator().manual_seed(0)
    for i, (name, args, kwargs, recipe) in enumerate(cases()):
        x = make_input(recipe, g)
        try:
            m = getattr(d, name)(*args, **kwargs, dtype=f64) if name not in ("Frame",) else getattr(d, name)(*args, **kwargs)
        except TypeError:
            m = getattr(d, name)(*args, **kwargs)
        extra = {"out_length": 200} if name in ("ISTFT", "Unframe") else {}
        xg = x.clone().requires_grad_(True)
        y = m(xg, **extra)
        ys = y if isinstance(y, (tuple, list)) else (y,)
        # gradient of sum_j <w_j, y_j> w.r.t. the input, w_j seeded per case (complex outputs / inputs through their real views)
        gw = torch.Generator().manual_seed(1000 + i)
        loss = 0
        for j, t in enumerate(ys):
            tr = torch.view_as_real(t) if t.is_complex() else t
            wj = torch.randn(tr.shape, generator=gw, dtype=f64)
            out[f"w{i}_{j}"] = wj.numpy()
            loss = loss + (tr * wj).sum()
        (gx,) = torch.autograd.grad(loss, xg)
        out[f"gx{i}"] = torch.view_as_real(gx).numpy() if gx.is_complex() else gx.numpy()
        out[f"x{i}"] = torch.view_as_real(x).numpy() if x.is_complex() else x.numpy()
        for j, t in enumerate(ys):
            t = torch.view_as_real(t) if t.is_complex() else t
            out[f"y{i}_{j}"] = t.detach().numpy()
        meta.append({"module": name, "args": args, "kwargs": kwargs, "input": recipe, "complex_input": bool(x.is_complex()),
                     "n_out": len(ys), "extra": extra})
    np.savez_compressed(os.path.join(HERE, "option_grid.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "option_grid.json"), "w"), separators=(",", ":"))
    print(f"wrote {len(meta)} cases")


if __name__ == "__main__":
    main()
