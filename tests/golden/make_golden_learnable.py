#!/usr/bin/env python3
"""TRAINED (perturbed) learnable parameters through the reference: a checkpoint in the reference's state-dict keys, an input, the
output and the gradients w.r.t. the input and every parameter -- float32, seed 0.  Build container only (imports /root/reference);
writes tests/golden/learnable.npz (arrays only).  The GPU test loads the checkpoint into this repo's modules with load_state_dict.
"""
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
f64 = torch.float32   # (the reference's learnable STFT builds its DFT matrix in the default dtype whatever `dtype` says: float32 throughout)


def npy(t):
    return t.detach().cpu().numpy()


def perturb(m, g, scale=0.05):
    with torch.no_grad():
        for p in m.parameters():
            p.add_(scale * p.abs().max() * torch.randn(p.shape, generator=g, dtype=p.dtype))


def record(out, tag, m, x, y_to_real=lambda y: y):
    """state (reference keys), input, output, gradients of sum(w * y) w.r.t. x and the parameters"""
    for k, v in m.state_dict().items():
        out[f"{tag}/state/{k}"] = npy(v)
    xg = x.clone().requires_grad_(True)
    y = y_to_real(m(xg))
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(5), dtype=y.dtype)
    grads = torch.autograd.grad((y * w).sum(), [xg] + list(m.parameters()))
    out[f"{tag}/x"], out[f"{tag}/y"], out[f"{tag}/w"], out[f"{tag}/gx"] = npy(x), npy(y), npy(w), npy(grads[0])
    for (k, _p), gp in zip(m.named_parameters(), grads[1:]):
        out[f"{tag}/gparam/{k}"] = npy(gp)


def main():
    g = torch.Generator().manual_seed(0)
    out = {}
    for fmt in ("power", "complex"):
        m = d.STFT(48, 20, 64, out_format=fmt, learnable=True, dtype=f64)
        perturb(m, g)
        x = torch.randn(3, 250, generator=g, dtype=f64)
        record(out, f"stft_{fmt}", m, x, (lambda y: torch.view_as_real(y)) if fmt == "complex" else (lambda y: y))
    m = d.ISTFT(48, 20, 64, learnable=True, dtype=f64)
    perturb(m, g)
    yc = torch.randn(3, 13, 33, 2, generator=g, dtype=f64)
    xg = yc.clone().requires_grad_(True)
    for k, v in m.state_dict().items():
        out[f"istft/state/{k}"] = npy(v)
    y = m(torch.view_as_complex(xg), out_length=250)
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(5), dtype=f64)
    grads = torch.autograd.grad((y * w).sum(), [xg] + list(m.parameters()))
    out["istft/x"], out["istft/y"], out["istft/w"], out["istft/gx"] = npy(yc), npy(y), npy(w), npy(grads[0])
    for (k, _p), gp in zip(m.named_parameters(), grads[1:]):
        out[f"istft/gparam/{k}"] = npy(gp)
    m = d.MFCC(fft_length=64, mfcc_order=6, n_channel=10, sample_rate=8000, learnable=True, dtype=f64)
    perturb(m, g, 0.02)
    with torch.no_grad():
        m.fbank.H.clamp_(min=0)
    record(out, "mfcc", m, torch.rand(3, 7, 33, generator=g, dtype=f64) + 0.1)
    np.savez_compressed(os.path.join(HERE, "learnable.npz"), **out)
    print("wrote", len(out), "arrays:", sorted(out)[:6], "...")


if __name__ == "__main__":
    main()
