"""Learnable bases (fftr.py:123-129, ifftr.py:125-129, fbank.py:112-122, unframe.py learnable window).

A learnable DFT matrix / filter bank / synthesis window is a TRAINING feature off the analysis hot path: the
hand-written kernels treat those tables as constants (their backward entries return no gradient for them).  As
SURVEY.md section 8(b) prescribes, these options fall back to a composition of stock PyTorch device operators
written here -- same results as the kernels at initialisation, gradients for the tables from autograd.  Nothing
in this file is used unless ``learnable`` is requested.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def dft_matrix(fft_length: int) -> np.ndarray:
    """(L, 2K) float64: [cos(2 pi n k / L) | -sin(2 pi n k / L)], K = L/2 + 1: x @ W = [Re | Im] of rfft(x)
    (the matrix fftr.py:123-129 takes from torch.fft.fft(eye))."""
    n = np.arange(fft_length, dtype=np.float64)[:, None]
    k = np.arange(fft_length // 2 + 1, dtype=np.float64)[None, :]
    ph = 2.0 * math.pi * n * k / fft_length
    return np.concatenate((np.cos(ph), -np.sin(ph)), axis=1)


def idft_matrix(fft_length: int, out_length: int | None) -> np.ndarray:
    """(2K, out_length) float64: [Re y | Im y] @ W = irfft(y)[:out_length] (ifftr.py:125-129): rows
    c_k / L cos(2 pi k n / L) over -c_k / L sin(2 pi k n / L), c = 1 at DC and Nyquist, else 2."""
    out_length = fft_length if out_length is None else out_length
    k = np.arange(fft_length // 2 + 1, dtype=np.float64)[:, None]
    n = np.arange(out_length, dtype=np.float64)[None, :]
    c = np.full((fft_length // 2 + 1, 1), 2.0)
    c[0] = c[-1] = 1.0
    ph = 2.0 * math.pi * k * n / fft_length
    return np.concatenate((c * np.cos(ph), -c * np.sin(ph)), axis=0) / fft_length


def rfft_with_basis(x: torch.Tensor, W: torch.Tensor, fft_length: int, fmt: int) -> torch.Tensor:
    """fftr.py:146-151 with the formatter of fftr.py:110-121 (fmt: 0 complex, 1 real, 2 imaginary, 3 amplitude, 4 power)."""
    if x.size(-1) < fft_length:
        x = F.pad(x, (0, fft_length - x.size(-1)))
    elif x.size(-1) > fft_length:
        x = x[..., :fft_length]
    re, im = torch.tensor_split(torch.matmul(x, W), 2, dim=-1)
    if fmt == 1:
        return re
    if fmt == 2:
        return im
    y = torch.complex(re, im)
    if fmt == 0:
        return y
    return y.abs() if fmt == 3 else y.abs().square()


def spectrum_with_basis(b, a, W, fft_length: int, eps: float, relative_floor_db, fmt: int) -> torch.Tensor:
    """spec.py:152-178 on top of the learnable amplitude transform (fmt: 0 db, 1 log-magnitude, 2 magnitude, 3 power)."""
    def amp(t):
        return rfft_with_basis(t, W, fft_length, 3)

    if b is None and a is None:
        raise ValueError("Either b or a must be specified.")
    if a is not None:   # private.py:200-209: gain K = a[0], denominator with a unit first coefficient
        K = a[..., :1]
        a1 = torch.cat((torch.ones_like(K), a[..., 1:]), dim=-1)
        X = K * (amp(b) / amp(a1)) if b is not None else K / amp(a1)
    else:
        X = amp(b)
    s = torch.square(X) + eps
    if relative_floor_db is not None:
        s = torch.maximum(s, torch.amax(s, dim=-1, keepdim=True) * 10 ** (relative_floor_db / 10))
    if fmt == 0:
        return 10 * torch.log10(s)
    if fmt == 1:
        return 0.5 * torch.log(s)
    return torch.sqrt(s) if fmt == 2 else s


def irfft_with_basis(y: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """ifftr.py:139-141."""
    return torch.matmul(torch.cat((y.real, y.imag), dim=-1), W)


def fbank_with_weights(x: torch.Tensor, H: torch.Tensor, floor: float, gamma: float, use_power: bool):
    """fbank.py:306-321 with H a Parameter: returns (y, E)."""
    s = x if use_power else torch.sqrt(x)
    y = torch.clip(torch.matmul(s, H), min=floor)
    y = torch.log(y) if gamma == 0 else (torch.pow(y, gamma) - 1) / gamma
    E = (2 * x[..., 1:-1]).sum(-1) + x[..., 0] + x[..., -1]
    return y, torch.log(E / (2 * (x.size(-1) - 1))).unsqueeze(-1)


def unframe_with_window(y: torch.Tensor, window: torch.Tensor, frame_period: int, center: bool,
                        out_length: int | None) -> torch.Tensor:
    """unframe.py:164-211 with the window a Parameter: overlap-add(y * w) / (overlap-add(w * w) + 1e-16)."""
    if y.dim() <= 1:
        raise ValueError("Input must be at least 2D tensor.")
    N, L = y.shape[-2:]
    lead = y.shape[:-2]
    if out_length is None and center:
        out_length = N * frame_period
    full = (N - 1) * frame_period + L

    def fold(fr):   # (B, N, L) -> (B, full): F.fold is the adjoint of unfold
        return F.fold(fr.transpose(-2, -1), (1, full), (1, L), stride=(1, frame_period))[:, 0, 0]

    yw = (y * window).reshape(-1, N, L)
    num = fold(yw)
    den = fold((window * window).expand(1, N, L))
    x = num / (den + 1e-16)
    s = L // 2 if center else 0
    e = None if out_length is None else s + out_length
    return x[:, s:e].reshape(*lead, -1)
