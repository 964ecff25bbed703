"""MFCC analysis of power spectra (reference: mfcc.py) -- SURVEY.md section 8(f), row 1."""
from __future__ import annotations

import torch

from .. import ops
from ..utils import tables
from ..utils.private import filter_values, to
from . import _learnable
from .base import BaseFunctionalModule, Precomputed
from .fbank import MelFilterBankAnalysis

_FORMATS = {0: "y", "y": "y", 1: "yE", "yE": "yE", 2: "yc", "yc": "yc", 3: "ycE", "ycE": "ycE"}


class MelFrequencyCepstralCoefficientsAnalysis(BaseFunctionalModule):
    """x:(..., L/2+1) power spectrum -> MFCC (..., M) (+ C0 / energy), mfcc.py:244-256:
    amplitude-domain filter bank -> DCT-II -> first M+1 coefficients times the liftering vector.
    The DCT and the lifter are ONE (C, M+1) matrix here (composed in float64, then cast)."""

    # mfcc.py:118-121: the reference keeps the learnable filter bank in its fbank layer
    _reference_state_keys = {"H": ("fbank.H", None)}

    def __init__(self, *, fft_length: int, mfcc_order: int, n_channel: int, sample_rate: int, lifter: int = 1,
                 f_min: float = 0, f_max: float | None = None, floor: float = 1e-5, gamma: float = 0,
                 scale: str = "htk", erb_factor: float | None = None, out_format: str | int = "y",
                 learnable: bool = False, device=None, dtype=None) -> None:
        super().__init__()
        self.in_dim = fft_length // 2 + 1
        # learnable (mfcc.py:118-121): the filter bank H becomes a Parameter (the DCT / lifter matrix W stays fixed)
        self._register_precomputed(self._precompute(**filter_values(locals(), drop_keys=["learnable"])),
                                   ("H",) if learnable else False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = MelFrequencyCepstralCoefficientsAnalysis._precompute(2 * x.size(-1) - 2, *args, **kwargs,
                                                                   device=x.device, dtype=x.dtype)
        return MelFrequencyCepstralCoefficientsAnalysis._apply_precomputed(pre, x=x)

    @staticmethod
    def _check(mfcc_order: int, n_channel: int, lifter: int) -> None:
        if mfcc_order < 0:
            raise ValueError("mfcc_order must be non-negative.")
        if n_channel <= mfcc_order:
            raise ValueError("mfcc_order must be less than n_channel.")
        if lifter < 0:
            raise ValueError("lifter must be non-negative.")

    @staticmethod
    def _precompute(fft_length, mfcc_order, n_channel, sample_rate, lifter=1, f_min=0, f_max=None, floor=1e-5,
                    gamma=0, scale="htk", erb_factor=None, out_format="y", device=None, dtype=None) -> Precomputed:
        MelFrequencyCepstralCoefficientsAnalysis._check(mfcc_order, n_channel, lifter)
        if out_format not in _FORMATS:
            raise ValueError(f"out_format {out_format} is not supported.")
        MelFilterBankAnalysis._check(fft_length, n_channel, sample_rate, f_min, f_max, floor, gamma, erb_factor)
        H = tables.fbank_matrix(fft_length, n_channel, sample_rate, f_min, f_max, scale, erb_factor)
        W = tables.dct_matrix(n_channel, 2)[:, : mfcc_order + 1] * tables.mfcc_lifter(mfcc_order, lifter)[None, :]
        return Precomputed(values={"floor": floor, "gamma": gamma, "out_format": _FORMATS[out_format]},
                           tensors={"H": to(H, device=device, dtype=dtype), "W": to(W, device=device, dtype=dtype)})

    @staticmethod
    def _forward(x: torch.Tensor, *, floor: float, gamma: float, out_format: str, H: torch.Tensor,
                 W: torch.Tensor) -> torch.Tensor:
        # amplitude-domain filter bank (mfcc.py:200 use_power=False) and DCT-II x truncation x lifter in ONE launch
        if H.requires_grad:
            yb, E = _learnable.fbank_with_weights(x, H, floor, gamma, False)
            cy = torch.matmul(yb, W)
        else:
            cy, E = ops.MfccFn.apply(x, H, W, floor, gamma, False)
        c, y = cy[..., :1], cy[..., 1:]
        if out_format == "y":
            return y
        if out_format == "yE":
            return torch.cat((y, E), dim=-1)
        if out_format == "yc":
            return torch.cat((y, c), dim=-1)
        return torch.cat((y, c, E), dim=-1)
