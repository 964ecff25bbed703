"""Mel filter-bank analysis of power spectra (reference: fbank.py) -- SURVEY.md section 8(f), row 1."""
from __future__ import annotations

import torch

from .. import ops
from ..utils import tables
from ..utils.private import check_size, filter_values, to
from . import _learnable
from .base import BaseFunctionalModule, Precomputed

_FORMATS = {0: "y", "y": "y", 1: "yE", "yE": "yE", 2: "y,E", "y,E": "y,E"}


class MelFilterBankAnalysis(BaseFunctionalModule):
    """x:(..., L/2+1) power spectrum -> y:(..., C) mel filter-bank output (and E:(..., 1) log energy),
    fbank.py:306-321; the weights follow fbank.py:232-291 (tables.fbank_matrix, float64 then cast)."""

    def __init__(self, *, fft_length: int, n_channel: int, sample_rate: int, f_min: float = 0,
                 f_max: float | None = None, floor: float = 1e-5, gamma: float = 0, scale: str = "htk",
                 erb_factor: float | None = None, use_power: bool = False, out_format: str | int = "y",
                 learnable: bool = False, device=None, dtype=None) -> None:
        super().__init__()
        self.in_dim = fft_length // 2 + 1
        # learnable (fbank.py:112-122): H becomes a Parameter; the analysis then runs on stock device operators
        # (modules/_learnable.py), which also give the gradient for H
        self._register_precomputed(self._precompute(**filter_values(locals(), drop_keys=["learnable"])),
                                   learnable=bool(learnable))

    def forward(self, x: torch.Tensor):
        check_size(x.size(-1), self.in_dim, "dimension of spectrum")
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs):
        pre = MelFilterBankAnalysis._precompute(2 * x.size(-1) - 2, *args, **kwargs, device=x.device, dtype=x.dtype)
        return MelFilterBankAnalysis._apply_precomputed(pre, x=x)

    @staticmethod
    def _check(fft_length, n_channel, sample_rate, f_min, f_max, floor, gamma, erb_factor) -> None:
        if fft_length <= 1:
            raise ValueError("fft_length must be greater than 1.")
        if n_channel <= 0:
            raise ValueError("n_channel must be positive.")
        if sample_rate <= 0:
            raise ValueError("sample_rate must be positive.")
        if f_min < 0 or sample_rate / 2 <= f_min:
            raise ValueError("Invalid f_min.")
        if f_max is not None and not (f_min < f_max <= sample_rate / 2):
            raise ValueError("Invalid f_min and f_max.")
        if floor <= 0:
            raise ValueError("floor must be positive.")
        if 1 < abs(gamma):
            raise ValueError("gamma must be in [-1, 1].")
        if erb_factor is not None and erb_factor <= 0:
            raise ValueError("erb_factor must be positive.")

    @staticmethod
    def _precompute(fft_length, n_channel, sample_rate, f_min=0, f_max=None, floor=1e-5, gamma=0, scale="htk",
                    erb_factor=None, use_power=False, out_format="y", device=None, dtype=None) -> Precomputed:
        MelFilterBankAnalysis._check(fft_length, n_channel, sample_rate, f_min, f_max, floor, gamma, erb_factor)
        if out_format not in _FORMATS:
            raise ValueError(f"out_format {out_format} is not supported.")
        H = tables.fbank_matrix(fft_length, n_channel, sample_rate, f_min, f_max, scale, erb_factor)
        return Precomputed(values={"floor": floor, "gamma": gamma, "use_power": use_power,
                                   "out_format": _FORMATS[out_format]},
                           tensors={"H": to(H, device=device, dtype=dtype)})

    @staticmethod
    def _forward(x: torch.Tensor, *, floor: float, gamma: float, use_power: bool, out_format: str, H: torch.Tensor):
        if H.requires_grad:
            y, E = _learnable.fbank_with_weights(x, H, floor, gamma, use_power)
        else:
            y, E = ops.FbankFn.apply(x, H, floor, gamma, use_power)
        if out_format == "y":
            return y
        if out_format == "yE":
            return torch.cat((y, E), dim=-1)
        return y, E
