"""Autocorrelation of framed waveforms (reference: diffsptk/modules/acorr.py)."""
from __future__ import annotations

import torch

from .. import ops
from ..utils.private import check_size, filter_values
from .base import BaseFunctionalModule, Precomputed

_FORMATS = {"naive": 0, "normalized": 1, "biased": 2, "unbiased": 3}


class Autocorrelation(BaseFunctionalModule):
    """x:(..., L) -> r:(..., M+1), r[m] = sum_l x[l] x[l+m] (acorr.py:110-120), formatted
    naive / normalized (r/r0) / biased (r/L) / unbiased (r/(L-m))."""

    _takes_input_size = True

    def __init__(self, frame_length: int, acr_order: int, out_format: str | int = "naive") -> None:
        super().__init__()
        self.in_dim = frame_length
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        check_size(x.size(-1), self.in_dim, "length of waveform")
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = Autocorrelation._precompute(x.size(-1), *args, **kwargs)
        return Autocorrelation._apply_precomputed(pre, x=x)

    @staticmethod
    def _check(frame_length: int, acr_order: int) -> None:
        if frame_length <= 0:
            raise ValueError("frame_length must be positive.")
        if frame_length <= acr_order:
            raise ValueError("acr_order must be less than frame_length.")

    @staticmethod
    def _precompute(frame_length: int, acr_order: int, out_format: str | int = "naive") -> Precomputed:
        Autocorrelation._check(frame_length, acr_order)
        if out_format in _FORMATS:
            fmt = _FORMATS[out_format]
        elif isinstance(out_format, int) and not isinstance(out_format, bool) and 0 <= out_format <= 3:
            fmt = out_format
        else:
            raise ValueError(f"out_format {out_format} is not supported.")
        return Precomputed(values={"acr_order": acr_order, "fmt": fmt})

    @staticmethod
    def _forward(x: torch.Tensor, *, acr_order: int, fmt: int) -> torch.Tensor:
        return ops.AcorrFn.apply(x, acr_order, fmt)
