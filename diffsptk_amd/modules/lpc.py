"""LPC analysis = Levinson-Durbin of the autocorrelation (reference: diffsptk/modules/lpc.py)."""
from __future__ import annotations

import torch

from .. import ops
from ..utils.private import filter_values
from .acorr import Autocorrelation
from .base import BaseFunctionalModule, Precomputed
from .levdur import LevinsonDurbin, default_eps


class LinearPredictiveCodingAnalysis(BaseFunctionalModule):
    """Framed waveform (..., L) -> gain and LPC coefficients (..., M+1) (lpc.py:137-139)."""

    _takes_input_size = True

    def __init__(self, frame_length: int, lpc_order: int, eps: float | None = None, device=None,
                 dtype=None) -> None:
        super().__init__()
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = LinearPredictiveCodingAnalysis._precompute(x.size(-1), *args, **kwargs, device=x.device,
                                                         dtype=x.dtype, module=False)
        return LinearPredictiveCodingAnalysis._apply_precomputed(pre, x=x)

    @staticmethod
    def _check() -> None:
        pass

    @staticmethod
    def _precompute(frame_length: int, lpc_order: int, eps: float | None, device, dtype,
                    module: bool = True) -> Precomputed:
        LinearPredictiveCodingAnalysis._check()
        Autocorrelation._check(frame_length, lpc_order)
        LevinsonDurbin._check(lpc_order, eps)
        return Precomputed(values={"frame_length": frame_length, "lpc_order": lpc_order,
                                   "eps": default_eps(eps, dtype)})

    @staticmethod
    def _forward(x: torch.Tensor, *, frame_length: int, lpc_order: int, eps: float) -> torch.Tensor:
        if x.size(-1) != frame_length:
            raise ValueError(f"Unexpected length of waveform (input {x.size(-1)} vs target {frame_length}).")
        return ops.LpcFn.apply(x, lpc_order, eps)
