"""Frequency transform of cepstra by a first-order all-pass (reference: freqt.py)."""
from __future__ import annotations

import torch

from .. import ops
from ..utils import tables
from ..utils.private import check_size, filter_values, to
from .base import BaseFunctionalModule, Precomputed


class FrequencyTransform(BaseFunctionalModule):
    """c:(..., M1+1) -> (..., M2+1) = c @ A, A the warping matrix built in float64 by the
    recursion of freqt.py:128-139."""

    _takes_input_size = True

    def __init__(self, in_order: int, out_order: int, alpha: float = 0, device=None, dtype=None) -> None:
        super().__init__()
        self.in_dim = in_order + 1
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, c: torch.Tensor) -> torch.Tensor:
        check_size(c.size(-1), self.in_dim, "dimension of cepstrum")
        return self._call_forward(c)

    @staticmethod
    def _func(c: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = FrequencyTransform._precompute(c.size(-1) - 1, *args, **kwargs, device=c.device, dtype=c.dtype)
        return FrequencyTransform._apply_precomputed(pre, c=c)

    @staticmethod
    def _check(in_order: int, out_order: int, alpha: float) -> None:
        if in_order < 0:
            raise ValueError("in_order must be non-negative.")
        if out_order < 0:
            raise ValueError("out_order must be non-negative.")
        if 1 <= abs(alpha):
            raise ValueError("alpha must be in (-1, 1).")

    @staticmethod
    def _precompute(in_order: int, out_order: int, alpha: float, device, dtype) -> Precomputed:
        FrequencyTransform._check(in_order, out_order, alpha)
        A = tables.freqt_matrix(in_order, out_order, alpha)
        return Precomputed(tensors={"A": to(A, device=device, dtype=dtype)})

    @staticmethod
    def _forward(c: torch.Tensor, *, A: torch.Tensor) -> torch.Tensor:
        return ops.MatmulRowsFn.apply(c, A)
