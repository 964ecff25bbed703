"""Yule-Walker solver (reference: diffsptk/modules/levdur.py)."""
from __future__ import annotations

import torch

from .. import ops
from ..utils.private import check_size, filter_values
from .base import BaseFunctionalModule, Precomputed


def default_eps(eps: float | None, dtype) -> float:
    # levdur.py:108-109: 1e-5 for float32 modules, 0 for float64
    if eps is None:
        return 1e-5 if (dtype or torch.get_default_dtype()) == torch.float else 0.0
    return eps


class LevinsonDurbin(BaseFunctionalModule):
    """r:(..., M+1) -> [K, a_1..a_M] with (toeplitz(r[:M]) + eps I) a = -r[1:],
    K = sqrt(r[1:].a + r[0]) (levdur.py:113-127); solved by the Levinson-Durbin recursion in
    float64 instead of the reference's dense LU."""

    _takes_input_size = True

    def __init__(self, lpc_order: int, eps: float | None = None, device=None, dtype=None) -> None:
        super().__init__()
        self.in_dim = lpc_order + 1
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, r: torch.Tensor) -> torch.Tensor:
        check_size(r.size(-1), self.in_dim, "dimension of autocorrelation")
        return self._call_forward(r)

    @staticmethod
    def _func(r: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = LevinsonDurbin._precompute(r.size(-1) - 1, *args, **kwargs, device=r.device, dtype=r.dtype)
        return LevinsonDurbin._apply_precomputed(pre, r=r)

    @staticmethod
    def _check(lpc_order: int, eps: float | None) -> None:
        if lpc_order < 0:
            raise ValueError("lpc_order must be non-negative.")
        if eps is not None and eps < 0:
            raise ValueError("eps must be non-negative.")

    @staticmethod
    def _precompute(lpc_order: int, eps: float | None, device, dtype) -> Precomputed:
        LevinsonDurbin._check(lpc_order, eps)
        return Precomputed(values={"eps": default_eps(eps, dtype)})

    @staticmethod
    def _forward(r: torch.Tensor, *, eps: float) -> torch.Tensor:
        return ops.LevdurFn.apply(r, eps)
