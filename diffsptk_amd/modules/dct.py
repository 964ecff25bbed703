"""Discrete cosine transform as a fixed matrix product (reference: dct.py)."""
from __future__ import annotations

import torch

from .. import ops
from ..utils import tables
from ..utils.private import check_size, filter_values, to
from .base import BaseFunctionalModule, Precomputed


class DiscreteCosineTransform(BaseFunctionalModule):
    """x:(..., L) -> (..., L) = x @ W, W the orthonormal DCT-I..IV matrix of dct.py:99-133."""

    def __init__(self, dct_length: int, dct_type: int = 2, device=None, dtype=None) -> None:
        super().__init__()
        self.in_dim = dct_length
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        check_size(x.size(-1), self.in_dim, "dimension of input")
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = DiscreteCosineTransform._precompute(x.size(-1), *args, **kwargs, device=x.device, dtype=x.dtype)
        return DiscreteCosineTransform._apply_precomputed(pre, x=x)

    @staticmethod
    def _check(dct_length: int, dct_type: int) -> None:
        if dct_length <= 0:
            raise ValueError("dct_length must be positive.")
        if not 1 <= dct_type <= 4:
            raise ValueError("dct_type must be in [1, 4].")

    @staticmethod
    def _precompute(dct_length: int, dct_type: int = 2, device=None, dtype=None) -> Precomputed:
        DiscreteCosineTransform._check(dct_length, dct_type)
        return Precomputed(tensors={"W": to(tables.dct_matrix(dct_length, dct_type), device=device, dtype=dtype)})

    @staticmethod
    def _forward(x: torch.Tensor, *, W: torch.Tensor) -> torch.Tensor:
        return ops.MatmulRowsFn.apply(x, W)
