"""Mel-cepstrum <-> MLSA digital filter coefficients (reference: mc2b.py, b2mc.py) -- SURVEY.md section 8(f), row 4."""
from __future__ import annotations

import torch

from .. import ops
from ..utils import tables
from ..utils.private import check_size, filter_values, to
from .base import BaseFunctionalModule, Precomputed


def _check_order_alpha(cep_order: int, alpha: float) -> None:
    if cep_order < 0:
        raise ValueError("cep_order must be non-negative.")
    if 1 <= abs(alpha):
        raise ValueError("alpha must be in (-1, 1).")


class MelCepstrumToMLSADigitalFilterCoefficients(BaseFunctionalModule):
    """mc:(..., M+1) -> b:(..., M+1), b[M] = mc[M], b[m] = mc[m] - alpha b[m+1] (mc2b.py:95-99), evaluated as the row
    product mc @ A with the triangular matrix of mc2b.py:111-118 on the library's row-product kernel."""

    _takes_input_size = True

    def __init__(self, cep_order: int, alpha: float = 0, device=None, dtype=None) -> None:
        super().__init__()
        self.in_dim = cep_order + 1
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, mc: torch.Tensor) -> torch.Tensor:
        check_size(mc.size(-1), self.in_dim, "dimension of cepstrum")
        return self._call_forward(mc)

    @staticmethod
    def _func(mc: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = MelCepstrumToMLSADigitalFilterCoefficients._precompute(mc.size(-1) - 1, *args, **kwargs, device=mc.device,
                                                                     dtype=mc.dtype)
        return MelCepstrumToMLSADigitalFilterCoefficients._apply_precomputed(pre, mc=mc)

    @staticmethod
    def _check(cep_order: int, alpha: float) -> None:
        _check_order_alpha(cep_order, alpha)

    @staticmethod
    def _precompute(cep_order: int, alpha: float = 0, device=None, dtype=None) -> Precomputed:
        _check_order_alpha(cep_order, alpha)
        return Precomputed(tensors={"A": to(tables.mc2b_matrix(cep_order, alpha), device=device, dtype=dtype)})

    @staticmethod
    def _forward(mc: torch.Tensor, *, A: torch.Tensor) -> torch.Tensor:
        return ops.MatmulRowsFn.apply(mc, A)


class MLSADigitalFilterCoefficientsToMelCepstrum(BaseFunctionalModule):
    """b:(..., M+1) -> mc:(..., M+1), mc[m] = b[m] + alpha b[m+1] (b2mc.py), as a row product."""

    _takes_input_size = True

    def __init__(self, cep_order: int, alpha: float = 0, device=None, dtype=None) -> None:
        super().__init__()
        self.in_dim = cep_order + 1
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, b: torch.Tensor) -> torch.Tensor:
        check_size(b.size(-1), self.in_dim, "dimension of cepstrum")
        return self._call_forward(b)

    @staticmethod
    def _func(b: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = MLSADigitalFilterCoefficientsToMelCepstrum._precompute(b.size(-1) - 1, *args, **kwargs, device=b.device,
                                                                     dtype=b.dtype)
        return MLSADigitalFilterCoefficientsToMelCepstrum._apply_precomputed(pre, b=b)

    @staticmethod
    def _check(cep_order: int, alpha: float) -> None:
        _check_order_alpha(cep_order, alpha)

    @staticmethod
    def _precompute(cep_order: int, alpha: float = 0, device=None, dtype=None) -> Precomputed:
        _check_order_alpha(cep_order, alpha)
        return Precomputed(tensors={"A": to(tables.b2mc_matrix(cep_order, alpha), device=device, dtype=dtype)})

    @staticmethod
    def _forward(b: torch.Tensor, *, A: torch.Tensor) -> torch.Tensor:
        return ops.MatmulRowsFn.apply(b, A)
