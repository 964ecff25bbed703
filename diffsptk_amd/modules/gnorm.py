"""Gain normalisation of generalized cepstra and its inverse (reference: gnorm.py, ignorm.py) -- SURVEY.md 8(f), rows 3-4.
Element-wise on (..., M+1) rows: one launch each without a gradient (dsa_gnorm_fwd), the stock tensor operations with one."""
from __future__ import annotations

import torch

from .. import ops
from ..utils.private import check_size, filter_values
from .base import BaseFunctionalModule, Precomputed


def get_gamma(gamma: float, c: int | None) -> float:
    """utils/private.py:233-238: the stage count c, when given, sets gamma = -1 / c."""
    if c is None or c == 0:
        return gamma
    if not 1 <= c:
        raise ValueError("c must be an integer greater than or equal to 1.")
    return -1 / c


def _check(cep_order: int, gamma: float) -> None:
    if cep_order < 0:
        raise ValueError("cep_order must be non-negative.")
    if 1 < abs(gamma):
        raise ValueError("gamma must be in [-1, 1].")


class GeneralizedCepstrumGainNormalization(BaseFunctionalModule):
    """x:(..., M+1) -> (K, x1 / (1 + gamma x0)), K = (1 + gamma x0)^(1/gamma) (exp(x0) for gamma = 0): gnorm.py:102-112."""

    _takes_input_size = True

    def __init__(self, cep_order: int, gamma: float = 0, c: int | None = None) -> None:
        super().__init__()
        self.in_dim = cep_order + 1
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        check_size(x.size(-1), self.in_dim, "dimension of cepstrum")
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = GeneralizedCepstrumGainNormalization._precompute(x.size(-1) - 1, *args, **kwargs)
        return GeneralizedCepstrumGainNormalization._apply_precomputed(pre, x=x)

    @staticmethod
    def _check(cep_order: int, gamma: float) -> None:
        _check(cep_order, gamma)

    @staticmethod
    def _precompute(cep_order: int, gamma: float = 0, c: int | None = None) -> Precomputed:
        gamma = get_gamma(gamma, c)
        _check(cep_order, gamma)
        return Precomputed(values={"gamma": gamma})

    @staticmethod
    def _forward(x: torch.Tensor, *, gamma: float) -> torch.Tensor:
        if ops.gnorm_applies(x):
            return ops.gnorm(x, gamma)
        x0, x1 = torch.split(x, [1, x.size(-1) - 1], dim=-1)
        if gamma == 0:
            return torch.cat((torch.exp(x0), x1), dim=-1)
        z = 1 + gamma * x0
        return torch.cat((torch.pow(z, 1 / gamma), x1 / z), dim=-1)


class GeneralizedCepstrumInverseGainNormalization(BaseFunctionalModule):
    """y:(..., M+1) -> ((K^gamma - 1) / gamma, y1 K^gamma) (log K for gamma = 0): ignorm.py:99-109."""

    _takes_input_size = True

    def __init__(self, cep_order: int, gamma: float = 0, c: int | None = None) -> None:
        super().__init__()
        self.in_dim = cep_order + 1
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, y: torch.Tensor) -> torch.Tensor:
        check_size(y.size(-1), self.in_dim, "dimension of cepstrum")
        return self._call_forward(y)

    @staticmethod
    def _func(y: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = GeneralizedCepstrumInverseGainNormalization._precompute(y.size(-1) - 1, *args, **kwargs)
        return GeneralizedCepstrumInverseGainNormalization._apply_precomputed(pre, y=y)

    @staticmethod
    def _check(cep_order: int, gamma: float) -> None:
        _check(cep_order, gamma)

    @staticmethod
    def _precompute(cep_order: int, gamma: float = 0, c: int | None = None) -> Precomputed:
        gamma = get_gamma(gamma, c)
        _check(cep_order, gamma)
        return Precomputed(values={"gamma": gamma})

    @staticmethod
    def _forward(y: torch.Tensor, *, gamma: float) -> torch.Tensor:
        if ops.gnorm_applies(y):
            return ops.gnorm(y, gamma, inverse=True)
        K, y1 = torch.split(y, [1, y.size(-1) - 1], dim=-1)
        if gamma == 0:
            return torch.cat((torch.log(K), y1), dim=-1)
        z = torch.pow(K, gamma)
        return torch.cat(((z - 1) / gamma, y1 * z), dim=-1)
