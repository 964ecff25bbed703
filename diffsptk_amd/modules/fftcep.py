"""Cepstral analysis by the improved cepstral method (reference: fftcep.py) -- SURVEY.md section 8(f), row 3."""
from __future__ import annotations

import torch

from .. import ops
from ..utils import tables
from ..utils.private import check_size, filter_values, to
from .base import BaseFunctionalModule, Precomputed


class CepstralAnalysis(BaseFunctionalModule):
    """x:(..., L/2+1) power spectrum -> (..., M+1) cepstrum (fftcep.py:116-136).  Every transform of the reference
    acts on a real even sequence and is a product with one cosine matrix; one launch per direction."""

    _takes_input_size = True

    def __init__(self, *, fft_length: int, cep_order: int, accel: float = 0, n_iter: int = 0, device=None,
                 dtype=None) -> None:
        super().__init__()
        self.in_dim = fft_length // 2 + 1
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        check_size(x.size(-1), self.in_dim, "dimension of spectrum")
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = CepstralAnalysis._precompute(2 * x.size(-1) - 2, *args, **kwargs, device=x.device, dtype=x.dtype)
        return CepstralAnalysis._apply_precomputed(pre, x=x)

    @staticmethod
    def _check(fft_length: int, cep_order: int, accel: float, n_iter: int) -> None:
        if fft_length <= 1:
            raise ValueError("fft_length must be greater than 1.")
        if cep_order < 0:
            raise ValueError("cep_order must be non-negative.")
        if fft_length < 2 * cep_order:
            raise ValueError("cep_order must be less than or equal to fft_length // 2.")
        if accel < 0:
            raise ValueError("accel must be non-negative.")
        if n_iter < 0:
            raise ValueError("n_iter must be non-negative.")

    @staticmethod
    def _precompute(fft_length: int, cep_order: int, accel: float = 0, n_iter: int = 0, device=None,
                    dtype=None) -> Precomputed:
        CepstralAnalysis._check(fft_length, cep_order, accel, n_iter)
        return Precomputed(values={"cep_order": cep_order, "accel": accel, "n_iter": n_iter},
                           tensors={"A": to(tables.even_cosine_matrix(fft_length), device=device, dtype=dtype)})

    @staticmethod
    def _forward(x: torch.Tensor, *, cep_order: int, accel: float, n_iter: int, A: torch.Tensor) -> torch.Tensor:
        return ops.FftcepFn.apply(x, A, cep_order, accel, n_iter)
