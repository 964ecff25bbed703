"""Inverse FFT of a half spectrum (reference: ifftr.py) -- SURVEY.md section 8(f), row 2."""
from __future__ import annotations

import torch

from .. import ops
from ..utils import tables
from ..utils.private import check_size, filter_values, to
from . import _learnable
from .base import BaseFunctionalModule, Precomputed


class RealValuedInverseFastFourierTransform(BaseFunctionalModule):
    """y:(..., L/2+1) complex -> x:(..., out_length) real = irfft(y)[..., :out_length] (ifftr.py:131-142).
    Computed as the adjoint of the forward DFT kernel applied to c_k / L * y (csrc/stft.hip)."""

    _takes_input_size = True

    def __init__(self, fft_length: int, out_length: int | None = None, learnable: bool = False, device=None,
                 dtype=None) -> None:
        super().__init__()
        self.in_dim = fft_length // 2 + 1
        self._register_precomputed(self._precompute(**filter_values(locals())), ("W",) if learnable else False)

    def forward(self, y: torch.Tensor) -> torch.Tensor:
        check_size(y.size(-1), self.in_dim, "length of spectrum")
        return self._call_forward(y)

    @staticmethod
    def _func(y: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = RealValuedInverseFastFourierTransform._precompute(2 * y.size(-1) - 2, *args, **kwargs, device=y.device,
                                                                dtype=ops._real_dtype(y))
        return RealValuedInverseFastFourierTransform._apply_precomputed(pre, y=y)

    @staticmethod
    def _check(fft_length: int, out_length: int | None) -> None:
        if fft_length <= 0 or fft_length % 2 == 1:
            raise ValueError("fft_length must be positive even.")
        if out_length is not None and (out_length <= 0 or fft_length < out_length):
            raise ValueError("out_length must be in [1, fft_length].")

    @staticmethod
    def _precompute(fft_length: int, out_length: int | None = None, learnable: bool = False, device=None,
                    dtype=None) -> Precomputed:
        RealValuedInverseFastFourierTransform._check(fft_length, out_length)
        if learnable:   # ifftr.py:125-129: the inverse DFT matrix becomes a Parameter (torch operators, _learnable.py)
            return Precomputed(values={"fft_length": fft_length, "out_length": out_length or fft_length},
                               tensors={"W": to(_learnable.idft_matrix(fft_length, out_length), device=device, dtype=dtype)})
        return Precomputed(values={"fft_length": fft_length, "out_length": out_length or fft_length},
                           tensors={"twiddle": to(tables.twiddle_table(fft_length), device=device, dtype=dtype)})

    @staticmethod
    def _forward(y: torch.Tensor, *, fft_length: int, out_length: int, twiddle: torch.Tensor | None = None,
                 W: torch.Tensor | None = None) -> torch.Tensor:
        if not y.is_complex():
            raise ValueError("Input must be a complex tensor.")
        if W is not None:
            return _learnable.irfft_with_basis(y, W)
        return ops.IfftrFn.apply(y, fft_length, out_length, twiddle)
