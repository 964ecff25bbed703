"""Real-valued FFT (reference: diffsptk/modules/fftr.py)."""
from __future__ import annotations

import torch

from .. import ops
from ..utils import tables
from ..utils.private import filter_values, to
from . import _learnable
from .base import BaseFunctionalModule, Precomputed

_FORMATS = {"complex": 0, "real": 1, "imaginary": 2, "amplitude": 3, "power": 4}


def fftr_format_code(out_format) -> int:
    if out_format in _FORMATS:
        return _FORMATS[out_format]
    if isinstance(out_format, int) and not isinstance(out_format, bool) and 0 <= out_format <= 4:
        return out_format
    raise ValueError(f"out_format {out_format} is not supported.")


class RealValuedFastFourierTransform(BaseFunctionalModule):
    """x:(..., L) -> rfft(x, n=fft_length) formatted as complex/real/imaginary/amplitude/power
    (fftr.py:136-151).  ``learnable=True`` (fftr.py:123-129) makes the DFT matrix a Parameter ``W`` and runs on
    stock device operators (modules/_learnable.py): a training feature off the kernels' hot path."""

    def __init__(self, fft_length: int | None, out_format: str | int = "complex", learnable: bool = False,
                 device=None, dtype=None) -> None:
        super().__init__()
        self._register_precomputed(self._precompute(**filter_values(locals())), ("W",) if learnable else False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = RealValuedFastFourierTransform._precompute(*args, **kwargs, learnable=False, device=x.device,
                                                         dtype=x.dtype)
        return RealValuedFastFourierTransform._apply_precomputed(pre, x=x)

    @staticmethod
    def _check(fft_length: int | None) -> None:
        if fft_length is not None and (fft_length <= 0 or fft_length % 2 == 1):
            raise ValueError("fft_length must be positive even.")

    @staticmethod
    def _precompute(fft_length: int | None, out_format: str | int, learnable: bool, device, dtype) -> Precomputed:
        RealValuedFastFourierTransform._check(fft_length)
        fmt = fftr_format_code(out_format)
        if learnable:
            if fft_length is None:
                raise ValueError("fft_length must be specified when learnable is True.")
            W = to(_learnable.dft_matrix(fft_length), device=device, dtype=dtype)
            return Precomputed(values={"fft_length": fft_length, "fmt": fmt}, tensors={"W": W})
        if fft_length is None:  # transform length follows the input (torch.fft.rfft(x, n=None))
            return Precomputed(values={"fft_length": None, "fmt": fmt})
        tw = to(tables.twiddle_table(fft_length), device=device, dtype=dtype)
        return Precomputed(values={"fft_length": fft_length, "fmt": fmt}, tensors={"twiddle": tw})

    @staticmethod
    def _forward(x: torch.Tensor, *, fft_length: int | None, fmt: int,
                 twiddle: torch.Tensor | None = None, W: torch.Tensor | None = None) -> torch.Tensor:
        if W is not None:
            return _learnable.rfft_with_basis(x, W, fft_length, fmt)
        if fft_length is None:
            fft_length = x.size(-1)
            if fft_length % 2 == 1:
                raise ValueError("fft_length must be positive even.")
            twiddle = to(tables.twiddle_table(fft_length), device=x.device, dtype=x.dtype)
        return ops.FftrFn.apply(x, fft_length, fmt, twiddle)
