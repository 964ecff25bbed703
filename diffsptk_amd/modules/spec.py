"""Spectrum of a rational transfer function K B(z)/A(z) (reference: diffsptk/modules/spec.py)."""
from __future__ import annotations

import torch

from .. import ops
from ..utils import tables
from ..utils.private import filter_values, to
from . import _learnable
from .base import BaseFunctionalModule, Precomputed

_FORMATS = {"db": 0, "log-magnitude": 1, "magnitude": 2, "power": 3}


def spec_format_code(out_format) -> int:
    if out_format in _FORMATS:
        return _FORMATS[out_format]
    if isinstance(out_format, int) and not isinstance(out_format, bool) and 0 <= out_format <= 3:
        return out_format
    raise ValueError(f"out_format {out_format} is not supported.")


class Spectrum(BaseFunctionalModule):
    """(b, a) -> format(max(|K B/A|^2 + eps, relative floor)) on fft_length//2+1 bins
    (spec.py:152-178); either of ``b`` (numerator) / ``a`` (gain + denominator) may be None."""

    def __init__(self, fft_length: int, *, eps: float = 0, relative_floor: float | None = None,
                 out_format: str | int = "power", learnable: bool = False) -> None:
        super().__init__()
        self._register_precomputed(self._precompute(**filter_values(locals())), ("W",) if learnable else False)

    def forward(self, b: torch.Tensor | None = None, a: torch.Tensor | None = None) -> torch.Tensor:
        return self._call_forward(b, a)

    @staticmethod
    def _func(b: torch.Tensor | None = None, a: torch.Tensor | None = None, *args, **kwargs) -> torch.Tensor:
        pre = Spectrum._precompute(*args, **kwargs, module=False)
        return Spectrum._apply_precomputed(pre, b=b, a=a)

    @staticmethod
    def _check(fft_length: int, eps: float, relative_floor: float | None) -> None:
        if fft_length <= 1:
            raise ValueError("fft_length must be greater than 1.")
        if eps < 0:
            raise ValueError("eps must be non-negative.")
        if relative_floor is not None and 0 <= relative_floor:
            raise ValueError("relative_floor must be negative.")

    @staticmethod
    def _precompute(fft_length: int, eps: float, relative_floor: float | None, out_format: str | int,
                    learnable: bool = False, module: bool = True) -> Precomputed:
        Spectrum._check(fft_length, eps, relative_floor)
        if fft_length % 2 == 1:
            raise ValueError("fft_length must be positive even.")
        if learnable:   # spec.py:133-141: the transform becomes a learnable DFT matrix (torch operators, _learnable.py)
            return Precomputed(values={"fft_length": fft_length, "eps": eps, "relative_floor": relative_floor,
                                       "fmt": spec_format_code(out_format)},
                               tensors={"W": to(_learnable.dft_matrix(fft_length), dtype=torch.get_default_dtype())})
        # the twiddle table follows the input's device/dtype at call time (Spectrum takes no
        # device/dtype argument in the reference either); cached per (device, dtype) below
        return Precomputed(values={"fft_length": fft_length, "eps": eps, "relative_floor": relative_floor,
                                   "fmt": spec_format_code(out_format)})

    @staticmethod
    def _forward(b: torch.Tensor | None, a: torch.Tensor | None, *, fft_length: int, eps: float,
                 relative_floor: float | None, fmt: int, W: torch.Tensor | None = None) -> torch.Tensor:
        if W is not None:
            return _learnable.spectrum_with_basis(b, a, W, fft_length, eps, relative_floor, fmt)
        if b is None and a is None:
            raise ValueError("Either b or a must be specified.")
        ref = b if b is not None else a
        tw = device_twiddle(fft_length, ref.device, ref.dtype)
        return ops.SpecFn.apply(b, a, fft_length, eps, relative_floor, fmt, tw)


_TW_CACHE: dict = {}


def device_twiddle(fft_length: int, device, dtype) -> torch.Tensor:
    key = (fft_length, str(device), dtype)
    tw = _TW_CACHE.get(key)
    if tw is None:
        tw = to(tables.twiddle_table(fft_length), device=device, dtype=dtype)
        _TW_CACHE[key] = tw
    return tw
