"""Inverse short-time Fourier transform (reference: istft.py) -- SURVEY.md section 8(f), row 2."""
from __future__ import annotations

import torch

from .. import _lib, ops
from ..utils import tables
from ..utils.private import filter_values, to
from .base import BaseFunctionalModule, Precomputed
from .ifftr import RealValuedInverseFastFourierTransform
from .unframe import Unframe


class InverseShortTimeFourierTransform(BaseFunctionalModule):
    """y:(..., T/P, N/2+1) complex spectrogram -> x:(..., T) = unframe(irfft(y)[..., :L]) (istft.py:186-193),
    ONE fused launch: the complex-cotangent STFT backward kernel is windowed inverse FFT + overlap-add."""

    def __init__(self, frame_length: int, frame_period: int, fft_length: int, *, center: bool = True,
                 window: str | int = "blackman", norm: str | int = "power", symmetric: bool = True,
                 learnable: bool | list[str] = False, device=None, dtype=None) -> None:
        super().__init__()
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, y: torch.Tensor, out_length: int | None = None) -> torch.Tensor:
        return self._call_forward(y, out_length)

    @staticmethod
    def _func(y: torch.Tensor, out_length: int | None, *args, **kwargs) -> torch.Tensor:
        pre = InverseShortTimeFourierTransform._precompute(*args, **kwargs, learnable=False, device=y.device,
                                                           dtype=ops._real_dtype(y))
        return InverseShortTimeFourierTransform._apply_precomputed(pre, y=y, out_length=out_length)

    @staticmethod
    def _check(learnable) -> None:
        if isinstance(learnable, (tuple, list)):
            if any(x not in ("basis", "window") for x in learnable):
                raise ValueError("An unsupported key is found in learnable.")
        elif not isinstance(learnable, bool):
            raise ValueError("learnable must be boolean or list.")

    @staticmethod
    def _precompute(frame_length: int, frame_period: int, fft_length: int, center: bool = True,
                    window: str | int = "blackman", norm: str | int = "power", symmetric: bool = True,
                    learnable: bool | list[str] = False, device=None, dtype=None) -> Precomputed:
        InverseShortTimeFourierTransform._check(learnable)
        if learnable:
            raise NotImplementedError("diffsptk_amd: learnable synthesis basis / window are not supported by this backend")
        RealValuedInverseFastFourierTransform._check(fft_length, frame_length)
        Unframe._check(frame_length, frame_period)
        w = tables.window_table(frame_length, window, norm, symmetric)
        return Precomputed(values={"frame_length": frame_length, "frame_period": frame_period, "fft_length": fft_length,
                                   "center": center},
                           tensors={"window": to(w, device=device, dtype=dtype),
                                    "twiddle": to(tables.twiddle_table(fft_length), device=device, dtype=dtype)})

    @staticmethod
    def _forward(y: torch.Tensor, out_length: int | None, *, frame_length: int, frame_period: int, fft_length: int,
                 center: bool, window: torch.Tensor, twiddle: torch.Tensor) -> torch.Tensor:
        if not y.is_complex():
            raise ValueError("Input must be a complex tensor.")
        return ops.IstftFn.apply(y, window, twiddle, frame_length, frame_period, fft_length, center, out_length,
                                 _lib.ALGO_AUTO)
