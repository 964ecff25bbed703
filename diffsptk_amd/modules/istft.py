"""Inverse short-time Fourier transform (reference: istft.py) -- SURVEY.md section 8(f), row 2."""
from __future__ import annotations

import torch

from .. import _lib, ops
from ..utils import tables
from ..utils.private import filter_values, to
from . import _learnable
from .base import BaseFunctionalModule, Precomputed
from .ifftr import RealValuedInverseFastFourierTransform
from .unframe import Unframe


class InverseShortTimeFourierTransform(BaseFunctionalModule):
    """y:(..., T/P, N/2+1) complex spectrogram -> x:(..., T) = unframe(irfft(y)[..., :L]) (istft.py:186-193),
    ONE fused launch: the complex-cotangent STFT backward kernel is windowed inverse FFT + overlap-add."""

    # istft.py:157-181: the learnable inverse-DFT matrix lives in ifftr, the learnable synthesis window in unframe as (1, L, 1)
    _reference_state_keys = {"W": ("ifftr.W", None), "window": ("unframe.window", (1, -1, 1))}

    def __init__(self, frame_length: int, frame_period: int, fft_length: int, *, center: bool = True,
                 window: str | int = "blackman", norm: str | int = "power", symmetric: bool = True,
                 learnable: bool | list[str] = False, device=None, dtype=None) -> None:
        super().__init__()
        pre = self._precompute(**filter_values(locals()))
        learn_window = learnable is True or (not isinstance(learnable, bool) and "window" in learnable)
        learn_basis = learnable is True or (not isinstance(learnable, bool) and "basis" in learnable)
        names = (("window",) if learn_window else ()) + (("W",) if learn_basis else ())
        self._register_precomputed(pre, names if names else False)

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
        RealValuedInverseFastFourierTransform._check(fft_length, frame_length)
        Unframe._check(frame_length, frame_period)
        w = tables.window_table(frame_length, window, norm, symmetric)
        tens = {"window": to(w, device=device, dtype=dtype),
                "twiddle": to(tables.twiddle_table(fft_length), device=device, dtype=dtype)}
        learn_basis = learnable is True or (not isinstance(learnable, bool) and "basis" in learnable)
        if learn_basis:
            tens["W"] = to(_learnable.idft_matrix(fft_length, frame_length), device=device, dtype=dtype)
        return Precomputed(values={"frame_length": frame_length, "frame_period": frame_period, "fft_length": fft_length,
                                   "center": center, "learn": bool(learnable)},
                           tensors=tens)

    @staticmethod
    def _forward(y: torch.Tensor, out_length: int | None, *, frame_length: int, frame_period: int, fft_length: int,
                 center: bool, window: torch.Tensor, twiddle: torch.Tensor, learn: bool = False,
                 W: torch.Tensor | None = None) -> torch.Tensor:
        if not y.is_complex():
            raise ValueError("Input must be a complex tensor.")
        if learn:   # learnable basis and / or window (istft.py:143-176): stock device operators, modules/_learnable.py
            fr = _learnable.irfft_with_basis(y, W) if W is not None else torch.fft.irfft(y, n=fft_length)[..., :frame_length]
            return _learnable.unframe_with_window(fr, window, frame_period, center, out_length)
        return ops.IstftFn.apply(y, window, twiddle, frame_length, frame_period, fft_length, center, out_length,
                                 _lib.ALGO_AUTO)
