"""Short-time Fourier transform = Frame -> Window -> Spectrum, fused into ONE kernel launch
(reference: diffsptk/modules/stft.py, where it is a cascade of three sub-modules)."""
from __future__ import annotations

import torch

from .. import _lib, ops
from ..utils import tables
from ..utils.private import filter_values, to
from . import _learnable
from .base import BaseFunctionalModule, Precomputed
from .frame import Frame
from .spec import Spectrum, spec_format_code
from .window import Window

LEARNABLES = ("basis", "window")


class ShortTimeFourierTransform(BaseFunctionalModule):
    """x:(..., T) -> (..., N, fft_length//2+1) spectrogram (stft.py:237-241).

    Same options as the reference (stft.py:86-104): framing (center / zmean / mode), window
    (type / norm / symmetric), spectrum (eps / relative_floor / out_format incl. "complex").
    ``learnable`` may contain "window" (the table becomes a Parameter, its gradient is computed
    by the backward kernel) and "basis" (stft.py:179-184: the DFT matrix becomes a Parameter ``W``; framing and
    windowing stay on the kernels, the transform runs on stock device operators, modules/_learnable.py).
    """

    # stft.py:186-235: the reference keeps the learnable window in its Window layer and the learnable DFT matrix in spec.fftr
    _reference_state_keys = {"window": ("window.window", None), "W": ("spec.fftr.W", None)}

    def __init__(self, frame_length: int, frame_period: int, fft_length: int, *, center: bool = True,
                 zmean: bool = False, mode: str = "constant", window: str | int = "blackman",
                 norm: str | int = "power", symmetric: bool = True, eps: float = 1e-9,
                 relative_floor: float | None = None, out_format: str | int = "power",
                 learnable: bool | list[str] = False, device=None, dtype=None) -> None:
        super().__init__()
        pre = self._precompute(**filter_values(locals()))
        learn_window = learnable is True or (not isinstance(learnable, bool) and "window" in learnable)
        learn_basis = learnable is True or (not isinstance(learnable, bool) and "basis" in learnable)
        names = (("window",) if learn_window else ()) + (("W",) if learn_basis else ())
        self._register_precomputed(pre, names if names else False)
        if out_format == "complex":   # stft.py:211-222: with complex output the reference's `spec` layer IS the transform
            self._reference_state_keys = {"window": ("window.window", None), "W": ("spec.W", None)}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = ShortTimeFourierTransform._precompute(*args, **kwargs, learnable=False, device=x.device,
                                                    dtype=x.dtype, module=False)
        return ShortTimeFourierTransform._apply_precomputed(pre, x=x)

    @staticmethod
    def _check(learnable: bool | list[str]) -> None:
        if isinstance(learnable, (tuple, list)):
            if any(x not in LEARNABLES for x in learnable):
                raise ValueError("An unsupported key is found in learnable.")
        elif not isinstance(learnable, bool):
            raise ValueError("learnable must be boolean or list.")

    @staticmethod
    def _precompute(frame_length: int, frame_period: int, fft_length: int, center: bool, zmean: bool,
                    mode: str, window: str | int, norm: str | int, symmetric: bool, eps: float,
                    relative_floor: float | None, out_format: str | int, learnable: bool | list[str],
                    device, dtype, module: bool = True) -> Precomputed:
        ShortTimeFourierTransform._check(learnable)
        learn_basis = learnable is True or (not isinstance(learnable, bool) and "basis" in learnable)
        # same validation as the three sub-modules of the reference cascade
        Frame._check(frame_length, frame_period)
        ops.pad_mode_code(mode)
        Window._check(frame_length, fft_length)
        if out_format == "complex":
            fmt = 4
            if fft_length <= 0 or fft_length % 2 == 1:
                raise ValueError("fft_length must be positive even.")
        else:
            Spectrum._check(fft_length, eps, relative_floor)
            if fft_length % 2 == 1:
                raise ValueError("fft_length must be positive even.")
            fmt = spec_format_code(out_format)
        w = tables.window_table(frame_length, window, norm, symmetric)
        tens = {"window": to(w, device=device, dtype=dtype),
                "twiddle": to(tables.twiddle_table(fft_length), device=device, dtype=dtype)}
        if learn_basis:
            tens["W"] = to(_learnable.dft_matrix(fft_length), device=device, dtype=dtype)
        return Precomputed(
            values={"frame_length": frame_length, "frame_period": frame_period, "fft_length": fft_length,
                    "center": center, "zmean": zmean, "mode": mode, "eps": eps,
                    "relative_floor": relative_floor, "fmt": fmt},
            tensors=tens,
        )

    @staticmethod
    def _forward(x: torch.Tensor, *, frame_length: int, frame_period: int, fft_length: int, center: bool,
                 zmean: bool, mode: str, eps: float, relative_floor: float | None, fmt: int,
                 window: torch.Tensor, twiddle: torch.Tensor, W: torch.Tensor | None = None) -> torch.Tensor:
        if W is not None:   # learnable basis: Frame and Window kernels, then the transform against W (stft.py:237-241)
            fr = ops.WindowFn.apply(ops.FrameFn.apply(x, frame_length, frame_period, center, zmean, mode), window, fft_length)
            if fmt == 4:
                return _learnable.rfft_with_basis(fr, W, fft_length, 0)
            return _learnable.spectrum_with_basis(fr, None, W, fft_length, eps, relative_floor, fmt)
        return ops.StftFn.apply(x, window, twiddle, frame_length, frame_period, fft_length, center, zmean,
                                mode, eps, relative_floor, fmt, _lib.ALGO_AUTO)
