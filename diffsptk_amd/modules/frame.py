"""Frame: waveform -> overlapping frames (reference: diffsptk/modules/frame.py)."""
from __future__ import annotations

import torch

from .. import ops
from ..utils.private import filter_values
from .base import BaseFunctionalModule, Precomputed


class Frame(BaseFunctionalModule):
    """x:(..., T) -> (..., N, L) with N = (T-1)//P + 1 (frame.py:120-141).

    ``center`` pads (L//2, (L-1)//2), otherwise (0, L-1); ``mode`` is the F.pad mode;
    ``zmean`` subtracts each frame's mean.  Unlike the reference, which returns a strided
    view of a padded copy, the output is materialised by one bit-exact gather kernel.
    """

    def __init__(self, frame_length: int, frame_period: int, *, center: bool = True,
                 zmean: bool = False, mode: str = "constant") -> None:
        super().__init__()
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        return Frame._apply_precomputed(Frame._precompute(*args, **kwargs), x=x)

    @staticmethod
    def _check(frame_length: int, frame_period: int) -> None:
        if frame_length <= 0:
            raise ValueError("frame_length must be positive.")
        if frame_period <= 0:
            raise ValueError("frame_period must be positive.")

    @staticmethod
    def _precompute(frame_length: int, frame_period: int, center: bool = True, zmean: bool = False,
                    mode: str = "constant") -> Precomputed:
        Frame._check(frame_length, frame_period)
        ops.pad_mode_code(mode)
        return Precomputed(values={"frame_length": frame_length, "frame_period": frame_period,
                                   "center": center, "zmean": zmean, "mode": mode})

    @staticmethod
    def _forward(x: torch.Tensor, *, frame_length: int, frame_period: int, center: bool, zmean: bool,
                 mode: str) -> torch.Tensor:
        return ops.FrameFn.apply(x, frame_length, frame_period, center, zmean, mode)
