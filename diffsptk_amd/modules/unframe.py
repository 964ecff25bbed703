"""Overlap-add of framed waveforms (reference: unframe.py) -- SURVEY.md section 8(f), row 2."""
from __future__ import annotations

import torch

from .. import ops
from ..utils import tables
from ..utils.private import check_size, filter_values, to
from . import _learnable
from .base import BaseFunctionalModule, Precomputed


class Unframe(BaseFunctionalModule):
    """y:(..., T/P, L) -> x:(..., T): fold(y * w) / (fold(w * w) + 1e-16), centred frames trimmed by L//2
    (unframe.py:164-211).  Overlap-add is the adjoint of framing: it runs on the Frame backward kernel."""

    _takes_input_size = True

    # unframe.py:155-157: the reference stores its (learnable) window as (1, L, 1)
    _reference_state_keys = {"window": ("window", (1, -1, 1))}

    def __init__(self, frame_length: int, frame_period: int, *, center: bool = True, window: str | int = "rectangular",
                 norm: str | int = "none", symmetric: bool = True, learnable: bool = False, device=None,
                 dtype=None) -> None:
        super().__init__()
        self.in_dim = frame_length
        # learnable: the synthesis window becomes a Parameter and the overlap-add runs on stock device operators
        # (modules/_learnable.py) -- the kernels return no gradient for the window of the DIVISOR fold(w * w)
        self.learnable = bool(learnable)
        self._register_precomputed(self._precompute(**filter_values(locals(), drop_keys=["learnable"])),
                                   ("window",) if learnable else False)

    def forward(self, y: torch.Tensor, out_length: int | None = None) -> torch.Tensor:
        check_size(y.size(-1), self.in_dim, "length of frame")
        if self.learnable:
            return _learnable.unframe_with_window(y, self.window, self.frame_period, self.center, out_length)
        return self._call_forward(y, out_length)

    @staticmethod
    def _func(y: torch.Tensor, out_length: int | None, *args, **kwargs) -> torch.Tensor:
        pre = Unframe._precompute(y.size(-1), *args, **kwargs, device=y.device, dtype=y.dtype)
        return Unframe._apply_precomputed(pre, y=y, out_length=out_length)

    @staticmethod
    def _check(frame_length: int, frame_period: int) -> None:
        if frame_period <= 0:
            raise ValueError("frame_period must be positive.")
        if frame_length <= 0:
            raise ValueError("frame_length must be positive.")
        if frame_length < frame_period:
            raise ValueError("frame_period must be less than or equal to frame_length.")

    @staticmethod
    def _precompute(frame_length: int, frame_period: int, center: bool = True, window: str | int = "rectangular",
                    norm: str | int = "none", symmetric: bool = True, device=None, dtype=None) -> Precomputed:
        Unframe._check(frame_length, frame_period)
        w = tables.window_table(frame_length, window, norm, symmetric)
        return Precomputed(values={"frame_period": frame_period, "center": center},
                           tensors={"window": to(w, device=device, dtype=dtype)})

    @staticmethod
    def _forward(y: torch.Tensor, out_length: int | None, *, frame_period: int, center: bool,
                 window: torch.Tensor) -> torch.Tensor:
        return ops.UnframeFn.apply(y, window, frame_period, center, out_length)
