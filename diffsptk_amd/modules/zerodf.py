"""Time-variant all-zero (FIR) filter and linear interpolation of frame-wise parameters (reference: zerodf.py,
linear_intpl.py) -- SURVEY.md section 8(f), row 4: the FIR core of the MLSA filter."""
from __future__ import annotations

import torch

from .. import ops
from ..utils.private import check_size, filter_values
from .base import BaseFunctionalModule, Precomputed


class LinearInterpolation(BaseFunctionalModule):
    """x:(..., N, D) -> (..., N P, D): frame n moves linearly to frame n + 1 over P samples, the last frame is held
    (linear_intpl.py:85-117).  Element-wise device operations."""

    def __init__(self, upsampling_factor: int) -> None:
        super().__init__()
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        return LinearInterpolation._apply_precomputed(LinearInterpolation._precompute(*args, **kwargs), x=x)

    @staticmethod
    def _check(upsampling_factor: int) -> None:
        if upsampling_factor <= 0:
            raise ValueError("The upsampling factor must be positive.")

    @staticmethod
    def _precompute(upsampling_factor: int) -> Precomputed:
        LinearInterpolation._check(upsampling_factor)
        return Precomputed(values={"upsampling_factor": upsampling_factor})

    @staticmethod
    def _forward(x: torch.Tensor, *, upsampling_factor: int) -> torch.Tensor:
        P = upsampling_factor
        if P == 1:
            return x
        d = x.dim()
        if d == 1:
            x = x.view(-1, 1)
        if x.dim() > 3:
            raise ValueError("Input must be 1D, 2D, or 3D tensor.")
        nxt = torch.cat((x[..., 1:, :], x[..., -1:, :]), dim=-2)
        w = (torch.arange(P, device=x.device, dtype=x.dtype) / P).view(P, 1)
        y = torch.lerp(x.unsqueeze(-2), nxt.unsqueeze(-2), w).reshape(*x.shape[:-2], x.size(-2) * P, x.size(-1))
        return y.view(-1) if d == 1 else y


class AllZeroDigitalFilter(BaseFunctionalModule):
    """x:(..., T), b:(..., T/P, M+1) -> y:(..., T): y[t] = sum_k h_t[k] x[t - k + zeroth_index] with the taps interpolated
    linearly between frames (zerodf.py:184-243; both of the reference's modes compute this), one kernel (csrc/mgc.hip)."""

    _takes_input_size = True

    def __init__(self, filter_order: int, frame_period: int, ignore_gain: bool = False, zeroth_index: int = 0,
                 mode: str = "direct", device=None, dtype=None) -> None:
        super().__init__()
        self.in_dim = filter_order + 1
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, x: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        check_size(b.size(-1), self.in_dim, "dimension of impulse response")
        return self._call_forward(x, b)

    @staticmethod
    def _func(x: torch.Tensor, b: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = AllZeroDigitalFilter._precompute(b.size(-1) - 1, *args, **kwargs)
        return AllZeroDigitalFilter._apply_precomputed(pre, x=x, b=b)

    @staticmethod
    def _check(filter_order: int, frame_period: int, ignore_gain: bool, zeroth_index: int) -> None:
        if filter_order < 0:
            raise ValueError("filter_order must be non-negative.")
        if frame_period <= 0:
            raise ValueError("frame_period must be positive.")
        if ignore_gain and zeroth_index not in (0, filter_order):
            raise ValueError("zeroth_index must be 0 or filter_order when ignore_gain is True.")
        if zeroth_index < 0 or zeroth_index > filter_order:
            raise ValueError("zeroth_index must be in [0, filter_order].")

    @staticmethod
    def _precompute(filter_order: int, frame_period: int, ignore_gain: bool = False, zeroth_index: int = 0,
                    mode: str = "direct", device=None, dtype=None) -> Precomputed:
        AllZeroDigitalFilter._check(filter_order, frame_period, ignore_gain, zeroth_index)
        if mode not in ("direct", "efficient"):
            raise ValueError("mode must be 'direct' or 'efficient'.")
        return Precomputed(values={"frame_period": frame_period, "ignore_gain": ignore_gain, "zeroth_index": zeroth_index})

    @staticmethod
    def _forward(x: torch.Tensor, b: torch.Tensor, *, frame_period: int, ignore_gain: bool, zeroth_index: int) -> torch.Tensor:
        check_size(x.size(-1), b.size(-2) * frame_period, "sequence length")
        return ops.zerodf(x, b, frame_period, zeroth_index, ignore_gain)
