"""Window: multiply by a window table and pad to the FFT length (reference: window.py)."""
from __future__ import annotations

import torch

from .. import ops
from ..utils import tables
from ..utils.private import check_size, filter_values, to
from .base import BaseFunctionalModule, Precomputed


class Window(BaseFunctionalModule):
    """x:(..., L1) -> (..., L2) = [x * w, 0...] (window.py:185-193).

    Window types: blackman, hamming, hanning, bartlett, trapezoidal, rectangular, nuttall
    (or their SPTK integer codes 0-6), povey, sine, vorbis, kbd; ``norm`` none/power/magnitude.
    ``learnable=True`` turns the table into a Parameter (gradient by a HIP reduction kernel).
    """

    _takes_input_size = True

    def __init__(self, in_length: int, out_length: int | None = None, *, window: str | int = "blackman",
                 norm: str | int = "power", symmetric: bool = True, learnable: bool = False,
                 device=None, dtype=None) -> None:
        super().__init__()
        self.in_dim = in_length
        self._register_precomputed(self._precompute(**filter_values(locals(), ["learnable"])), learnable)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        check_size(x.size(-1), self.in_dim, "input length")
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = Window._precompute(x.size(-1), *args, **kwargs, device=x.device, dtype=x.dtype)
        return Window._apply_precomputed(pre, x=x)

    @staticmethod
    def _check(in_length: int, out_length: int | None) -> None:
        if in_length <= 0:
            raise ValueError("in_length must be positive.")
        if out_length is not None and out_length <= 0:
            raise ValueError("out_length must be positive.")

    @staticmethod
    def _precompute(in_length: int, out_length: int | None, window: str | int, norm: str | int,
                    symmetric: bool, device, dtype) -> Precomputed:
        Window._check(in_length, out_length)
        w = tables.window_table(in_length, window, norm, symmetric)
        return Precomputed(values={"out_length": out_length},
                           tensors={"window": to(w, device=device, dtype=dtype)})

    @staticmethod
    def _forward(x: torch.Tensor, *, out_length: int | None, window: torch.Tensor) -> torch.Tensor:
        return ops.WindowFn.apply(x, window, out_length)
