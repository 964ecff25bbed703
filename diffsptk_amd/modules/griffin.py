"""Griffin-Lim phase reconstruction (reference: griffin.py) -- SURVEY.md section 8(f), row 2."""
from __future__ import annotations

import math

import torch

from .. import ops
from ..utils.private import filter_values
from .base import BaseFunctionalModule, Precomputed
from .istft import InverseShortTimeFourierTransform
from .stft import ShortTimeFourierTransform


class GriffinLim(BaseFunctionalModule):
    """y:(..., T/P, N/2+1) power spectrogram -> x:(..., T) by the accelerated Griffin-Lim iteration
    (griffin.py:263-284).  One step = inverse STFT (the STFT backward kernel with inverse-transform weights) ->
    complex STFT -> ONE element-wise launch for the momentum mix, the projection c / (|c| + eps) and the next
    spectrogram sqrt(y) * angle.  When a gradient with respect to y is wanted, the same iteration runs unrolled with the
    element-wise part as complex tensor arithmetic (the reference's own formulation, griffin.py:263-292) around the
    differentiable STFT / inverse STFT kernels, so autograd differentiates through it as in the reference."""

    def __init__(self, frame_length: int, frame_period: int, fft_length: int, *, center: bool = True,
                 mode: str = "constant", window: str | int = "blackman", norm: str | int = "power",
                 symmetric: bool = True, n_iter: int = 100, alpha: float = 0.99, beta: float = 0.99,
                 gamma: float = 1.1, init_phase: str = "random", verbose: bool = False, device=None, dtype=None) -> None:
        super().__init__()
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, y: torch.Tensor, out_length: int | None = None) -> torch.Tensor:
        return self._call_forward(y, out_length)

    @staticmethod
    def _func(y: torch.Tensor, out_length: int | None, *args, **kwargs) -> torch.Tensor:
        pre = GriffinLim._precompute(*args, **kwargs, device=y.device, dtype=y.dtype)
        return GriffinLim._apply_precomputed(pre, y=y, out_length=out_length)

    @staticmethod
    def _check(n_iter: int, alpha: float, beta: float, gamma: float) -> None:
        if n_iter < 0:
            raise ValueError("n_iter must be non-negative.")
        if alpha < 0:
            raise ValueError("alpha must be non-negative.")
        if beta < 0:
            raise ValueError("beta must be non-negative.")
        if gamma < 0:
            raise ValueError("gamma must be non-negative.")

    @staticmethod
    def _precompute(frame_length: int, frame_period: int, fft_length: int, center: bool = True, mode: str = "constant",
                    window: str | int = "blackman", norm: str | int = "power", symmetric: bool = True,
                    n_iter: int = 100, alpha: float = 0.99, beta: float = 0.99, gamma: float = 1.1,
                    init_phase: str = "random", verbose: bool = False, device=None, dtype=None) -> Precomputed:
        GriffinLim._check(n_iter, alpha, beta, gamma)
        if init_phase not in ("zeros", "random"):
            raise ValueError(f"init_phase: {init_phase} is not supported.")
        stft = ShortTimeFourierTransform(frame_length, frame_period, fft_length, center=center, zmean=False, mode=mode,
                                         window=window, norm=norm, symmetric=symmetric, eps=0, relative_floor=None,
                                         out_format="complex", device=device, dtype=dtype)
        istft = InverseShortTimeFourierTransform(frame_length, frame_period, fft_length, center=center, window=window,
                                                 norm=norm, symmetric=symmetric, device=device, dtype=dtype)
        return Precomputed(values={"n_iter": n_iter, "alpha": alpha, "beta": beta, "gamma": gamma,
                                   "init_phase": init_phase, "verbose": verbose},
                           layers={"stft": stft, "istft": istft})

    @staticmethod
    def _forward(y: torch.Tensor, out_length: int | None, *, n_iter: int, alpha: float, beta: float, gamma: float,
                 init_phase: str, verbose: bool, stft, istft) -> torch.Tensor:
        eps = 1e-16
        if y.requires_grad and torch.is_grad_enabled():
            return GriffinLim._iterate_autograd(y, out_length, n_iter=n_iter, alpha=alpha, beta=beta, gamma=gamma,
                                                init_phase=init_phase, stft=stft, istft=istft, eps=eps)
        with torch.no_grad():
            return GriffinLim._iterate(y, out_length, n_iter=n_iter, alpha=alpha, beta=beta, gamma=gamma,
                                       init_phase=init_phase, verbose=verbose, stft=stft, istft=istft, eps=eps)

    @staticmethod
    def _iterate_autograd(y, out_length, *, n_iter, alpha, beta, gamma, init_phase, stft, istft, eps):
        """griffin.py:263-292 with a graph: the momentum mix and the projection as complex tensor arithmetic, the transforms on
        the differentiable kernels (same random phase draw as the graph-free path)."""
        s = torch.sqrt(y + eps)
        phase = 2 * math.pi * torch.rand_like(y.detach().contiguous()) if init_phase == "random" else torch.zeros_like(y)
        angle = torch.polar(torch.ones_like(phase), phase)
        t_prev = d_prev = None
        for n in range(n_iter):
            t = stft(istft(s * angle, out_length=out_length))[..., : s.size(-2), :]
            if n == 0:
                c = d = t
            else:
                t = (1 - gamma) * d_prev + gamma * t
                diff = t - t_prev
                c = t + alpha * diff
                d = t + beta * diff
            angle = c / (c.abs() + eps)
            t_prev, d_prev = t, d
        return istft(s * angle, out_length=out_length)

    @staticmethod
    def _iterate(y, out_length, *, n_iter, alpha, beta, gamma, init_phase, verbose, stft, istft, eps):
        yc = y.contiguous()
        phase = None
        if init_phase == "random":   # griffin.py:188-189 (the generator is torch's, as in the reference)
            phase = 2 * math.pi * torch.rand_like(yc)
        t_prev = torch.empty(*yc.shape, 2, device=yc.device, dtype=yc.dtype)
        d_prev = torch.empty_like(t_prev)
        z = ops.griffin_update(None, yc, phase, t_prev, d_prev, True, alpha, beta, gamma, eps)
        for n in range(n_iter):
            t = stft(istft(z, out_length=out_length))                 # griffin.py:269
            if verbose:   # griffin.py:286-290: |STFT| of the current waveform estimate, BEFORE the projection, against the target
                c = t[..., : yc.size(-2), :].abs()
                s = torch.sqrt(yc + eps)
                snr = -10 * torch.log10(torch.linalg.norm(c - s) / torch.linalg.norm(s))
                print(f"  iter {n + 1:5d}: SNR = {float(snr):g}")
            z = ops.griffin_update(t, yc, None, t_prev, d_prev, n == 0, alpha, beta, gamma, eps, out=z)
        return istft(z, out_length=out_length)                        # griffin.py:292
