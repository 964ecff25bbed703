"""Mel-generalized cepstrum conversion (reference: mgc2mgc.py) -- SURVEY.md section 8(f), rows 3-4.

The reference chains elementary steps (mgc2mgc.py:176-300): gamma scalings, gain (de)normalisation, the frequency
transform (a row product: the library's freqt kernel) and the generalized cepstral transformation, which it
evaluates with FFTs (mgc2mgc.py:333-361).  Here the two FFTs of that step run on the library's real-transform
kernels (the sequence is real, so the half spectrum carries everything) and the pointwise spectrum arithmetic in
between -- like the gain steps -- is element-wise device code.
"""
from __future__ import annotations

import math

import torch

from .. import ops
from ..utils import tables
from ..utils.private import check_size, filter_values, to
from .base import BaseFunctionalModule, Precomputed
from .gnorm import GeneralizedCepstrumGainNormalization as _Gnorm
from .gnorm import GeneralizedCepstrumInverseGainNormalization as _Ignorm
from .spec import device_twiddle


def gc2gc(c1: torch.Tensor, out_order: int, in_gamma: float, out_gamma: float, n_fft: int) -> torch.Tensor:
    """GeneralizedCepstrumToGeneralizedCepstrum._forward (mgc2mgc.py:333-361)."""
    tw = device_twiddle(n_fft, c1.device, c1.dtype)
    if not (torch.is_grad_enabled() and c1.requires_grad):
        y = ops.gc2gc_fused(c1, out_order, in_gamma, out_gamma, n_fft, tw)   # one launch, the spectra never leave LDS
        if y is not None:
            return y
    else:
        y = ops.gc2gc_fn(c1, out_order, in_gamma, out_gamma, n_fft, tw)      # with a graph: one launch forward, one backward
        if y is not None:
            return y
    c01 = torch.cat((torch.zeros_like(c1[..., :1]), c1[..., 1:]), dim=-1)
    C1 = ops.FftrFn.apply(c01, n_fft, 0, tw)                       # half of fft(c01, n_fft): the sequence is real
    if in_gamma == 0:
        mag, ang = torch.exp(C1.real), C1.imag                     # cexp
    else:
        z = 1 + in_gamma * C1
        mag, ang = z.abs() ** (1 / in_gamma), z.angle() / in_gamma
    if out_gamma == 0:
        C2 = torch.log(mag)                                        # clog keeps the log-magnitude only (private.py:318-319)
    else:
        # the reference forms polar(r, theta) and takes .angle() of it again (mgc2mgc.py:349-355): the phase is WRAPPED to
        # (-pi, pi] before it is scaled by out_gamma, which matters for non-integer out_gamma once |theta| > pi (strong
        # resonances, |in_gamma| < 1); cos is even, so which end of the interval the boundary maps to does not matter
        ang = torch.remainder(ang + math.pi, 2 * math.pi) - math.pi
        C2 = (mag ** out_gamma * torch.cos(ang * out_gamma) - 1) / out_gamma
    c02 = ops.IfftrFn.apply(torch.complex(C2, torch.zeros_like(C2)), n_fft, out_order + 1, tw)   # ifft(C2).real[:M2+1]
    return torch.cat((c1[..., :1], 2 * c02[..., 1:]), dim=-1)


def _gc2gc_with_scalar_steps(c, out_order, ig, og, n_fft, gnorm_before, ignorm_after, tail_mul, zeroth_mul):
    """gnorm -> gc2gc -> ignorm -> GammaMultiplication -> ZerothGammaMultiplication of mgc2mgc.py:217-300 in ONE launch (the
    scalar steps are flags of dsa_gc2gc_fwd; as separate operators each was a pass over the (..., out_order + 1) rows in memory).
    Forward only; None when a gradient is wanted or the configuration has no fused kernel."""
    if torch.is_grad_enabled() and c.requires_grad:
        return None
    flags = (1 if gnorm_before else 0) | (2 if ignorm_after else 0) | (4 if tail_mul else 0) | (8 if zeroth_mul else 0)
    return ops.gc2gc_fused(c, out_order, ig, og, n_fft, device_twiddle(n_fft, c.device, c.dtype), flags)


def _scale_tail(c: torch.Tensor, s: float) -> torch.Tensor:
    return torch.cat((c[..., :1], c[..., 1:] * s), dim=-1)


class MelGeneralizedCepstrumToMelGeneralizedCepstrum(BaseFunctionalModule):
    """mc:(..., M1+1) -> (..., M2+1) between (alpha, gamma, normalised, multiplied) representations (mgc2mgc.py)."""

    _takes_input_size = True

    def __init__(self, in_order: int, out_order: int, in_alpha: float = 0, out_alpha: float = 0, in_gamma: float = 0,
                 out_gamma: float = 0, in_norm: bool = False, out_norm: bool = False, in_mul: bool = False,
                 out_mul: bool = False, n_fft: int = 512, device=None, dtype=None) -> None:
        super().__init__()
        self.in_dim = in_order + 1
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, mc: torch.Tensor) -> torch.Tensor:
        check_size(mc.size(-1), self.in_dim, "dimension of cepstrum")
        return self._call_forward(mc)

    @staticmethod
    def _func(mc: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = MelGeneralizedCepstrumToMelGeneralizedCepstrum._precompute(mc.size(-1) - 1, *args, **kwargs, device=mc.device,
                                                                         dtype=mc.dtype)
        return MelGeneralizedCepstrumToMelGeneralizedCepstrum._apply_precomputed(pre, mc=mc)

    @staticmethod
    def _check(in_order, out_order, in_alpha, out_alpha, in_gamma, out_gamma, in_mul, n_fft) -> None:
        if in_order < 0:
            raise ValueError("in_order must be non-negative.")
        if out_order < 0:
            raise ValueError("out_order must be non-negative.")
        if 1 <= abs(in_alpha):
            raise ValueError("in_alpha must be in (-1, 1).")
        if 1 <= abs(out_alpha):
            raise ValueError("out_alpha must be in (-1, 1).")
        if 1 < abs(in_gamma):
            raise ValueError("in_gamma must be in [-1, 1].")
        if 1 < abs(out_gamma):
            raise ValueError("out_gamma must be in [-1, 1].")
        if n_fft <= max(in_order, out_order) + 1:
            raise ValueError("n_fft must be much larger than order of cepstrum.")
        if 0 == in_gamma and in_mul:
            raise ValueError("Invalid combination of in_gamma and in_mul.")

    @staticmethod
    def _precompute(in_order: int, out_order: int, in_alpha: float = 0, out_alpha: float = 0, in_gamma: float = 0,
                    out_gamma: float = 0, in_norm: bool = False, out_norm: bool = False, in_mul: bool = False,
                    out_mul: bool = False, n_fft: int = 512, device=None, dtype=None) -> Precomputed:
        MelGeneralizedCepstrumToMelGeneralizedCepstrum._check(in_order, out_order, in_alpha, out_alpha, in_gamma,
                                                              out_gamma, in_mul, n_fft)
        alpha = (out_alpha - in_alpha) / (1 - in_alpha * out_alpha)
        tens = {}
        if alpha != 0:
            tens["A"] = to(tables.freqt_matrix(in_order, out_order, alpha), device=device, dtype=dtype)
        return Precomputed(values={"cfg": (in_order, out_order, alpha, in_gamma, out_gamma, bool(in_norm), bool(out_norm),
                                           bool(in_mul), bool(out_mul), n_fft)}, tensors=tens)

    @staticmethod
    def _forward(mc: torch.Tensor, *, cfg, A: torch.Tensor | None = None) -> torch.Tensor:
        in_order, out_order, alpha, ig, og, in_norm, out_norm, in_mul, out_mul, n_fft = cfg
        gn, ign = _Gnorm._forward, _Ignorm._forward
        c = mc
        if not in_norm and in_mul:                       # ZerothGammaDivision
            c = torch.cat(((c[..., :1] - 1) / ig, c[..., 1:]), dim=-1)
        if alpha == 0:
            if in_order == out_order and ig == og:
                if not in_mul and out_mul:
                    c = _scale_tail(c, ig)
                if not in_norm and out_norm:
                    c = gn(c, gamma=ig)
                if in_norm and not out_norm:
                    c = ign(c, gamma=og)
                if in_mul and not out_mul:
                    c = _scale_tail(c, 1 / og)
            else:
                if in_mul:
                    c = _scale_tail(c, 1 / ig)
                fused = _gc2gc_with_scalar_steps(c, out_order, ig, og, n_fft, not in_norm, not out_norm, out_mul,
                                                 not out_norm and out_mul)
                if fused is not None:
                    return fused
                if not in_norm:
                    c = gn(c, gamma=ig)
                c = gc2gc(c, out_order, ig, og, n_fft)
                if not out_norm:
                    c = ign(c, gamma=og)
                if out_mul:
                    c = _scale_tail(c, og)
        else:
            if in_mul:
                c = _scale_tail(c, 1 / ig)
            if in_norm:
                c = ign(c, gamma=ig)
            c = ops.MatmulRowsFn.apply(c, A)             # FrequencyTransform: the library's row-product kernel
            if ig != og:
                fused = _gc2gc_with_scalar_steps(c, out_order, ig, og, n_fft, True, not out_norm, out_mul, not out_norm and out_mul)
                if fused is not None:
                    return fused
            if out_norm or ig != og:
                c = gn(c, gamma=ig)
            if ig != og:
                c = gc2gc(c, out_order, ig, og, n_fft)
            if not out_norm and ig != og:
                c = ign(c, gamma=og)
            if out_mul:
                c = _scale_tail(c, og)
        if not out_norm and out_mul:                     # ZerothGammaMultiplication
            c = torch.cat((c[..., :1] * og + 1, c[..., 1:]), dim=-1)
        return c
