"""Mel-generalized cepstrum -> spectrum (reference: mgc2sp.py) -- SURVEY.md section 8(f), row 4."""
from __future__ import annotations

import math

import torch

from .. import ops
from ..utils import tables
from ..utils.private import check_size, filter_values, to
from .base import BaseFunctionalModule, Precomputed
from .mgc2mgc import MelGeneralizedCepstrumToMelGeneralizedCepstrum as _Mgc2mgc
from .spec import device_twiddle

_FORMATS = {"db": 0, "log-magnitude": 1, "magnitude": 2, "power": 3, "cycle": 4, "radian": 5, "degree": 6, "complex": 7}


class MelGeneralizedCepstrumToSpectrum(BaseFunctionalModule):
    """mc:(..., M+1) -> (..., L/2+1): the cepstrum of order L/2 (mgc2mgc to alpha = gamma = 0) transformed by the
    library's real-FFT kernel; the formatter works on log-magnitude (real part) / phase (imaginary part), mgc2sp.py:152-202."""

    _takes_input_size = True

    def __init__(self, cep_order: int, fft_length: int, *, alpha: float = 0, gamma: float = 0, norm: bool = False,
                 mul: bool = False, n_fft: int = 512, out_format: str | int = "power", device=None, dtype=None) -> None:
        super().__init__()
        self.in_dim = cep_order + 1
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, mc: torch.Tensor) -> torch.Tensor:
        check_size(mc.size(-1), self.in_dim, "dimension of cepstrum")
        return self._call_forward(mc)

    @staticmethod
    def _func(mc: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = MelGeneralizedCepstrumToSpectrum._precompute(mc.size(-1) - 1, *args, **kwargs, device=mc.device, dtype=mc.dtype)
        return MelGeneralizedCepstrumToSpectrum._apply_precomputed(pre, mc=mc)

    @staticmethod
    def _check() -> None:
        pass

    @staticmethod
    def _precompute(cep_order: int, fft_length: int, alpha: float = 0, gamma: float = 0, norm: bool = False,
                    mul: bool = False, n_fft: int = 512, out_format: str | int = "power", device=None,
                    dtype=None) -> Precomputed:
        if out_format in _FORMATS:
            fmt = _FORMATS[out_format]
        elif isinstance(out_format, int) and not isinstance(out_format, bool) and 0 <= out_format <= 6:
            fmt = out_format
        else:
            raise ValueError(f"out_format {out_format} is not supported.")
        pre = _Mgc2mgc._precompute(cep_order, fft_length // 2, in_alpha=alpha, out_alpha=0, in_gamma=gamma, out_gamma=0,
                                   in_norm=norm, out_norm=False, in_mul=mul, out_mul=False, n_fft=n_fft, device=device,
                                   dtype=dtype)
        tens = dict(pre.tensors)
        if gamma == 0 and not norm and not mul and cep_order <= fft_length // 2:
            # the conversion is linear here (frequency transform to order L/2, then the real transform): both folded into one
            # (M+1, L/2+1) matrix per part, so the spectrum is ONE row product on the matrix cores + the formatter
            W_re, W_im = tables.cepstrum_to_spectrum_matrices(cep_order, fft_length, pre.values["cfg"][2])   # the warp alpha -> 0
            if fmt <= 3 or fmt == 7:
                tens["W_re"] = to(W_re, device=device, dtype=dtype)
            if fmt >= 4:
                tens["W_im"] = to(W_im, device=device, dtype=dtype)
            tens.pop("A", None)   # the frequency-transform matrix is folded into W_re / W_im (equal gammas: the reference's
                                  # chain is the frequency transform alone, mgc2mgc.py:263-279 -- no n_fft-point step to alias)
        return Precomputed(values={"fmt": fmt, "fft_length": fft_length, "cfg": pre.values["cfg"]}, tensors=tens)

    @staticmethod
    def _forward(mc: torch.Tensor, *, fmt: int, fft_length: int, cfg, A: torch.Tensor | None = None,
                 W_re: torch.Tensor | None = None, W_im: torch.Tensor | None = None) -> torch.Tensor:
        if W_re is not None or W_im is not None:
            re = ops.MatmulRowsFn.apply(mc, W_re) if W_re is not None else None
            im = ops.MatmulRowsFn.apply(mc, W_im) if W_im is not None else None
            return MelGeneralizedCepstrumToSpectrum._format(fmt, re, im)
        c = _Mgc2mgc._forward(mc, cfg=cfg, A=A)
        n = (c.size(-1) - 1) * 2
        if fmt <= 3:   # formats of the log-magnitude: the real part alone
            re = ops.FftrFn.apply(c, n, 1, device_twiddle(n, c.device, c.dtype))
            if fmt == 0:
                return re * (20 / math.log(10))
            if fmt == 1:
                return re
            return torch.exp(re) if fmt == 2 else torch.exp(2 * re)
        if fmt <= 6:   # formats of the phase: the imaginary part alone
            im = ops.FftrFn.apply(c, n, 2, device_twiddle(n, c.device, c.dtype))
            return im / math.pi if fmt == 4 else (im if fmt == 5 else im * (180 / math.pi))
        sp = ops.FftrFn.apply(c, n, 0, device_twiddle(n, c.device, c.dtype))
        return torch.polar(torch.exp(sp.real), sp.imag)

    @staticmethod
    def _format(fmt: int, re: torch.Tensor | None, im: torch.Tensor | None) -> torch.Tensor:
        """mgc2sp.py:193-202 on the log-magnitude `re` / the phase `im`."""
        if fmt == 0:
            return re * (20 / math.log(10))
        if fmt == 1:
            return re
        if fmt == 2:
            return torch.exp(re)
        if fmt == 3:
            return torch.exp(2 * re)
        if fmt == 4:
            return im / math.pi
        if fmt == 5:
            return im
        if fmt == 6:
            return im * (180 / math.pi)
        return torch.polar(torch.exp(re), im)
