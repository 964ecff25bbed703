"""Pseudo MGLSA (mel-generalized log spectrum approximation) synthesis filter (reference: mglsadf.py) -- SURVEY.md
section 8(f), row 4.

The three modes that do not need torchlpc, for phase in {minimum, maximum, zero}:
  multi-stage   (mglsadf.py:254-386)  cepstrum of order `cep_order` (mgc2mgc), then `taylor_order` passes of the
                                      time-variant FIR kernel (csrc/mgc.hip:zerodf) summed with Taylor weights of exp;
  single-stage  (mglsadf.py:389-526)  impulse response of length `ir_length` (mgc2mgc to gamma = 1, or exp of the
                                      Hermitian transform for zero phase), then ONE pass of the FIR kernel;
  freq-domain   (mglsadf.py:529-644)  complex STFT of the excitation times the filter's complex spectrum (mgc2sp),
                                      inverse STFT -- the fused STFT / ISTFT kernels of the analysis path.
`phase="mixed"` (mglsadf.py:144-147, 240-246) takes mc:(.., N + M + 1) = c_{-N} .. c_{-1}, c_0 .. c_M with
filter_order = (N, M) (or one integer for N = M): the maximum-phase part rides on the same kernels as a second cepstrum.
`mode="pade-approx"` (recursive filter through torchlpc, absent here) raises NotImplementedError.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from ..utils.private import check_size
from .gnorm import GeneralizedCepstrumGainNormalization as _Gnorm
from .gnorm import get_gamma
from .istft import InverseShortTimeFourierTransform
from .mc2b import MelCepstrumToMLSADigitalFilterCoefficients, MLSADigitalFilterCoefficientsToMelCepstrum
from .mgc2mgc import MelGeneralizedCepstrumToMelGeneralizedCepstrum
from .mgc2sp import MelGeneralizedCepstrumToSpectrum
from .spec import device_twiddle
from .stft import ShortTimeFourierTransform
from .zerodf import LinearInterpolation


def _mirror(x: torch.Tensor, half: bool = False) -> torch.Tensor:
    x0, x1 = x[..., :1], x[..., 1:]
    if half:
        x1 = x1 * 0.5
    return torch.cat((x1.flip(-1), x0, x1), dim=-1)



def _taylor_stages(x: torch.Tensor, c: torch.Tensor, P: int, z0: int, order: int, a: torch.Tensor | None = None) -> torch.Tensor:
    """exp(F) x ~ sum_i F^i x / i!  (mglsadf.py:356-365): x <- F x / i, y <- y + x, `order` times.  Without a graph a stage
    (filter, 1 / i, running sum) is ONE launch, rounded like the three operations it replaces; with one, the differentiable
    filter and two element-wise operations per stage."""
    if a is not None:
        # learnable=True (mglsadf.py:344-349, 376-379): term i enters the sum times a[i] (a starts as ones): the differentiable
        # filter per stage, element-wise scale and sum, so that autograd also reaches a
        y = x * a[0]
        cur = x
        for i in range(1, order + 1):
            cur = ops.zerodf(cur, c, P, z0, False) * (1.0 / i)
            y = y + cur * a[i]
        return y
    if order >= 1 and ops.zerodf_taylor_shapes_ok(x, c, P) and torch.is_grad_enabled() and (x.requires_grad or c.requires_grad):
        return ops.ZerodfTaylorFn.apply(x, c, P, z0, order)      # with a graph: one launch per stage in either direction
    if order >= 1 and ops.zerodf_taylor_supported(x, c, P):
        y = x.clone(memory_format=torch.contiguous_format)
        cur = x
        for i in range(1, order + 1):
            cur, y = ops.zerodf_taylor(cur, c, P, z0, 1.0 / i, y, want_y=i < order)
        return y
    y = x
    cur = x
    for i in range(1, order + 1):
        cur = ops.zerodf(cur, c, P, z0, False) * (1.0 / i)
        y = y + cur
    return y

def _taylor_key_on_save(_module, state, prefix, _meta):   # (module-level: the module stays picklable)
    if prefix + "a" in state:
        state[prefix + "mglsadf.a"] = state.pop(prefix + "a")


def _taylor_key_on_load(_module, state, prefix, *_rest):
    if prefix + "mglsadf.a" in state:
        state[prefix + "a"] = state.pop(prefix + "mglsadf.a")


class PseudoMGLSADigitalFilter(nn.Module):
    """x:(..., T) excitation, mc:(..., T/P, M+1) mel-generalized cepstrum -> y:(..., T) (mglsadf.py:211-252)."""

    def __init__(self, filter_order: int, frame_period: int, *, alpha: float = 0, gamma: float = 0, c: int | None = None,
                 ignore_gain: bool = False, phase: str = "minimum", mode: str = "multi-stage", device=None, dtype=None,
                 **kwargs) -> None:
        super().__init__()
        # state_dict compatibility: the reference keeps the learnable Taylor coefficients in its inner filter (`mglsadf.a`, mglsadf.py:346)
        self._register_state_dict_hook(_taylor_key_on_save)
        self._register_load_state_dict_pre_hook(_taylor_key_on_load, with_module=True)
        if phase not in ("minimum", "maximum", "zero", "mixed"):
            raise ValueError(f"phase {phase} is not supported.")
        if phase != "mixed" and not isinstance(filter_order, int):
            raise ValueError("filter_order must be an integer when phase is not 'mixed'.")
        if mode == "pade-approx":
            raise NotImplementedError("diffsptk_amd: the pade-approx MLSA filter needs torchlpc and is not provided")
        if mode not in ("multi-stage", "single-stage", "freq-domain"):
            raise ValueError(f"mode {mode} is not supported.")
        gamma = get_gamma(gamma, c)
        if phase == "mixed":
            self._init_mixed(filter_order, frame_period, alpha, gamma, ignore_gain, mode, dict(device=device, dtype=dtype), kwargs)
            return
        M = filter_order
        self.filter_order, self.frame_period, self.mode, self.phase = M, frame_period, mode, phase
        self.alpha, self.gamma, self.ignore_gain = alpha, gamma, ignore_gain
        kw = dict(device=device, dtype=dtype)
        if mode == "multi-stage":
            self.taylor_order = kwargs.pop("taylor_order", 20)
            cep_order = kwargs.pop("cep_order", 199)
            n_fft = kwargs.pop("n_fft", 512)
            learnable = bool(kwargs.pop("learnable", False))
            if self.taylor_order < 0:
                raise ValueError("taylor_order must be non-negative.")
            # learnable=True: one weight per Taylor term, initialised to ones (mglsadf.py:344-349)
            self.a = nn.Parameter(torch.ones(self.taylor_order + 1, device=device, dtype=dtype)) if learnable else None
            if alpha == 0 and gamma == 0:
                cep_order = M
            self.cep_order = cep_order
            self.mgc2c = MelGeneralizedCepstrumToMelGeneralizedCepstrum(M, cep_order, in_alpha=alpha, in_gamma=gamma, n_fft=n_fft, **kw)
            self.linear_intpl = LinearInterpolation(frame_period)
        elif mode == "single-stage":
            self.ir_length = kwargs.pop("ir_length", 2000)
            self.n_fft = kwargs.pop("n_fft", 4096)
            if phase == "zero":
                self.mgc2c = MelGeneralizedCepstrumToMelGeneralizedCepstrum(M, self.ir_length - 1, in_alpha=alpha, in_gamma=gamma,
                                                                            n_fft=self.n_fft, **kw)
            else:
                self.mgc2ir = MelGeneralizedCepstrumToMelGeneralizedCepstrum(M, self.ir_length - 1, in_alpha=alpha, in_gamma=gamma,
                                                                             out_gamma=1, out_mul=True, n_fft=self.n_fft, **kw)
        else:
            frame_length = kwargs.pop("frame_length", 400)
            fft_length = kwargs.pop("fft_length", 512)
            n_fft = kwargs.pop("n_fft", 512)
            if frame_length <= 2 * frame_period:
                raise ValueError("frame_period must be less than half of frame_length.")
            if ignore_gain:
                self.mc2b = MelCepstrumToMLSADigitalFilterCoefficients(M, alpha, **kw)
                self.b2mc = MLSADigitalFilterCoefficientsToMelCepstrum(M, alpha, **kw)
            self.mgc2sp = MelGeneralizedCepstrumToSpectrum(M, fft_length, alpha=alpha, gamma=gamma, out_format="complex", n_fft=n_fft, **kw)
            self.stft = ShortTimeFourierTransform(frame_length, frame_period, fft_length, out_format="complex", **kw, **kwargs)
            self.istft = InverseShortTimeFourierTransform(frame_length, frame_period, fft_length, **kw, **kwargs)
            kwargs = {}
        if kwargs:
            raise TypeError(f"unexpected arguments for mode {mode}: {sorted(kwargs)}")

    # ---- mixed phase (mglsadf.py:144-147, 240-246): filter_order = (N, M), maximum-phase part first ----
    def _init_mixed(self, filter_order, frame_period, alpha, gamma, ignore_gain, mode, kw, kwargs) -> None:
        def pair(v):
            return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))

        N, M = pair(filter_order)
        self.orders, self.frame_period, self.mode, self.phase = (N, M), frame_period, mode, "mixed"
        self.alpha, self.gamma, self.ignore_gain = alpha, gamma, ignore_gain
        mgc = MelGeneralizedCepstrumToMelGeneralizedCepstrum
        if mode == "multi-stage":
            self.taylor_order = kwargs.pop("taylor_order", 20)
            co_max, co_min = pair(kwargs.pop("cep_order", 199))
            n_fft = kwargs.pop("n_fft", 512)
            learnable = bool(kwargs.pop("learnable", False))
            if self.taylor_order < 0:
                raise ValueError("taylor_order must be non-negative.")
            self.a = nn.Parameter(torch.ones(self.taylor_order + 1, **kw)) if learnable else None
            if alpha == 0 and gamma == 0:                                    # mglsadf.py:281-282
                co_max, co_min = N, M
            self.cep_orders = (co_max, co_min)
            self.mgc2c = nn.ModuleList([mgc(M, co_min, in_alpha=alpha, in_gamma=gamma, n_fft=n_fft, **kw),
                                        mgc(N, co_max, in_alpha=alpha, in_gamma=gamma, n_fft=n_fft, **kw)])
            self.linear_intpl = LinearInterpolation(frame_period)
        elif mode == "single-stage":
            il_max, il_min = pair(kwargs.pop("ir_length", 2000))
            self.ir_lengths, self.n_fft = (il_max, il_min), kwargs.pop("n_fft", 4096)
            if self.n_fft < il_max + il_min - 1:
                raise ValueError("n_fft must be large value.")
            self.mgc2c = nn.ModuleList([mgc(M, il_min - 1, in_alpha=alpha, in_gamma=gamma, n_fft=self.n_fft, **kw),
                                        mgc(N, il_max - 1, in_alpha=alpha, in_gamma=gamma, n_fft=self.n_fft, **kw)])
        else:
            frame_length = kwargs.pop("frame_length", 400)
            fft_length = kwargs.pop("fft_length", 512)
            n_fft = kwargs.pop("n_fft", 512)
            if frame_length <= 2 * frame_period:
                raise ValueError("frame_period must be less than half of frame_length.")
            if ignore_gain:
                self.mc2b = nn.ModuleList([MelCepstrumToMLSADigitalFilterCoefficients(o, alpha, **kw) for o in (M, N)])
                self.b2mc = nn.ModuleList([MLSADigitalFilterCoefficientsToMelCepstrum(o, alpha, **kw) for o in (M, N)])
            self.mgc2sp = nn.ModuleList([MelGeneralizedCepstrumToSpectrum(o, fft_length, alpha=alpha, gamma=gamma, out_format="complex",
                                                                          n_fft=n_fft, **kw) for o in (M, N)])
            self.stft = ShortTimeFourierTransform(frame_length, frame_period, fft_length, out_format="complex", **kw, **kwargs)
            self.istft = InverseShortTimeFourierTransform(frame_length, frame_period, fft_length, **kw, **kwargs)
            kwargs.clear()
        if kwargs:
            raise TypeError(f"unexpected arguments for mode {mode}: {sorted(kwargs)}")

    def _forward_mixed(self, x: torch.Tensor, mc: torch.Tensor) -> torch.Tensor:
        N, M = self.orders
        check_size(mc.size(-1), N + M + 1, "dimension of mel-cepstrum")
        check_size(x.size(-1), mc.size(-2) * self.frame_period, "sequence length")
        P = self.frame_period
        mc_min = mc[..., N:]
        mc_max = torch.cat((torch.zeros_like(mc[..., :1]), mc[..., :N].flip(-1)), dim=-1)   # (0, c_{-1}, .., c_{-N})
        if self.mode == "multi-stage":                                       # mglsadf.py:356-365
            c_min, c_max = self.mgc2c[0](mc_min), self.mgc2c[1](mc_max)
            c0 = c_min[..., :1] + c_max[..., :1]
            c = torch.cat((c_max[..., 1:].flip(-1), torch.zeros_like(c0), c_min[..., 1:]), dim=-1).contiguous()
            y = _taylor_stages(x, c, P, self.cep_orders[0], self.taylor_order, self.a)
            if not self.ignore_gain:
                y = y * torch.exp(self.linear_intpl(c0)).squeeze(-1)
            return y
        if self.mode == "single-stage":                                      # mglsadf.py:507-521
            il_max, il_min = self.ir_lengths
            c_min, c_max = self.mgc2c[0](mc_min), self.mgc2c[1](mc_max)
            c0 = torch.zeros_like(c_min[..., :1]) if self.ignore_gain else c_min[..., :1] + c_max[..., :1]
            c = torch.cat((c_max[..., 1:].flip(-1), c0, c_min[..., 1:]), dim=-1)
            c = torch.nn.functional.pad(c, (0, self.n_fft - c.size(-1)))
            shift = il_max - 1
            c = torch.roll(c, -shift, dims=-1).contiguous()
            # c2mpir.py:98-101 with ir_length = n_fft: ifft(exp(fft(c))).real; c is real, so the half spectrum carries it all
            tw = device_twiddle(self.n_fft, c.device, c.dtype)
            C = ops.FftrFn.apply(c, self.n_fft, 0, tw)
            h = ops.IfftrFn.apply(torch.exp(C), self.n_fft, self.n_fft, tw)
            h = torch.roll(h, shift, dims=-1)[..., : il_min + il_max - 1]
            return ops.zerodf(x, h.contiguous(), P, shift, False)
        Hs = []                                                              # mglsadf.py:617-637
        for i, c in enumerate((mc_min, mc_max)):
            if self.ignore_gain:
                b = _Gnorm._forward(self.mc2b[i](c), gamma=self.gamma)
                b = torch.cat((torch.zeros_like(b[..., :1]), b[..., 1:]), dim=-1)
                c = self.b2mc[i](b)
            Hs.append(self.mgc2sp[i](c))
        return self.istft(Hs[0] * Hs[1].conj() * self.stft(x), out_length=x.size(-1))

    def forward(self, x: torch.Tensor, mc: torch.Tensor) -> torch.Tensor:
        if self.phase == "mixed":
            return self._forward_mixed(x, mc)
        check_size(mc.size(-1), self.filter_order + 1, "dimension of mel-cepstrum")
        check_size(x.size(-1), mc.size(-2) * self.frame_period, "sequence length")
        P = self.frame_period
        if self.mode == "multi-stage":
            c = self.mgc2c(mc)
            c0 = c[..., :1]
            c = torch.cat((torch.zeros_like(c0), c[..., 1:]), dim=-1)          # remove_gain(c, value=0)
            z0 = 0
            if self.phase == "maximum":
                c, z0 = c.flip(-1), self.cep_order
            elif self.phase == "zero":
                c, z0 = _mirror(c, half=True), self.cep_order
            y = _taylor_stages(x, c, P, z0, self.taylor_order, self.a)
            if not self.ignore_gain:
                y = y * torch.exp(self.linear_intpl(c0)).squeeze(-1)
            return y
        if self.mode == "single-stage":
            L = self.ir_length
            if self.phase in ("minimum", "maximum"):
                h = self.mgc2ir(mc)
                if self.ignore_gain:
                    h = h / h[..., :1]
                z0 = 0
                if self.phase == "maximum":
                    h, z0 = h.flip(-1), L - 1
            else:
                c = self.mgc2c(mc)
                c = torch.cat((c[..., :1], 0.5 * c[..., 1:]), dim=-1)
                if self.ignore_gain:
                    c = torch.cat((torch.zeros_like(c[..., :1]), c[..., 1:]), dim=-1)
                # ifft(exp(hfft(c, n))).real[:L]: c is real, so hfft(c) = 2 Re rfft(c) - c[0] and the result is the inverse
                # real transform of a real, even spectrum -- both on the library's real-FFT kernels
                tw = device_twiddle(self.n_fft, c.device, c.dtype)
                Xh = 2 * ops.FftrFn.apply(c, self.n_fft, 1, tw) - c[..., :1]
                E = torch.exp(Xh)
                h = ops.IfftrFn.apply(torch.complex(E, torch.zeros_like(E)), self.n_fft, L, tw)
                h, z0 = _mirror(h), L - 1
            return ops.zerodf(x, h.contiguous(), P, z0, False)
        c = mc
        if self.ignore_gain:
            b = _Gnorm._forward(self.mc2b(mc), gamma=self.gamma)
            b = torch.cat((torch.zeros_like(b[..., :1]), b[..., 1:]), dim=-1)
            c = self.b2mc(b)
        H = self.mgc2sp(c)
        if self.phase == "maximum":
            H = H.conj()
        elif self.phase == "zero":
            H = H.abs()
        return self.istft(H * self.stft(x), out_length=x.size(-1))
