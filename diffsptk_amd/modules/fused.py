"""STFT -> mel filter bank (-> MFCC) without the spectrogram's round trip through memory -- SURVEY.md section 8(f), row 1.

The reference runs ``fbank(stft(x))`` as two modules (README.md:238-243 of the reference; stft.py:237-241 then
fbank.py:306-321 / mfcc.py:244-256), writing and re-reading the (B, N, 257) power spectrogram in between.  ``fuse``
keeps exactly those two modules and their semantics, and runs them as ONE kernel launch when the configuration is
the one the packed STFT kernel serves (csrc/stft_pk.h, ``FBM`` variants): the mel sums come out of the registers that
hold the power values, as segmented scans over the lanes.  A gradient flows back through three launches that never
need the spectrogram either (ops.StftFbankFn).  Everything else -- other sizes, options, dtypes, matrices without the
triangular structure, learnable tables -- runs the two stages on their own kernels, unchanged."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from .fbank import MelFilterBankAnalysis
from .frame import Frame
from .lpc import LinearPredictiveCodingAnalysis
from .mcep import MelCepstralAnalysis
from .mfcc import MelFrequencyCepstralCoefficientsAnalysis
from .stft import ShortTimeFourierTransform
from .window import Window

_SPEC_POWER, _FFT, _FRAME = 3, 512, 400


class FusedSTFTFilterBank(nn.Module):
    """``analysis(stft(x))`` for ``analysis`` a MelFilterBankAnalysis or a MelFrequencyCepstralCoefficientsAnalysis.

    ``last_path`` tells which route the last call took ("fused" / "two-stage") -- for tests and profiles."""

    def __init__(self, stft: ShortTimeFourierTransform, analysis: nn.Module) -> None:
        super().__init__()
        if not isinstance(stft, ShortTimeFourierTransform):
            raise ValueError("stft must be a ShortTimeFourierTransform.")
        if not isinstance(analysis, (MelFilterBankAnalysis, MelFrequencyCepstralCoefficientsAnalysis)):
            raise ValueError("analysis must be a MelFilterBankAnalysis or a MelFrequencyCepstralCoefficientsAnalysis.")
        if stft.fft_length // 2 + 1 != analysis.in_dim:
            raise ValueError("stft and analysis disagree on fft_length.")
        self.stft = stft
        self.analysis = analysis
        self.last_path = None

    def _fusable(self, x: torch.Tensor) -> bool:
        s, a = self.stft, self.analysis
        if x.dtype != torch.float32 or not x.is_cuda or x.size(-1) < 1:
            return False
        if any(isinstance(getattr(m, n, None), nn.Parameter) for m in (s, a) for n in ("window", "W", "H")):
            return False   # learnable tables run on the differentiable stages
        if not (s.fmt == _SPEC_POWER and s.mode == "constant" and not s.zmean and s.relative_floor is None
                and s.fft_length == _FFT and s.frame_length == _FRAME and s.frame_period % 2 == 0
                and 3 * s.frame_period + 512 <= 2176):
            return False
        return a.out_format == "y" and ops.fbank_scan_plan(a.H) is not None and ops.fbank_bins_table(a.H) is not None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        s, a = self.stft, self.analysis
        if not self._fusable(x):
            self.last_path = "two-stage"
            return a(s(x))
        self.last_path = "fused"
        is_mfcc = isinstance(a, MelFrequencyCepstralCoefficientsAnalysis)
        use_power = False if is_mfcc else a.use_power          # mfcc.py:200: amplitude domain
        # with a gradient needed the same launch runs as an autograd Function (backward: channel cotangents -> bins -> the
        # STFT backward kernel; ops.StftFbankFn)
        y = ops.StftFbankFn.apply(x, s.window, s.twiddle, a.H, ops.fbank_scan_plan(a.H), s.frame_length, s.frame_period,
                                  s.fft_length, s.center, s.eps, a.floor, a.gamma, use_power)
        if is_mfcc:   # DCT-II x truncation x lifter (mfcc.py:249-252): one row product on (.., C) values; C0 dropped ("y")
            y = ops.MatmulRowsFn.apply(y, a.W)[..., 1:]
        return y


class FusedSTFTMelCepstralAnalysis(nn.Module):
    """``mcep(stft(x))`` -- the BASELINE hot path -- as ONE launch (dsa_stft_mcep_fwd, csrc/mcep_mfma_f16.h: the persistent
    mel-cepstral wave computes the 16 power spectra of its tile from the waveform itself, with the packed STFT kernel's
    instructions, and keeps their logarithms in registers; 320 + 100 bytes of memory traffic per frame instead of 1348 + 1128).
    Power values, and with them the mel-cepstra, are those of the two-kernel path.  With a gradient wanted the launch also
    writes the spectrogram and the Newton history, and the backward is the two modules' own.  Other configurations
    (sizes, options, dtypes, learnable tables) run the two modules unchanged.  ``last_path``: "fused" / "two-stage"."""

    def __init__(self, stft: ShortTimeFourierTransform, analysis: MelCepstralAnalysis) -> None:
        super().__init__()
        if not isinstance(stft, ShortTimeFourierTransform):
            raise ValueError("stft must be a ShortTimeFourierTransform.")
        if not isinstance(analysis, MelCepstralAnalysis):
            raise ValueError("analysis must be a MelCepstralAnalysis.")
        if stft.fft_length // 2 + 1 != analysis.in_dim:
            raise ValueError("stft and analysis disagree on fft_length.")
        self.stft = stft
        self.analysis = analysis
        self.last_path = None

    def _fusable(self, x: torch.Tensor) -> bool:
        s, a = self.stft, self.analysis
        if any(isinstance(getattr(s, n, None), nn.Parameter) for n in ("window", "W")) or getattr(s, "W", None) is not None:
            return False
        # (round 6: every pad mode of Frame -- frame.py:130-137 --, zmean -- frame.py:139-140 -- and the relative floor -- spec.py:174-176 --
        #  run in the one launch: own instantiations of the kernel; the other output formats keep two stages)
        if not (s.fmt == _SPEC_POWER and s.mode in ("constant", "reflect", "replicate", "circular")):
            return False
        if s.mode == "reflect" and not ((s.frame_length // 2 if s.center else s.frame_length - 1) < x.size(-1) or s.frame_length == 1):
            return False   # (F.pad rejects it: let the stage raise its own error)
        if a.algo == ops._lib.ALGO_GENERIC:
            return False
        return ops.stft_mcep_fusable(x, s.window, a.G, s.frame_length, s.frame_period, s.fft_length, a.cep_order)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        s, a = self.stft, self.analysis
        if not self._fusable(x):
            self.last_path = "two-stage"
            return a(s(x))
        self.last_path = "fused"
        return ops.StftMcepFn.apply(x, s.window, s.twiddle, a.G, a.D, a.E, a.alpha_vector, s.frame_length, s.frame_period,
                                    s.fft_length, s.center, s.eps, a.cep_order, a.n_iter, s.mode, s.zmean, s.relative_floor)


class FusedFrameWindowLPC(nn.Module):
    """``lpc(window(frame(x)))`` -- the LPC branch of the reference's README (README.md:198-201; BASELINE configs[3]) -- as ONE
    launch forward and ONE launch backward (dsa_frame_window_lpc_fwd / _bwd, csrc/lpc.hip): the framed and windowed (B N, L) tensors
    of the module chain (two of them forward, two more backward) never exist.  Same modules, same semantics; configurations the
    launches do not cover (float64, other orders, learnable windows, a gradient through non-constant padding, frame lengths
    above 512 ...) run the three modules unchanged.  ``exact_lag_sums``: float64 lag sums on the vector unit instead of binary16
    splits on the matrix pipe (for near-singular frames).  ``last_path``: "fused" / "fused-forward" / "three-stage"."""

    def __init__(self, frame: Frame, window: Window, lpc: LinearPredictiveCodingAnalysis, exact_lag_sums: bool = False) -> None:
        super().__init__()
        if not isinstance(frame, Frame):
            raise ValueError("frame must be a Frame.")
        if not isinstance(window, Window):
            raise ValueError("window must be a Window.")
        if not isinstance(lpc, LinearPredictiveCodingAnalysis):
            raise ValueError("lpc must be a LinearPredictiveCodingAnalysis.")
        if window.in_dim != frame.frame_length or (window.out_length or window.in_dim) != lpc.frame_length:
            raise ValueError("frame, window and lpc disagree on frame_length.")
        self.frame, self.window, self.lpc = frame, window, lpc
        self.exact_lag_sums = bool(exact_lag_sums)
        self.last_path = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        f, wn, lp = self.frame, self.window, self.lpc
        w = wn.window
        L, P, M = f.frame_length, f.frame_period, lp.lpc_order
        plain = (x.is_cuda and x.dim() >= 1 and x.size(-1) >= 1 and not f.zmean and (wn.out_length or wn.in_dim) == wn.in_dim
                 and not isinstance(w, nn.Parameter) and w.dtype == x.dtype and w.device == x.device and M < L
                 and x.dtype in (torch.float32, torch.float64))
        want_grad = torch.is_grad_enabled() and x.requires_grad
        if plain and not want_grad:
            try:
                y = ops.frame_window_lpc(x, w, L, P, M, lp.eps, f.center, f.mode, self.exact_lag_sums)
                self.last_path = "fused-forward"
                return y
            except ops._lib.BackendError:      # a geometry the fused forward does not take (frame too long for LDS ...)
                pass
        if plain and want_grad and ops.frame_window_lpc_bwd_supported(x, L, P, M, f.center, f.mode):
            self.last_path = "fused"
            return ops.FrameWindowLpcFn.apply(x, w, L, P, M, lp.eps, f.center, self.exact_lag_sums)
        self.last_path = "three-stage"
        return lp(wn(f(x)))


def fuse(first: nn.Module, *rest: nn.Module, **options) -> nn.Module:
    """One launch where a fused kernel applies, the modules themselves elsewhere:

    * ``fuse(stft, analysis)(x) == analysis(stft(x))`` for ``analysis`` a MelCepstralAnalysis (the hot path), a
      MelFilterBankAnalysis or a MelFrequencyCepstralCoefficientsAnalysis;
    * ``fuse(frame, window, lpc)(x) == lpc(window(frame(x)))`` -- the LPC branch, forward and backward."""
    if isinstance(first, Frame):
        if len(rest) != 2:
            raise ValueError("fuse(frame, window, lpc) takes a Frame, a Window and a LinearPredictiveCodingAnalysis.")
        return FusedFrameWindowLPC(first, rest[0], rest[1], **options)
    if len(rest) != 1 or options:
        raise ValueError("fuse(stft, analysis) takes a ShortTimeFourierTransform and one analysis module.")
    analysis = rest[0]
    if isinstance(analysis, MelCepstralAnalysis):
        return FusedSTFTMelCepstralAnalysis(first, analysis)
    return FusedSTFTFilterBank(first, analysis)
