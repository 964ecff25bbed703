"""Functional API: one free function per module, each delegating to ``Module._func``
(reference: diffsptk/functional.py:23,797,859,905,1659,1700,1956,2916,2963,3142)."""
from __future__ import annotations

import functools

from torch import Tensor

from . import modules as nn


def acorr(x: Tensor, acr_order: int, out_format: str | int = "naive") -> Tensor:
    """Autocorrelation of framed waveforms x:(..., L) -> (..., M+1)."""
    return nn.Autocorrelation._func(x, acr_order=acr_order, out_format=out_format)


def fftr(x: Tensor, fft_length: int | None = None, out_format: str | int = "complex") -> Tensor:
    """Real FFT x:(..., L) -> (..., fft_length/2+1)."""
    return nn.RealValuedFastFourierTransform._func(x, fft_length=fft_length, out_format=out_format)


def frame(x: Tensor, frame_length: int = 400, frame_period: int = 80, center: bool = True,
          zmean: bool = False, mode: str = "constant") -> Tensor:
    """Framing x:(..., T) -> (..., T/P, L)."""
    return nn.Frame._func(x, frame_length, frame_period, center=center, zmean=zmean, mode=mode)


def dct(x: Tensor, dct_type: int = 2) -> Tensor:
    """Discrete cosine transform x:(..., L) -> (..., L)."""
    return nn.DiscreteCosineTransform._func(x, dct_type=dct_type)


def fbank(x: Tensor, n_channel: int, sample_rate: int, f_min: float = 0, f_max: float | None = None,
          floor: float = 1e-5, gamma: float = 0, scale: str = "htk", erb_factor: float | None = None,
          use_power: bool = False, out_format: str | int = "y"):
    """Mel filter-bank analysis of power spectra x:(..., L/2+1) -> (..., C) (and log energy)."""
    return nn.MelFilterBankAnalysis._func(x, n_channel=n_channel, sample_rate=sample_rate, f_min=f_min, f_max=f_max,
                                          floor=floor, gamma=gamma, scale=scale, erb_factor=erb_factor,
                                          use_power=use_power, out_format=out_format)


def mfcc(x: Tensor, mfcc_order: int, n_channel: int, sample_rate: int, lifter: int = 1, f_min: float = 0,
         f_max: float | None = None, floor: float = 1e-5, gamma: float = 0, scale: str = "htk",
         erb_factor: float | None = None, out_format: str | int = "y") -> Tensor:
    """MFCC analysis of power spectra x:(..., L/2+1) -> (..., M) (+ C0 / energy)."""
    return nn.MelFrequencyCepstralCoefficientsAnalysis._func(
        x, mfcc_order=mfcc_order, n_channel=n_channel, sample_rate=sample_rate, lifter=lifter, f_min=f_min,
        f_max=f_max, floor=floor, gamma=gamma, scale=scale, erb_factor=erb_factor, out_format=out_format)


def freqt(c: Tensor, out_order: int, alpha: float = 0) -> Tensor:
    """Frequency transform c:(..., M1+1) -> (..., M2+1)."""
    return nn.FrequencyTransform._func(c, out_order=out_order, alpha=alpha)


def ifftr(y: Tensor, out_length: int | None = None) -> Tensor:
    """Inverse FFT of a half spectrum y:(..., L/2+1) complex -> (..., out_length) real."""
    return nn.RealValuedInverseFastFourierTransform._func(y, out_length=out_length)


def istft(y: Tensor, *, out_length: int | None = None, frame_length: int = 400, frame_period: int = 80,
          fft_length: int = 512, center: bool = True, window: str | int = "blackman", norm: str | int = "power",
          symmetric: bool = True) -> Tensor:
    """Inverse STFT y:(..., T/P, N/2+1) complex -> (..., T)."""
    return nn.InverseShortTimeFourierTransform._func(y, out_length, frame_length=frame_length, frame_period=frame_period,
                                                     fft_length=fft_length, center=center, window=window, norm=norm,
                                                     symmetric=symmetric)


def fftcep(x: Tensor, cep_order: int, accel: float = 0, n_iter: int = 0) -> Tensor:
    """Cepstral analysis of power spectra x:(..., L/2+1) -> (..., M+1)."""
    return nn.CepstralAnalysis._func(x, cep_order=cep_order, accel=accel, n_iter=n_iter)


def griffin(y: Tensor, *, out_length: int | None = None, frame_length: int = 400, frame_period: int = 80,
            fft_length: int = 512, center: bool = True, mode: str = "constant", window: str | int = "blackman",
            norm: str | int = "power", symmetric: bool = True, n_iter: int = 100, alpha: float = 0.99,
            beta: float = 0.99, gamma: float = 1.1, init_phase: str = "random", verbose: bool = False) -> Tensor:
    """Griffin-Lim phase reconstruction y:(..., T/P, N/2+1) power spectrogram -> (..., T)."""
    return nn.GriffinLim._func(y, out_length, frame_length=frame_length, frame_period=frame_period, fft_length=fft_length,
                               center=center, mode=mode, window=window, norm=norm, symmetric=symmetric, n_iter=n_iter,
                               alpha=alpha, beta=beta, gamma=gamma, init_phase=init_phase, verbose=verbose)


def unframe(y: Tensor, out_length: int | None = None, *, frame_period: int = 80, center: bool = True,
            window: str | int = "rectangular", norm: str | int = "none", symmetric: bool = True) -> Tensor:
    """Overlap-add framed waveforms y:(..., T/P, L) -> (..., T)."""
    return nn.Unframe._func(y, out_length, frame_period=frame_period, center=center, window=window, norm=norm,
                            symmetric=symmetric)


def levdur(r: Tensor, eps: float | None = None) -> Tensor:
    """Solve the Yule-Walker system r:(..., M+1) -> gain and LPC coefficients (..., M+1)."""
    return nn.LevinsonDurbin._func(r, eps=eps)


def lpc(x: Tensor, lpc_order: int, eps: float | None = None) -> Tensor:
    """LPC analysis of framed waveforms x:(..., L) -> (..., M+1)."""
    return nn.LinearPredictiveCodingAnalysis._func(x, lpc_order=lpc_order, eps=eps)


def mcep(x: Tensor, cep_order: int, alpha: float = 0, n_iter: int = 0) -> Tensor:
    """Mel-cepstral analysis of power spectra x:(..., L/2+1) -> (..., M+1)."""
    return nn.MelCepstralAnalysis._func(x, cep_order=cep_order, alpha=alpha, n_iter=n_iter)


def spec(b: Tensor | None = None, a: Tensor | None = None, *, fft_length: int = 512, eps: float = 0,
         relative_floor: float | None = None, out_format: str | int = "power") -> Tensor:
    """Spectrum of K B(z)/A(z): (..., M+1), (..., N+1) -> (..., L/2+1)."""
    return nn.Spectrum._func(b, a, fft_length=fft_length, eps=eps, relative_floor=relative_floor,
                             out_format=out_format)


def stft(x: Tensor, *, frame_length: int = 400, frame_period: int = 80, fft_length: int = 512,
         center: bool = True, zmean: bool = False, mode: str = "constant", window: str | int = "blackman",
         norm: str | int = "power", symmetric: bool = True, eps: float = 1e-9,
         relative_floor: float | None = None, out_format: str | int = "power") -> Tensor:
    """Short-time Fourier transform x:(..., T) -> (..., T/P, N/2+1)."""
    return nn.ShortTimeFourierTransform._func(
        x, frame_length=frame_length, frame_period=frame_period, fft_length=fft_length, center=center,
        zmean=zmean, mode=mode, window=window, norm=norm, symmetric=symmetric, eps=eps,
        relative_floor=relative_floor, out_format=out_format)


def window(x: Tensor, out_length: int | None = None, *, window: str | int = "blackman",
           norm: str | int = "power", symmetric: bool = True) -> Tensor:
    """Windowing x:(..., L1) -> (..., L2)."""
    return nn.Window._func(x, out_length, window=window, norm=norm, symmetric=symmetric)


# ----------------------------------------------------------------------------- SURVEY 8(f) rows 3-4
def mc2b(mc: Tensor, alpha: float = 0) -> Tensor:
    """Mel-cepstrum -> MLSA digital filter coefficients (functional.py:1936)."""
    return nn.MelCepstrumToMLSADigitalFilterCoefficients._func(mc, alpha=alpha)


def b2mc(b: Tensor, alpha: float = 0) -> Tensor:
    """MLSA digital filter coefficients -> mel-cepstrum (functional.py:86)."""
    return nn.MLSADigitalFilterCoefficientsToMelCepstrum._func(b, alpha=alpha)


def gnorm(x: Tensor, gamma: float = 0, c: int | None = None) -> Tensor:
    """Gain normalisation of a generalized cepstrum (functional.py:961)."""
    return nn.GeneralizedCepstrumGainNormalization._func(x, gamma=gamma, c=c)


def ignorm(y: Tensor, gamma: float = 0, c: int | None = None) -> Tensor:
    """Inverse gain normalisation (functional.py:1387)."""
    return nn.GeneralizedCepstrumInverseGainNormalization._func(y, gamma=gamma, c=c)


def mgc2mgc(mc: Tensor, out_order: int, in_alpha: float = 0, out_alpha: float = 0, in_gamma: float = 0,
            out_gamma: float = 0, in_norm: bool = False, out_norm: bool = False, in_mul: bool = False,
            out_mul: bool = False, n_fft: int = 512) -> Tensor:
    """Mel-generalized cepstrum conversion (functional.py:2186)."""
    return nn.MelGeneralizedCepstrumToMelGeneralizedCepstrum._func(
        mc, out_order=out_order, in_alpha=in_alpha, out_alpha=out_alpha, in_gamma=in_gamma, out_gamma=out_gamma,
        in_norm=in_norm, out_norm=out_norm, in_mul=in_mul, out_mul=out_mul, n_fft=n_fft)


def mgc2sp(mc: Tensor, fft_length: int, alpha: float = 0, gamma: float = 0, norm: bool = False, mul: bool = False,
           n_fft: int = 512, out_format: str | int = "power") -> Tensor:
    """Mel-generalized cepstrum -> spectrum (functional.py:2257)."""
    return nn.MelGeneralizedCepstrumToSpectrum._func(mc, fft_length=fft_length, alpha=alpha, gamma=gamma, norm=norm,
                                                     mul=mul, n_fft=n_fft, out_format=out_format)


def mgcep(x: Tensor, cep_order: int, alpha: float = 0, gamma: float = 0, c: int | None = None, n_iter: int = 0) -> Tensor:
    """Mel-generalized cepstral analysis of power spectra (functional.py: mgcep).  A module underneath (the reference's
    is a BaseNonFunctionalModule too): the composed matrices are cached per configuration by tables.mgcep_matrices."""
    return _mgcep_module(2 * x.size(-1) - 2, cep_order, float(alpha), float(gamma), c, n_iter, x.device, x.dtype)(x)


@functools.lru_cache(maxsize=16)
def _mgcep_module(fft_length, cep_order, alpha, gamma, c, n_iter, device, dtype):
    # one module per (configuration, device, dtype): a call does not upload the composed matrices again
    return nn.MelGeneralizedCepstralAnalysis(fft_length=fft_length, cep_order=cep_order, alpha=alpha, gamma=gamma, c=c,
                                             n_iter=n_iter, device=device, dtype=dtype)


def zerodf(x: Tensor, b: Tensor, frame_period: int = 80, ignore_gain: bool = False, zeroth_index: int = 0,
           mode: str = "direct") -> Tensor:
    """Time-variant all-zero filter (functional.py:3250-3294: zerodf)."""
    return nn.AllZeroDigitalFilter._func(x, b, frame_period=frame_period, ignore_gain=ignore_gain,
                                         zeroth_index=zeroth_index, mode=mode)


def linear_intpl(x: Tensor, upsampling_factor: int = 80) -> Tensor:
    """Linear interpolation of frame-wise parameters (functional.py: linear_intpl)."""
    return nn.LinearInterpolation._func(x, upsampling_factor=upsampling_factor)
