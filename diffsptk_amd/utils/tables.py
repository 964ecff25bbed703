"""Host-side precomputation (float64, then cast) of every table the kernels consume.

The reference builds its tables the same way -- in float64, cast to the module dtype:
window (window.py:134-183), warping matrices (freqt.py:128-139, mcep.py:276-284),
alpha_vector (mcep.py:179-181).  What is new here is the COMPOSITION of the linear stages of
the mel-cepstral Newton iteration into three dense matrices (see csrc/mcep.hip).
Pure numpy: testable without a GPU.
"""
from __future__ import annotations

import functools
import math

import numpy as np

WINDOW_NAMES = ("blackman", "hamming", "hanning", "bartlett", "trapezoidal", "rectangular", "nuttall")


def _bessel_i0(x: np.ndarray) -> np.ndarray:
    s = np.ones_like(x)
    t = np.ones_like(x)
    q = x * x / 4.0
    for k in range(1, 200):
        t = t * q / (k * k)
        s = s + t
        if np.all(t < 1e-18 * s):
            break
    return s


def window_table(L: int, window="blackman", norm="power", symmetric: bool = True) -> np.ndarray:
    """Window._precompute (window.py:122-183) in float64."""
    if isinstance(window, int) and not isinstance(window, bool) and 0 <= window < len(WINDOW_NAMES):
        window = WINDOW_NAMES[window]
    n = np.arange(L, dtype=np.float64)
    D = float(L - 1 if symmetric else L)  # cosine-sum denominator (periodic = one longer, cut)
    if L == 1:
        base = {"ph": np.zeros(1), "bart": np.ones(1)}
    else:
        base = {"ph": 2.0 * math.pi * n / D, "bart": 1.0 - np.abs(2.0 * n / D - 1.0)}
    ph, bart = base["ph"], base["bart"]
    hann = 0.5 - 0.5 * np.cos(ph) if L > 1 else np.ones(1)
    sine = np.sin(math.pi * (n + 0.5) / (L if symmetric else L + 1))
    if window == "blackman":
        w = 0.42 - 0.5 * np.cos(ph) + 0.08 * np.cos(2 * ph) if L > 1 else np.ones(1)
    elif window == "hamming":
        w = 0.54 - 0.46 * np.cos(ph) if L > 1 else np.ones(1)
    elif window == "hanning":
        w = hann
    elif window == "bartlett":
        w = bart
    elif window == "trapezoidal":
        w = np.minimum(2.0 * bart, 1.0)
    elif window == "rectangular":
        w = np.ones(L)
    elif window == "nuttall":
        size = float(L if not symmetric else L - 1)
        c1 = np.array([0.355768, -0.487396, 0.144232, -0.012604])
        c2 = np.arange(0, 8, 2, dtype=np.float64) * (math.pi / size) if size > 0 else np.zeros(4)
        w = (c1 * np.cos(np.outer(n, c2))).sum(1)
    elif window == "povey":
        w = hann ** 0.85
    elif window == "sine":
        w = sine
    elif window == "vorbis":
        w = np.sin(math.pi * 0.5 * sine ** 2)
    elif window == "kbd":
        if not symmetric:
            raise ValueError("periodic is not supported for kbd window.")
        nk = L // 2 + 1
        if nk == 1:
            kais = np.ones(1)
        else:
            r = (np.arange(nk) - (nk - 1) / 2.0) / ((nk - 1) / 2.0)
            kais = _bessel_i0(12.0 * np.sqrt(np.maximum(0.0, 1.0 - r * r))) / _bessel_i0(np.array([12.0]))
        cs = np.cumsum(kais)
        half = np.sqrt(cs[:-1] / cs[-1])
        w = np.concatenate([half, half[::-1]])
    else:
        raise ValueError(f"window {window} is not supported.")
    w = np.asarray(w, dtype=np.float64).copy()
    if norm in (0, "none"):
        pass
    elif norm in (1, "power"):
        w /= math.sqrt(float((w * w).sum()))
    elif norm in (2, "magnitude"):
        w /= float(w.sum())
    else:
        raise ValueError(f"norm {norm} is not supported.")
    return w


@functools.lru_cache(maxsize=64)
def twiddle_table(nfft: int) -> np.ndarray:
    """(nfft, 2) = (cos, -sin)(2 pi m / nfft): exp(-2 pi i m / nfft)."""
    m = np.arange(nfft, dtype=np.float64)
    a = 2.0 * math.pi * m / nfft
    return np.stack([np.cos(a), -np.sin(a)], axis=1)


def freqt_matrix(in_order: int, out_order: int, alpha: float) -> np.ndarray:
    """First-order all-pass warping matrix, returned as (in_order+1, out_order+1) so that
    out = c @ A (FrequencyTransform._precompute, freqt.py:115-139)."""
    L1, L2 = in_order + 1, out_order + 1
    A = np.zeros((L2, L1), dtype=np.float64)
    A[0, :] = alpha ** np.arange(L1, dtype=np.float64)
    if L1 > 1 and L2 > 1:
        A[1, 1:] = A[0, :-1] * (1.0 - alpha * alpha) * np.arange(1, L1, dtype=np.float64)
    for i in range(2, L2):
        prev, cur = A[i - 1], A[i]
        # cur[j] = prev[j-1] + alpha * (cur[j-1] - prev[j]): first-order recursion along j
        t = prev[:-1] - alpha * prev[1:]
        acc = 0.0
        for j in range(1, L1):
            acc = t[j - 1] + alpha * acc
            cur[j] = acc
    return np.ascontiguousarray(A.T)


def coef_freqt_matrix(in_order: int, out_order: int, alpha: float) -> np.ndarray:
    """CoefficientsFrequencyTransform._precompute (mcep.py:264-284), (in_order+1, out_order+1)."""
    L1, L2 = in_order + 1, out_order + 1
    A = np.zeros((L2, L1), dtype=np.float64)
    A[:, 0] = (-alpha) ** np.arange(L2, dtype=np.float64)
    for i in range(1, L2):
        prev, cur = A[i - 1], A[i]
        t = prev[:-1] - alpha * prev[1:]
        acc = cur[0]
        for j in range(1, L1):
            acc = t[j - 1] + alpha * acc
            cur[j] = acc
    return np.ascontiguousarray(A.T)


@functools.lru_cache(maxsize=16)
def mcep_matrices(fft_length: int, cep_order: int, alpha: float):
    """Compose the linear stages of MelCepstralAnalysis._forward (mcep.py:189-224).

    Returns float64 (G, D, E, alpha_vector, A_freqt, A_ifreqt, A_rfreqt):
      G (H+1, M+1):  log X -> irfft -> halve c[0], c[H] -> keep [:H+1] -> freqt     (:204-207)
      D (M+1, H+1):  mc -> ifreqt -> zero-pad to nfft -> rfft -> real part           (:210-211)
      E (H+1, 2M+1): d -> irfft -> keep [:H+1] -> rfreqt                             (:214-215)
    """
    n, H, M = fft_length, fft_length // 2, cep_order
    k = np.arange(H + 1, dtype=np.float64)
    cosm = np.cos(2.0 * math.pi * np.outer(k, k) / n)  # [bin, time] symmetric
    # irfft of a REAL half spectrum (C2R ignores nothing here), restricted to time 0..H:
    #   c[t] = (1/n) (S[0] + (-1)^t S[H] + 2 sum_{0<k<H} S[k] cos(2 pi k t / n))
    IC = 2.0 * cosm / n
    IC[0, :] = 1.0 / n
    IC[H, :] = np.cos(math.pi * k) / n
    A_f = freqt_matrix(H, M, alpha)
    A_i = freqt_matrix(M, H, -alpha)
    A_r = coef_freqt_matrix(H, 2 * M, alpha)
    halve = np.ones(H + 1)
    halve[0] = 0.5
    halve[H] = 0.5
    G = (IC * halve[None, :]) @ A_f
    D = A_i @ cosm  # Re rfft of a sequence supported on 0..H: sum_t c[t] cos(2 pi k t / n)
    E = IC @ A_r
    av = (-alpha) ** np.arange(M + 1, dtype=np.float64)
    return G, D, E, av, A_f, A_i, A_r
