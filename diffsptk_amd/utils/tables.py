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


def cepstrum_to_spectrum_matrices(in_order: int, fft_length: int, alpha: float) -> tuple[np.ndarray, np.ndarray]:
    """(W_re, W_im), each (in_order+1, fft_length/2+1): the frequency transform of a (mel-)cepstrum to order fft_length/2
    (freqt.py:115-139; the identity with zero padding at alpha = 0) followed by the real transform of that cepstrum
    (mgc2sp.py:193-202: `fftr` of the length-(fft_length/2+1) sequence at fft_length points) as ONE matrix each for the
    real and the imaginary part:  Re = c @ W_re,  Im = c @ W_im.  Valid where mgc2mgc is linear (gamma = 0, no gain
    normalisation, no gamma multiplication)."""
    H = fft_length // 2
    if alpha != 0:
        A = freqt_matrix(in_order, H, alpha)                       # (in_order+1, H+1)
    else:
        A = np.eye(in_order + 1, H + 1, dtype=np.float64)
    ang = 2.0 * np.pi * np.outer(np.arange(H + 1, dtype=np.float64), np.arange(H + 1, dtype=np.float64)) / fft_length
    return np.ascontiguousarray(A @ np.cos(ang)), np.ascontiguousarray(-(A @ np.sin(ang)))


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


# ------------------------------------------------------------------ mel filter bank / DCT
def hz_to_auditory(f, scale: str):
    """Auditory scales of the reference (utils/private.py:241-274)."""
    f = np.asarray(f, dtype=np.float64)
    if scale == "htk":
        return 1127.0 * np.log1p(f / 700.0)
    if scale in ("oshaughnessy", "mel"):
        return 2595.0 * np.log10(1.0 + f / 700.0)
    if scale in ("chakroborty", "inverted-mel"):
        return 2195.286 - 2595.0 * np.log10(1.0 + (4031.25 - f) / 700.0)
    if scale in ("traunmuller", "bark"):
        return (26.81 * f) / (1960.0 + f) - 0.53
    if scale == "linear":
        return f
    raise ValueError(f"scale {scale} is not supported.")


def auditory_to_hz(z, scale: str):
    """Inverse of hz_to_auditory (utils/private.py:277-288)."""
    z = np.asarray(z, dtype=np.float64)
    if scale == "htk":
        return 700.0 * np.expm1(z / 1127.0)
    if scale in ("oshaughnessy", "mel"):
        return 700.0 * (np.power(10.0, z / 2595.0) - 1.0)
    if scale in ("chakroborty", "inverted-mel"):
        return 4031.25 - 700.0 * (np.power(10.0, (2195.286 - z) / 2595.0) - 1.0)
    if scale in ("traunmuller", "bark"):
        return 1960.0 * (z + 0.53) / (26.28 - z)
    if scale == "linear":
        return z
    raise ValueError(f"scale {scale} is not supported.")


def fbank_matrix(fft_length: int, n_channel: int, sample_rate: int, f_min: float = 0.0, f_max: float | None = None,
                 scale: str = "htk", erb_factor: float | None = None) -> np.ndarray:
    """Triangular filter-bank weights H (L/2+1, C) of MelFilterBankAnalysis._precompute (fbank.py:232-291)."""
    K = fft_length // 2 + 1
    if f_max is None:
        f_max = sample_rate / 2
    H = np.zeros((K, n_channel), dtype=np.float64)
    if erb_factor is None:
        z_lo = float(hz_to_auditory(f_min, scale))
        z_hi = float(hz_to_auditory(f_max, scale))
        k_lo = max(1, int(f_min / sample_rate * fft_length + 1.5))
        k_hi = min(fft_length // 2, int(f_max / sample_rate * fft_length + 0.5))
        # channel centres 1 .. C+1 on the auditory axis; centre 0 would be z_lo itself
        centres = (z_hi - z_lo) / (n_channel + 1) * np.arange(1, n_channel + 2) + z_lo
        widths = np.diff(np.concatenate(([z_lo], centres)))
        for k in range(k_lo, k_hi):
            z = float(hz_to_auditory(sample_rate * k / fft_length, scale))
            m = int(np.argmax(z <= centres))          # first centre at or above the bin
            w = (centres[m] - z) / widths[m]           # weight of the lower neighbour
            if m > 0:
                H[k, m - 1] = w
            if m < n_channel:
                H[k, m] = 1.0 - w
        return H
    a, b, c = erb_factor * 6.23e-6, erb_factor * 93.39e-3, erb_factor * 28.52

    def edge_centre(f, first):
        s = 1.0 if first else -1.0
        ah, bh, ch = s * 0.5 / (700.0 + f), s * 700.0 / (700.0 + f), -s * 0.5 * f * (1.0 + 700.0 / (700.0 + f))
        bb, cb = (b - bh) / (a - ah), (c - ch) / (a - ah)
        return 0.5 * (-bb + np.sqrt(bb * bb - 4.0 * cb))

    zc = np.linspace(float(hz_to_auditory(edge_centre(f_min, True), scale)),
                     float(hz_to_auditory(edge_centre(f_max, False), scale)), n_channel)
    fc = auditory_to_hz(zc, scale)
    erb = a * fc ** 2 + b * fc + c
    fl = -(700.0 + erb) + np.sqrt(erb ** 2 + (700.0 + fc) ** 2)
    fh = fl + 2.0 * erb
    f = np.linspace(0.0, sample_rate / 2, K)
    for m in range(n_channel):
        up = (fl[m] <= f) & (f < fc[m])
        H[up, m] = (f[up] - fl[m]) / (fc[m] - fl[m])
        dn = (fc[m] <= f) & (f <= fh[m])
        H[dn, m] = (fh[m] - f[dn]) / (fh[m] - fc[m])
    return H


def fbank_scan_plan(H: np.ndarray):
    """Per-lane plan of the fused STFT -> filter-bank kernel (csrc/stft_pk.h, `FB` variant) for a 257-bin filter-bank
    matrix H (K, C) whose rows have at most two non-zero entries, in ADJACENT channels j - 1 ("down" slope) and j ("up"
    slope), with j non-decreasing over the bins: what `fbank_matrix` builds for the mel / auditory scales
    (fbank.py:232-291).  Returns None for any other matrix (the two-kernel path serves those).

    Bin k then lies in "interval" j(k) between two channel centres, and channel c is
        sum over interval c of up(k) s(k)  +  sum over interval c + 1 of down(k) s(k)  (+ the end bins 0 and 256).
    In the kernel a lane holds the bins (2 l + 1, 2 l + 2) ("lower half") and (255 - 2 l, 254 - 2 l) ("upper half") of
    a frame, so an interval is a run of neighbouring lanes and its two sums come out of a SEGMENTED inclusive scan
    over the lanes (DPP row shifts 1, 2, 4, 8, row broadcasts 15 and 31; the `mask` entries say per lane and step
    whether the source lane belongs to the same run).  A lane whose two bins lie in different intervals closes the
    first interval itself (`isM`: its first bin + the scan value of the previous lane) and starts a new run with
    the second; `isE` marks the lanes whose scan value is the total of interval `jE`.
    tools/proto_fbank_scan.py executes this plan lane by lane in numpy against `s @ H`."""
    H = np.asarray(H, dtype=np.float64)
    if H.ndim != 2 or H.shape[0] != 257 or not (1 <= H.shape[1] <= 126) or not np.all(np.isfinite(H)):
        return None
    K, C = H.shape
    jk = np.zeros(K, dtype=np.int64)
    wd = np.zeros(K)
    wu = np.zeros(K)
    prev = 0
    for k in range(1, K - 1):
        nz = np.flatnonzero(H[k])
        if len(nz) == 0:
            j = prev
        elif len(nz) == 1:
            c = int(nz[0])
            if c >= prev:
                j, wu[k] = c, H[k, c]
            elif c + 1 >= prev:
                j, wd[k] = c + 1, H[k, c]
            else:
                return None
        elif len(nz) == 2 and nz[1] == nz[0] + 1 and nz[1] >= prev:
            j, wd[k], wu[k] = int(nz[1]), H[k, nz[0]], H[k, nz[1]]
        else:
            return None
        jk[k] = prev = j
    lane = np.arange(64)
    plan = {name: np.zeros((64, 2)) for name in ("wd0", "wd1", "wu0", "wu1", "nb", "mM")}
    plan["mask"] = np.zeros((64, 2, 6))
    for name in ("isE", "isM", "jE", "jM"):
        plan[name] = np.zeros((64, 2), dtype=np.int64)
    for h in range(2):
        b0 = 2 * lane + 1 if h == 0 else 255 - 2 * lane     # first / second bin of a lane in scan order
        b1 = 2 * lane + 2 if h == 0 else 254 - 2 * lane
        j0, j1 = jk[b0], jk[b1]
        plan["wd0"][:, h], plan["wu0"][:, h] = wd[b0], wu[b0]
        plan["wd1"][:, h], plan["wu1"][:, h] = wd[b1], wu[b1]
        if h == 1:   # bin 128 belongs to the lower half
            plan["wd1"][63, h] = plan["wu1"][63, h] = 0.0
        boundary = j0 != j1
        plan["nb"][:, h] = np.where(boundary, 0.0, 1.0)
        run = np.zeros(64, dtype=np.int64)
        for ln in range(1, 64):
            if not boundary[ln] and j1[ln - 1] == j0[ln]:
                run[ln] = run[ln - 1] + 1
        for step, d in enumerate((1, 2, 4, 8)):
            plan["mask"][:, h, step] = run >= d
        plan["mask"][:, h, 4] = ((lane // 16) % 2 == 1) & (run >= lane % 16 + 1)
        plan["mask"][:, h, 5] = (lane >= 32) & (run >= lane - 31)
        plan["mM"][1:, h] = j1[:-1] == j0[1:]
        plan["isE"][:, h] = np.append(j0[1:] != j1[:-1], True)
        plan["isM"][:, h] = boundary
        plan["jE"][:, h], plan["jM"][:, h] = j1, j0
    plan["C"] = C
    plan["h0"] = np.zeros(128)
    plan["h256"] = np.zeros(128)
    plan["h0"][:C], plan["h256"][:C] = H[0], H[K - 1]
    return plan


def fbank_scan_table(plan) -> np.ndarray:
    """The plan of `fbank_scan_plan` as the (64, 32) float32 table the kernel reads (integer fields as bit patterns):
    columns 0-7 weights (wd0, wu0, wd1, wu1) x (lower, upper), 8-9 nb, 10-21 scan masks [half][step], 22-23 mM,
    24 slot indices (jE lower | jE upper << 8 | jM lower << 16 | jM upper << 24), 25 flags (isE lower, isE upper,
    isM lower, isM upper), 26 / 27 H[0, lane], H[256, lane], 28 / 29 H[0, lane + 64], H[256, lane + 64],
    30 which of the four sums channel `lane` (bits 0-3) and channel `lane + 64` (bits 4-7) reads exist; bit 8: some channel weights bin 0 or 256."""
    t = np.zeros((64, 32), dtype=np.float32)
    for i, name in enumerate(("wd0", "wu0", "wd1", "wu1")):
        t[:, 2 * i:2 * i + 2] = plan[name]
    t[:, 8:10] = plan["nb"]
    t[:, 10:22] = plan["mask"].reshape(64, 12)
    t[:, 22:24] = plan["mM"]
    ti = t.view(np.int32)
    ti[:, 24] = plan["jE"][:, 0] | (plan["jE"][:, 1] << 8) | (plan["jM"][:, 0] << 16) | (plan["jM"][:, 1] << 24)
    ti[:, 25] = plan["isE"][:, 0] | (plan["isE"][:, 1] << 1) | (plan["isM"][:, 0] << 2) | (plan["isM"][:, 1] << 3)
    valid = np.zeros((2, 129), dtype=np.int64)   # (half, interval): some lane writes the interval's sums
    for h in range(2):
        valid[h, plan["jE"][plan["isE"][:, h] != 0, h]] = 1
        valid[h, plan["jM"][plan["isM"][:, h] != 0, h]] = 1
    C = int(plan["C"])
    for r in range(2):
        c = np.arange(64) + 64 * r
        ok = c < C
        cc = np.minimum(c, 127)
        bits = valid[0, cc] | (valid[1, cc] << 1) | (valid[0, cc + 1] << 2) | (valid[1, cc + 1] << 3)
        ti[:, 30] |= np.where(ok, bits, 0).astype(np.int32) << (4 * r)
    if np.any(plan["h0"] != 0) or np.any(plan["h256"] != 0):
        ti[:, 30] |= 256   # bit 8 (every lane): bins 0 / 256 carry weight
    t[:, 26], t[:, 28] = plan["h0"][:64], plan["h0"][64:]
    t[:, 27], t[:, 29] = plan["h256"][:64], plan["h256"][64:]
    return t


def dct_matrix(dct_length: int, dct_type: int = 2) -> np.ndarray:
    """Orthonormal DCT-I..IV matrix W (L, L), y = x @ W (dct.py:99-133)."""
    L = dct_length
    n = np.arange(L, dtype=np.float64)
    k = np.arange(L, dtype=np.float64)
    if dct_type in (2, 4):
        n = n + 0.5
    if dct_type in (3, 4):
        k = k + 0.5
    n = n * (math.pi / ((L - 1) if dct_type == 1 else L))
    if dct_type == 1:
        z0 = np.full(L, 1.0)
        z0[0] = z0[-1] = math.sqrt(0.5)
        z1 = np.full(L, 2.0)
        z1[0] = z1[-1] = 1.0
        z = z0[None, :] * np.sqrt(z1 / (L - 1))[:, None]
    elif dct_type == 2:
        zz = np.full(L, 2.0)
        zz[0] = 1.0
        z = np.sqrt(zz / L)[None, :]
    elif dct_type == 3:
        zz = np.full(L, 2.0)
        zz[0] = 1.0
        z = np.sqrt(zz / L)[:, None]
    elif dct_type == 4:
        z = math.sqrt(2.0 / L)
    else:
        raise ValueError(f"dct_type {dct_type} is not supported.")
    return z * np.cos(k[None, :] * n[:, None])


def mfcc_lifter(mfcc_order: int, lifter: int) -> np.ndarray:
    """Liftering vector of mfcc.py:224-226 (entry 0 scales C0 by sqrt 2)."""
    r = np.arange(mfcc_order + 1, dtype=np.float64)
    v = 1.0 + (lifter / 2.0) * np.sin((math.pi / lifter) * r)   # lifter = 0: ZeroDivisionError, as in the reference
    v[0] = math.sqrt(2.0)
    return v


def even_cosine_matrix(fft_length: int) -> np.ndarray:
    """A (H, H), H = L/2 + 1: A[k][n] = c_k cos(2 pi k n / L), c = 1 at k = 0 and L/2, else 2.  For a real even
    sequence e, hfft(e)[:H] = e @ A and irfft(e)[:H] = ihfft(e).real = e @ A / L (fftcep.py:122-129)."""
    H = fft_length // 2 + 1
    k = np.arange(H, dtype=np.float64)
    c = np.where((k == 0) | (k == H - 1), 1.0, 2.0)
    idx = np.outer(np.arange(H), np.arange(H)) % fft_length          # exact argument reduction
    return c[:, None] * np.cos(2.0 * math.pi * idx / fft_length)


# ------------------------------------------------------------------ cepstrum conversions / mgcep
def mc2b_matrix(cep_order: int, alpha: float) -> np.ndarray:
    """mc2b.py:111-118: b = mc @ A, A[k][m] = (-alpha)^(k-m) for k >= m (b[m] = mc[m] - alpha b[m+1])."""
    n = cep_order + 1
    k = np.arange(n)
    d = k[:, None] - k[None, :]
    return np.where(d >= 0, (-alpha) ** np.maximum(d, 0).astype(np.float64), 0.0)


def b2mc_matrix(cep_order: int, alpha: float) -> np.ndarray:
    """b2mc.py: mc = b @ A, mc[m] = b[m] + alpha b[m+1]."""
    n = cep_order + 1
    A = np.eye(n)
    A[np.arange(1, n), np.arange(n - 1)] = alpha
    return A


def mgcep_freqt_matrix(in_order: int, out_order: int, alpha: float) -> np.ndarray:
    """CoefficientsFrequencyTransform of mgcep.py:255-282 (first row e_0, second row alpha^(j-1) (1 - alpha^2)),
    (in_order + 1, out_order + 1) -- NOT the matrix of the same name in mcep.py (coef_freqt_matrix above)."""
    L1, L2 = in_order + 1, out_order + 1
    A = np.zeros((L2, L1), dtype=np.float64)
    A[0, 0] = 1.0
    if L2 > 1 and L1 > 1:
        A[1, 1:] = alpha ** np.arange(L1 - 1, dtype=np.float64) * (1.0 - alpha * alpha)
    for i in range(2, L2):
        prev, cur = A[i - 1], A[i]
        acc = 0.0
        for j in range(1, L1):
            acc = prev[j - 1] + alpha * (acc - prev[j])
            cur[j] = acc
    return np.ascontiguousarray(A.T)


@functools.lru_cache(maxsize=8)
def mgcep_matrices(fft_length: int, cep_order: int, alpha: float):
    """The linear stages of MelGeneralizedCepstralAnalysis.forward (mgcep.py:181-249) composed in float64, so that a
    Newton step is a handful of row products around the pointwise spectrum arithmetic (H = L/2, K = H + 1 bins):
      Cr, Ci (M+1, K)   b -> cfreqt -> rfft(., L): real / imaginary part                       (:191-193)
      Pr     (K, 2M+1)  real half spectrum -> irfft -> pfreqt -> ptrans                         (:212,219)
      Qr, Qi (K, 2M+1)  complex half spectrum (re, im) -> irfft -> pfreqt -> qtrans             (:217,220)
      Rr, Ri (K, M+1)   complex half spectrum -> irfft -> rfreqt                                (:218)
      R1     (K, M+1)   gamma = -1: r = pfreqt(irfft(x))[:M+1]                                  (:213-215)
      Q1     (K, 2M+1)  gamma = -1: q = qtrans(pfreqt(irfft(x)))"""
    L, H, M = fft_length, fft_length // 2, cep_order
    K = H + 1
    n = np.arange(L, dtype=np.float64)
    k = np.arange(K, dtype=np.float64)
    ph = 2.0 * math.pi * np.outer(n, k) / L                      # (L, K)
    cf = mgcep_freqt_matrix(M, L - 1, -alpha)                    # (M+1, L)
    Cr, Ci = cf @ np.cos(ph), -(cf @ np.sin(ph))
    c = np.full((K, 1), 2.0)
    c[0] = c[-1] = 1.0
    IR, II = c * np.cos(ph.T) / L, -c * np.sin(ph.T) / L         # (K, L): irfft of (re, im)
    pf = mgcep_freqt_matrix(L - 1, 2 * M, alpha)                 # (L, 2M+1)
    rf = mgcep_freqt_matrix(L - 1, M, alpha)                     # (L, M+1)
    nn = 2 * M + 1
    P = np.eye(nn)
    P[np.arange(nn - 1), np.arange(1, nn)] = alpha               # A[:, 1:].fill_diagonal_(alpha), :299
    P[0, 0] -= alpha * alpha
    P[0, 1] += alpha
    P[-1, -1] += alpha
    Q = np.eye(nn)
    Q[np.arange(1, nn), np.arange(nn - 1)] = alpha               # A[1:].fill_diagonal_(alpha), :322
    Q[1, 0] = 0
    Q[1, 1] += alpha
    Pt, Qt = P.T, Q.T                                            # the modules right-multiply by A.T
    base_r, base_i = IR @ pf, II @ pf
    return {"Cr": Cr, "Ci": Ci, "Pr": base_r @ Pt, "Qr": base_r @ Qt, "Qi": base_i @ Qt, "Rr": IR @ rf, "Ri": II @ rf,
            "R1": base_r[:, : M + 1].copy(), "Q1": base_r @ Qt}


def mgcep_step_images(fft_length: int, cep_order: int, alpha: float) -> np.ndarray:
    """Operand images of dsa_mgcep_step (csrc/mgc.hip: layout in the kernel's header comment): per 16-bin tile the A operands of
    the first chain (Cr, Ci) and of the second chain (Pr[:, :M] | Qr[:, 2:] | Qi[:, 2:] | Rr | Ri in 16-column tiles) in
    v_mfma_f32_16x16x4_f32 lane order, float32.  fft_length 512, cep_order <= 24."""
    if fft_length != 512 or not 1 <= cep_order <= 24:
        raise ValueError("mgcep_step_images: fft_length 512 and cep_order <= 24 only")
    M, K = cep_order, fft_length // 2 + 1
    m = mgcep_matrices(fft_length, cep_order, float(alpha))
    Cr, Ci = m["Cr"], m["Ci"]                                    # (M + 1, K)
    mats = [(m["Pr"][:, :M], 2), (m["Qr"][:, 2:], 3), (m["Qi"][:, 2:], 3), (m["Rr"], 2), (m["Ri"], 2)]
    lanes = np.arange(64)
    li, lg = lanes & 15, lanes >> 4
    out = np.zeros((17, 768 + 3072), dtype=np.float64)
    for mt in range(17):
        a1 = np.zeros((2, 6, 64))
        for ci_, C in enumerate((Cr, Ci)):
            for ks in range(6):
                row = 1 + 4 * ks + lg
                col = 16 * mt + li
                ok = (row <= M) & (col < K)
                a1[ci_, ks, ok] = C[row[ok], col[ok]]
        a2 = np.zeros((12, 64, 4))
        c = 0
        for W, ntile in mats:
            for t in range(ntile):
                for r in range(4):
                    b = 16 * mt + 4 * lg + r
                    col = 16 * t + li
                    ok = (b < K) & (col < W.shape[1])
                    a2[c, ok, r] = W[b[ok], col[ok]]
                c += 1
        out[mt, :768] = a1.reshape(-1)
        out[mt, 768:] = a2.reshape(-1)
    return out.astype(np.float32)


MGCEP_STEP_H_LOG2_SC = 12   # scale of the (Cr, Ci) images of dsa_mgcep_step_solve (|C| <= 1.5)
MGCEP_STEP_H_LOG2_SW = 20   # scale of the (Pr, Qr, Qi, Rr, Ri) images (|W| <= 0.008)


def mgcep_step_h_images(fft_length: int, cep_order: int, alpha: float) -> np.ndarray:
    """Binary16 hi / lo operand images of dsa_mgcep_step_solve (csrc/mgcep_step_f16.h), float16, shape (9, 16384): per STAGE of 32
    bins (two 16-bin tiles t = 0, 1; bins 257 .. 287 of the ninth stage are zero rows)
      [2 t][2 (Cr, Ci)][2 (hi, lo)][64 lane][8 i]   first chain, A operand of v_mfma_f32_16x16x32_f16: row = bin 32 j + 16 t + (lane & 15),
                                                     k-slot (g = lane >> 4, i) <-> coefficient 1 + 8 g + i (zero past the order)
      [12 c][2 (hi, lo)][64 lane][8 i]              second chain: row = column 16 tile_c + (lane & 15) of the chain's matrix,
                                                     k-slot (g, i = 4 t + r) <-> bin 32 j + 16 t + 4 g + r
    chains c as in mgcep_step_images: 0-1 Pr[:, :M] | 2-4 Qr[:, 2:] | 5-7 Qi[:, 2:] | 8-9 Rr | 10-11 Ri.  Each float32 entry v is stored as
    hi = binary16(S v), lo = binary16(S v - hi) with S = 2^12 (C) / 2^20 (the others).  fft_length 512, cep_order 24."""
    if fft_length != 512 or cep_order != 24:
        raise ValueError("mgcep_step_h_images: fft_length 512 and cep_order 24 only")
    M, K = cep_order, fft_length // 2 + 1
    m = mgcep_matrices(fft_length, cep_order, float(alpha))
    Cr, Ci = m["Cr"], m["Ci"]                                    # (M + 1, K)
    mats = [(m["Pr"][:, :M], 2), (m["Qr"][:, 2:], 3), (m["Qi"][:, 2:], 3), (m["Rr"], 2), (m["Ri"], 2)]
    lanes = np.arange(64)
    li, lg = lanes & 15, lanes >> 4
    sc, sw = float(2 ** MGCEP_STEP_H_LOG2_SC), float(2 ** MGCEP_STEP_H_LOG2_SW)
    out = np.zeros((9, 16384), dtype=np.float16)

    def hilo(v):
        hi = v.astype(np.float16)
        lo = (v - hi.astype(np.float64)).astype(np.float16)
        return hi, lo

    for j in range(9):
        c1 = np.zeros((2, 2, 2, 64, 8), dtype=np.float16)
        for t in range(2):
            for ci_, C in enumerate((Cr, Ci)):
                v = np.zeros((64, 8))
                for i in range(8):
                    row = 1 + 8 * lg + i
                    col = 32 * j + 16 * t + li
                    ok = (row <= M) & (col < K)
                    v[ok, i] = sc * C[row[ok], col[ok]]
                c1[t, ci_, 0], c1[t, ci_, 1] = hilo(v)
        w2 = np.zeros((12, 2, 64, 8), dtype=np.float16)
        c = 0
        for W, ntile in mats:
            for tc in range(ntile):
                v = np.zeros((64, 8))
                for i in range(8):
                    b = 32 * j + 16 * (i >> 2) + 4 * lg + (i & 3)
                    col = 16 * tc + li
                    ok = (b < K) & (col < W.shape[1])
                    v[ok, i] = sw * W[b[ok], col[ok]]
                w2[c, 0], w2[c, 1] = hilo(v)
                c += 1
        out[j, :4096] = c1.reshape(-1)
        out[j, 4096:] = w2.reshape(-1)
    return out


def mgcep_step_h_buffer(fft_length: int, cep_order: int, alpha: float) -> np.ndarray:
    """What dsa_mgcep_step_solve takes as `images_h`, as bytes: mgcep_step_h_images (9 x 16384 binary16) followed by 240 float32 -- the
    matrices at the Nyquist bin, which the kernel applies with float32 multiply-adds instead of a ninth stage of 32 bins:
    Cr[1:25, 256] | Ci[1:25, 256] | rows 256 of Pr[:, :24] (32, zero-padded) | Qr[:, 2:] (48) | Qi[:, 2:] (48) | Rr (32) | Ri (32)."""
    img = mgcep_step_h_images(fft_length, cep_order, alpha)
    M, H = cep_order, fft_length // 2
    m = mgcep_matrices(fft_length, cep_order, float(alpha))
    t = np.zeros(240, dtype=np.float32)
    t[0:24] = m["Cr"][1:M + 1, H]
    t[24:48] = m["Ci"][1:M + 1, H]
    t[48:48 + M] = m["Pr"][H, :M]
    t[80:80 + 2 * M - 1] = m["Qr"][H, 2:]
    t[128:128 + 2 * M - 1] = m["Qi"][H, 2:]
    t[176:176 + M + 1] = m["Rr"][H, :]
    t[208:208 + M + 1] = m["Ri"][H, :]
    return np.concatenate([np.ascontiguousarray(img).view(np.uint8).reshape(-1), t.view(np.uint8)])


def mgcep_step_bwd_h_images(fft_length: int, cep_order: int, alpha: float) -> np.ndarray:
    """Binary16 hi / lo operand images of dsa_mgcep_step_bwd_h (csrc/mgcep_step_f16.h), float16, shape (9, 22528): per STAGE of 32 bins
      [2 t][2 (Cr, Ci)][2 (hi, lo)][64 lane][8 i]   the forward's first chain (re, im recomputed), as in mgcep_step_h_images
      [2 t][7 ks][2 (hi, lo)][64 lane][8 i]          the second-chain matrices with BINS as rows: row = bin 32 j + 16 t + (lane & 15),
                                                     k-slot (g, i) <-> column 8 g + i of the k-step's segment: ks 0 Pr[:, :24] | 1, 2 Qr[:, 2:]
                                                     columns 0 .. 31, 32 .. 63 | 3, 4 Qi[:, 2:] | 5 Rr | 6 Ri (zero past the matrix)
      [2 (Cr, Ci)][2 tc][2 (hi, lo)][64 lane][8 i]   (Cr, Ci) with COEFFICIENTS as rows: row = coefficient 1 + 16 tc + (lane & 15),
                                                     k-slot (g, i = 4 t + r) <-> bin 32 j + 16 t + 4 g + r
    Scales as in mgcep_step_h_images (2^12 for C, 2^20 for the others).  fft_length 512, cep_order 24."""
    if fft_length != 512 or cep_order != 24:
        raise ValueError("mgcep_step_bwd_h_images: fft_length 512 and cep_order 24 only")
    M, K = cep_order, fft_length // 2 + 1
    m = mgcep_matrices(fft_length, cep_order, float(alpha))
    Cr, Ci = m["Cr"], m["Ci"]
    segs = [(m["Pr"][:, :M], 0), (m["Qr"][:, 2:], 0), (m["Qr"][:, 2:], 32), (m["Qi"][:, 2:], 0), (m["Qi"][:, 2:], 32), (m["Rr"], 0), (m["Ri"], 0)]
    lanes = np.arange(64)
    li, lg = lanes & 15, lanes >> 4
    sc, sw = float(2 ** MGCEP_STEP_H_LOG2_SC), float(2 ** MGCEP_STEP_H_LOG2_SW)
    out = np.zeros((9, 22528), dtype=np.float16)

    def hilo(v):
        hi = v.astype(np.float16)
        lo = (v - hi.astype(np.float64)).astype(np.float16)
        return hi, lo

    for j in range(9):
        c1 = np.zeros((2, 2, 2, 64, 8), dtype=np.float16)
        wt = np.zeros((2, 7, 2, 64, 8), dtype=np.float16)
        ct = np.zeros((2, 2, 2, 64, 8), dtype=np.float16)
        for t in range(2):
            binrow = 32 * j + 16 * t + li
            for ci_, C in enumerate((Cr, Ci)):
                v = np.zeros((64, 8))
                for i in range(8):
                    row = 1 + 8 * lg + i
                    ok = (row <= M) & (binrow < K)
                    v[ok, i] = sc * C[row[ok], binrow[ok]]
                c1[t, ci_, 0], c1[t, ci_, 1] = hilo(v)
            for ks, (W, c0) in enumerate(segs):
                v = np.zeros((64, 8))
                for i in range(8):
                    col = c0 + 8 * lg + i
                    ok = (binrow < K) & (col < W.shape[1])
                    v[ok, i] = sw * W[binrow[ok], col[ok]]
                wt[t, ks, 0], wt[t, ks, 1] = hilo(v)
        for ci_, C in enumerate((Cr, Ci)):
            for tc in range(2):
                v = np.zeros((64, 8))
                for i in range(8):
                    b = 32 * j + 16 * (i >> 2) + 4 * lg + (i & 3)
                    row = 1 + 16 * tc + li
                    ok = (b < K) & (row <= M)
                    v[ok, i] = sc * C[row[ok], b[ok]]
                ct[ci_, tc, 0], ct[ci_, tc, 1] = hilo(v)
        out[j, :4096] = c1.reshape(-1)
        out[j, 4096:4096 + 14336] = wt.reshape(-1)
        out[j, 4096 + 14336:] = ct.reshape(-1)
    return out


def mgcep_step_bwd_images(fft_length: int, cep_order: int, alpha: float) -> np.ndarray:
    """Operand images of dsa_mgcep_step_bwd (csrc/mgc.hip:mgcep_step_bwd_kernel): per 16-bin tile the forward's first-chain
    operands, the second-chain matrices TRANSPOSED (bin rows x column k-steps: Pr[:, :M] 6 | Qr[:, 2:] 12 | Qi[:, 2:] 12 | Rr 7 |
    Ri 7 k-steps) and (Cr, Ci) with coefficient rows x bin k-steps, in v_mfma_f32_16x16x4_f32 lane order, float32."""
    if fft_length != 512 or not 1 <= cep_order <= 24:
        raise ValueError("mgcep_step_bwd_images: fft_length 512 and cep_order <= 24 only")
    M, K = cep_order, fft_length // 2 + 1
    m = mgcep_matrices(fft_length, cep_order, float(alpha))
    Cr, Ci = m["Cr"], m["Ci"]                                    # (M + 1, K)
    mats = [(m["Pr"][:, :M], 6), (m["Qr"][:, 2:], 12), (m["Qi"][:, 2:], 12), (m["Rr"], 7), (m["Ri"], 7)]
    fwd = mgcep_step_images(fft_length, cep_order, alpha).astype(np.float64)
    lanes = np.arange(64)
    li, lg = lanes & 15, lanes >> 4
    out = np.zeros((17, 768 + 44 * 64 + 16 * 64), dtype=np.float64)
    for mt in range(17):
        out[mt, :768] = fwd[mt, :768]
        a2 = np.zeros((44, 64))
        c = 0
        for W, nks in mats:
            for ks in range(nks):
                b = 16 * mt + li
                col = 4 * ks + lg
                ok = (b < K) & (col < W.shape[1])
                a2[c, ok] = W[b[ok], col[ok]]
                c += 1
        a3 = np.zeros((2, 2, 4, 64))
        for ci_, C in enumerate((Cr, Ci)):
            for t in range(2):
                for r in range(4):
                    row = 1 + 16 * t + li
                    b = 16 * mt + 4 * lg + r
                    ok = (row <= M) & (b < K)
                    a3[ci_, t, r, ok] = C[row[ok], b[ok]]
        out[mt, 768:768 + 44 * 64] = a2.reshape(-1)
        out[mt, 768 + 44 * 64:] = a3.reshape(-1)
    return out.astype(np.float32)


def fbank_bins_table(H: np.ndarray):
    """Per-bin table of the fused filter bank's backward (dsa_fbank_bins_bwd; the C twin is dsa_fbank_bins_plan): row k =
    (bits of c_k as float32, w0, w1, 0) with H[k, c_k] = w0 and H[k, c_k + 1] = w1 the only non-zero entries of row k.
    None when a bin feeds more than two adjacent channels."""
    H = np.asarray(H, dtype=np.float64)
    if H.ndim != 2 or not np.all(np.isfinite(H)):
        return None
    K, C = H.shape
    idx = np.zeros(K, dtype=np.int32)
    out = np.zeros((K, 4), dtype=np.float32)
    for k in range(K):
        nz = np.flatnonzero(H[k])
        if len(nz) > 2 or (len(nz) == 2 and nz[1] != nz[0] + 1):
            return None
        if len(nz):
            idx[k] = nz[0]
            out[k, 1] = H[k, nz[0]]
            if len(nz) == 2:
                out[k, 2] = H[k, nz[1]]
    out[:, 0] = idx.view(np.float32)
    return out
