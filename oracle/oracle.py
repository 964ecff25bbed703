"""ORACLE -- TEST INFRASTRUCTURE ONLY.

numpy/ctypes front-end of the C restatement in ``sptk_oracle.c`` (the CPU restatement of
the sp-nitech/diffsptk STFT -> mel-cepstrum / LPC path).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product package ``diffsptk_amd`` never does.

Pinned against golden vectors generated from the reference itself
(``tests/golden/make_golden.py``), see ``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsptk_oracle.so")
_lib = None

WINDOWS = {
    "blackman": 0, "hamming": 1, "hanning": 2, "bartlett": 3, "trapezoidal": 4,
    "rectangular": 5, "nuttall": 6, "povey": 7, "sine": 8, "vorbis": 9, "kbd": 10,
}
NORMS = {"none": 0, "power": 1, "magnitude": 2}
PAD_MODES = {"constant": 0, "reflect": 1, "replicate": 2, "circular": 3}
SPEC_FORMATS = {"db": 0, "log-magnitude": 1, "magnitude": 2, "power": 3, "complex": 4}
FFTR_FORMATS = {"complex": 0, "real": 1, "imaginary": 2, "amplitude": 3, "power": 4}
ACORR_FORMATS = {"naive": 0, "normalized": 1, "biased": 2, "unbiased": 3}


def build(force: bool = False) -> str:
    """Compile the C oracle (gcc) if needed and return the library path."""
    src = [os.path.join(_HERE, f) for f in ("sptk_oracle.c", "sptk_oracle_impl.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsptk_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _sfx(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32"
    if dtype == np.float64:
        return "_f64"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _real(dtype):
    return C.c_float if np.dtype(dtype) == np.float32 else C.c_double


def _code(table, key):
    return key if isinstance(key, int) else table[key]


def _as2d(x):
    x = np.ascontiguousarray(x)
    return x.reshape(-1, x.shape[-1]), x.shape[:-1]


def num_frames(T: int, P: int) -> int:
    return 0 if T <= 0 else (T - 1) // P + 1


def window_table(L, window="blackman", norm="power", symmetric=True, dtype=np.float64):
    w = np.empty(L, dtype=np.float64)
    rc = lib().oracle_window_table(_code(WINDOWS, window), int(L), _code(NORMS, norm),
                                   int(bool(symmetric)), _ptr(w))
    if rc != 0:
        raise ValueError("unsupported window configuration")
    return w.astype(dtype)


def frame(x, L, P, center=True, zmean=False, mode="constant"):
    x2, lead = _as2d(x)
    B, T = x2.shape
    N = num_frames(T, P)
    y = np.empty((B, N, L), dtype=x2.dtype)
    fn = getattr(lib(), "oracle_frame" + _sfx(x2.dtype))
    fn(_ptr(x2), C.c_long(B), C.c_long(T), int(L), int(P), int(center), int(zmean),
       _code(PAD_MODES, mode), _ptr(y))
    return y.reshape(*lead, N, L)


def window(x, w, out_length=None):
    x2, lead = _as2d(x)
    F, L = x2.shape
    L2 = L if out_length is None else out_length
    w = np.ascontiguousarray(w, dtype=x2.dtype)
    y = np.empty((F, L2), dtype=x2.dtype)
    getattr(lib(), "oracle_window" + _sfx(x2.dtype))(_ptr(x2), C.c_long(F), L, _ptr(w), L2, _ptr(y))
    return y.reshape(*lead, L2)


def fftr(x, fft_length=None, out_format="complex"):
    x2, lead = _as2d(x)
    F, Lin = x2.shape
    n = Lin if fft_length is None else fft_length
    fmt = _code(FFTR_FORMATS, out_format)
    K = n // 2 + 1
    y = np.empty((F, K, 2) if fmt == 0 else (F, K), dtype=x2.dtype)
    getattr(lib(), "oracle_fftr" + _sfx(x2.dtype))(_ptr(x2), C.c_long(F), min(Lin, n), n, fmt, _ptr(y))
    if fmt == 0:
        return (y[..., 0] + 1j * y[..., 1]).reshape(*lead, K)
    return y.reshape(*lead, K)


def spec(b=None, a=None, fft_length=512, eps=0.0, relative_floor=None, out_format="power"):
    ref = b if b is not None else a
    b2 = _as2d(np.asarray(b))[0] if b is not None else None
    a2 = _as2d(np.asarray(a))[0] if a is not None else None
    dt = np.asarray(ref).dtype
    lead = np.asarray(ref).shape[:-1]
    F = (b2 if b2 is not None else a2).shape[0]
    K = fft_length // 2 + 1
    y = np.empty((F, K), dtype=dt)
    R = _real(dt)
    getattr(lib(), "oracle_spec" + _sfx(dt))(
        _ptr(b2), 0 if b2 is None else b2.shape[1], _ptr(a2), 0 if a2 is None else a2.shape[1],
        C.c_long(F), int(fft_length), R(eps), int(relative_floor is not None),
        R(0.0 if relative_floor is None else relative_floor), _code(SPEC_FORMATS, out_format), _ptr(y))
    return y.reshape(*lead, K)


def stft(x, frame_length, frame_period, fft_length, *, center=True, zmean=False,
         mode="constant", window="blackman", norm="power", symmetric=True, eps=1e-9,
         relative_floor=None, out_format="power"):
    x2, lead = _as2d(x)
    B, T = x2.shape
    dt = x2.dtype
    N = num_frames(T, frame_period)
    K = fft_length // 2 + 1
    fmt = _code(SPEC_FORMATS, out_format)
    w = window_table(frame_length, window, norm, symmetric, dtype=dt)
    y = np.empty((B, N, K, 2) if fmt == 4 else (B, N, K), dtype=dt)
    R = _real(dt)
    getattr(lib(), "oracle_stft" + _sfx(dt))(
        _ptr(x2), C.c_long(B), C.c_long(T), int(frame_length), int(frame_period),
        int(fft_length), _ptr(w), int(center), int(zmean), _code(PAD_MODES, mode), R(eps),
        int(relative_floor is not None), R(0.0 if relative_floor is None else relative_floor),
        fmt, _ptr(y))
    if fmt == 4:
        return (y[..., 0] + 1j * y[..., 1]).reshape(*lead, N, K)
    return y.reshape(*lead, N, K)


def freqt_matrix(in_order, out_order, alpha, dtype=np.float64):
    At = np.empty((in_order + 1, out_order + 1), dtype=dtype)
    getattr(lib(), "oracle_freqt_matrix" + _sfx(dtype))(int(in_order), int(out_order),
                                                        C.c_double(alpha), _ptr(At))
    return At


def rfreqt_matrix(in_order, out_order, alpha, dtype=np.float64):
    At = np.empty((in_order + 1, out_order + 1), dtype=dtype)
    getattr(lib(), "oracle_rfreqt_matrix" + _sfx(dtype))(int(in_order), int(out_order),
                                                         C.c_double(alpha), _ptr(At))
    return At


def freqt(c, out_order, alpha=0.0):
    c2, lead = _as2d(c)
    F, L1 = c2.shape
    At = freqt_matrix(L1 - 1, out_order, alpha, dtype=c2.dtype)
    out = np.empty((F, out_order + 1), dtype=c2.dtype)
    getattr(lib(), "oracle_matmul" + _sfx(c2.dtype))(_ptr(c2), C.c_long(F), L1, _ptr(At),
                                                     out_order + 1, _ptr(out))
    return out.reshape(*lead, out_order + 1)


def mcep(X, cep_order, alpha=0.0, n_iter=0, return_trace=False):
    X2, lead = _as2d(X)
    F, K = X2.shape
    n = 2 * (K - 1)
    dt = X2.dtype
    mc = np.empty((F, cep_order + 1), dtype=dt)
    trace = np.empty((n_iter + 1, F, cep_order + 1), dtype=dt) if return_trace else None
    rc = getattr(lib(), "oracle_mcep" + _sfx(dt))(_ptr(X2), C.c_long(F), n, int(cep_order),
                                                  C.c_double(alpha), int(n_iter), _ptr(mc), _ptr(trace))
    if rc != 0:
        raise np.linalg.LinAlgError("singular Toeplitz-plus-Hankel system")
    mc = mc.reshape(*lead, cep_order + 1)
    if return_trace:
        return mc, trace.reshape(n_iter + 1, *lead, cep_order + 1)
    return mc


def acorr(x, acr_order, out_format="naive"):
    x2, lead = _as2d(x)
    F, L = x2.shape
    r = np.empty((F, acr_order + 1), dtype=x2.dtype)
    getattr(lib(), "oracle_acorr" + _sfx(x2.dtype))(_ptr(x2), C.c_long(F), L, int(acr_order),
                                                    _code(ACORR_FORMATS, out_format), _ptr(r))
    return r.reshape(*lead, acr_order + 1)


def _default_eps(dtype, eps):
    if eps is None:  # levdur.py:108-109
        return 1e-5 if np.dtype(dtype) == np.float32 else 0.0
    return eps


def levdur(r, eps=None):
    r2, lead = _as2d(r)
    F, M1 = r2.shape
    out = np.empty((F, M1), dtype=r2.dtype)
    R = _real(r2.dtype)
    rc = getattr(lib(), "oracle_levdur" + _sfx(r2.dtype))(_ptr(r2), C.c_long(F), M1 - 1,
                                                          R(_default_eps(r2.dtype, eps)), _ptr(out))
    if rc != 0:
        raise np.linalg.LinAlgError("singular Yule-Walker system")
    return out.reshape(*lead, M1)


def lpc(x, lpc_order, eps=None):
    x2, lead = _as2d(x)
    F, L = x2.shape
    out = np.empty((F, lpc_order + 1), dtype=x2.dtype)
    R = _real(x2.dtype)
    rc = getattr(lib(), "oracle_lpc" + _sfx(x2.dtype))(_ptr(x2), C.c_long(F), L, int(lpc_order),
                                                       R(_default_eps(x2.dtype, eps)), _ptr(out))
    if rc != 0:
        raise np.linalg.LinAlgError("singular Yule-Walker system")
    return out.reshape(*lead, lpc_order + 1)


def stft_mcep(x, *, frame_length=400, frame_period=80, fft_length=512, cep_order=24,
              alpha=0.42, n_iter=10, **stft_kw):
    """The headline path: STFT power -> mel-cepstrum."""
    return mcep(stft(x, frame_length, frame_period, fft_length, **stft_kw), cep_order, alpha, n_iter)


def frame_window_lpc(x, *, frame_length=400, frame_period=80, lpc_order=24, eps=1e-5,
                     window="blackman", norm="power"):
    """The LPC branch: Frame -> Window -> LPC (README.md:198-201 of the reference)."""
    fr = frame(x, frame_length, frame_period)
    w = window_table(frame_length, window, norm, True, dtype=fr.dtype)
    return lpc(window_apply(fr, w), lpc_order, eps)


def window_apply(x, w):
    return window(x, w, None)


# ------------------------------------------------------------------ mel filter bank / MFCC
def _to_aud(f, scale):
    f = np.asarray(f, dtype=np.float64)
    return {"htk": lambda: 1127 * np.log1p(f / 700), "mel": lambda: 2595 * np.log10(1 + f / 700),
            "inverted-mel": lambda: 2195.286 - 2595 * np.log10(1 + (4031.25 - f) / 700),
            "bark": lambda: 26.81 * f / (1960 + f) - 0.53, "linear": lambda: f}[scale]()


def fbank_matrix(fft_length, n_channel, sample_rate, f_min=0.0, f_max=None, scale="htk"):
    """Triangular weights (K, C) of fbank.py:242-266 (erb_factor None), vectorised over the bins."""
    f_max = sample_rate / 2 if f_max is None else f_max
    K = fft_length // 2 + 1
    lo, hi = _to_aud(f_min, scale), _to_aud(f_max, scale)
    k0 = max(1, int(f_min / sample_rate * fft_length + 1.5))
    k1 = min(fft_length // 2, int(f_max / sample_rate * fft_length + 0.5))
    cen = lo + (hi - lo) / (n_channel + 1) * np.arange(0, n_channel + 2)   # cen[0] = lo .. cen[C+1] = hi
    H = np.zeros((K, n_channel))
    ks = np.arange(k0, k1)
    z = _to_aud(sample_rate * ks / fft_length, scale)
    up = np.searchsorted(cen[1:], z, side="left") + 1                        # first centre index >= z (1-based)
    w_lo = (cen[up] - z) / (cen[up] - cen[up - 1])
    for k, m, w in zip(ks, up, w_lo):
        if m >= 2:
            H[k, m - 2] = w
        if m <= n_channel:
            H[k, m - 1] = 1 - w
    return H


def fbank(X, H, floor=1e-5, gamma=0.0, use_power=False):
    """y:(..., C), E:(..., 1) through the C restatement (oracle_fbank)."""
    X2, lead = _as2d(X)
    Hc = np.ascontiguousarray(H, dtype=X2.dtype)
    F, K = X2.shape
    Cn = Hc.shape[1]
    y = np.empty((F, Cn), dtype=X2.dtype)
    E = np.empty((F,), dtype=X2.dtype)
    getattr(lib(), "oracle_fbank" + _sfx(X2.dtype))(_ptr(X2), C.c_long(F), K, _ptr(Hc), Cn, C.c_double(floor),
                                                    C.c_double(gamma), int(bool(use_power)), _ptr(y), _ptr(E))
    return y.reshape(*lead, Cn), E.reshape(*lead, 1)


def dct2_matrix(L):
    n = (np.arange(L) + 0.5) * (np.pi / L)
    z = np.sqrt(np.where(np.arange(L) == 0, 1.0, 2.0) / L)
    return z[None, :] * np.cos(np.outer(n, np.arange(L)))


def mfcc(X, H, mfcc_order, lifter=1, floor=1e-5, gamma=0.0):
    """(y:(..., M), c:(..., 1), E:(..., 1)) of mfcc.py:244-256."""
    fb, E = fbank(X, H, floor, gamma, False)
    cy = fb @ dct2_matrix(H.shape[1]).astype(fb.dtype)[:, : mfcc_order + 1]
    lift = 1 + (lifter / 2) * np.sin(np.pi / lifter * np.arange(mfcc_order + 1))
    lift[0] = np.sqrt(2.0)
    cy = cy * lift.astype(fb.dtype)
    return cy[..., 1:], cy[..., :1], E


# ------------------------------------------------------------------ inverse path
def ifftr(Y, out_length=None):
    """irfft(Y)[..., :out_length] by the definition (ifftr.py:138): own O(N^2) sum, float64."""
    Y = np.asarray(Y)
    K = Y.shape[-1]
    n = 2 * (K - 1)
    L = n if out_length is None else out_length
    k = np.arange(K)
    c = np.where((k == 0) | (k == K - 1), 1.0, 2.0) / n
    ang = 2 * np.pi * np.outer(k, np.arange(L)) / n
    return (Y.real * c) @ np.cos(ang) - (Y.imag * c) @ np.sin(ang)


def unframe(y, frame_period, center=True, w=None, out_length=None):
    """fold(y w) / (fold(w w) + 1e-16) with the centre trim (unframe.py:176-207): plain loops."""
    y = np.asarray(y, dtype=np.float64)
    N, L = y.shape[-2:]
    w = np.ones(L) if w is None else np.asarray(w, dtype=np.float64)
    lead = y.shape[:-2]
    y2 = y.reshape(-1, N, L)
    full = (N - 1) * frame_period + L
    num = np.zeros((y2.shape[0], full))
    den = np.zeros(full)
    for f in range(N):
        num[:, f * frame_period: f * frame_period + L] += y2[:, f] * w
        den[f * frame_period: f * frame_period + L] += w * w
    x = num / (den + 1e-16)
    s = L // 2 if center else 0
    if out_length is None and center:
        out_length = N * frame_period
    e = None if out_length is None else s + out_length
    return x[:, s:e].reshape(*lead, -1)


def istft(Y, frame_length, frame_period, center=True, w=None, out_length=None):
    """unframe(ifftr(Y)[..., :L]) (istft.py:186-193)."""
    return unframe(ifftr(Y, frame_length), frame_period, center, w, out_length)


def griffin(y, frame_length, frame_period, fft_length, *, out_length=None, center=True, window="blackman",
            norm="power", symmetric=True, n_iter=100, alpha=0.99, beta=0.99, gamma=1.1, phase=None):
    """GriffinLim._forward (griffin.py:263-292) with the initial phase given (None: zeros): float64 numpy over
    this oracle's own stft / istft."""
    y = np.asarray(y, dtype=np.float64)
    N = y.shape[-2]
    eps = 1e-16
    s = np.sqrt(y + eps)
    angle = np.exp(1j * (np.zeros_like(s) if phase is None else np.asarray(phase, dtype=np.float64)))
    w = window_table(frame_length, window, norm, symmetric)
    kw = dict(center=center, window=window, norm=norm, symmetric=symmetric, eps=0.0, out_format="complex")
    t_prev = d_prev = 0
    for n in range(n_iter):
        x = istft(s * angle, frame_length, frame_period, center, w, out_length)
        t = stft(x, frame_length, frame_period, fft_length, **kw)[..., :N, :]
        if n == 0:
            c = d = t
        else:
            t = (1 - gamma) * d_prev + gamma * t
            diff = t - t_prev
            c = t + alpha * diff
            d = t + beta * diff
        angle = c / (np.abs(c) + eps)
        t_prev, d_prev = t, d
    return istft(s * angle, frame_length, frame_period, center, w, out_length)


def fftcep(X, cep_order, accel=0.0, n_iter=0):
    """CepstralAnalysis._forward (fftcep.py:116-136) with numpy's FFTs on float64."""
    X = np.asarray(X, dtype=np.float64)
    N = cep_order + 1
    H = X.shape[-1]
    e = np.fft.irfft(np.log(X))                                   # fftcep.py:122
    v = e[..., :N].copy()
    pad = [(0, 0)] * (X.ndim - 1)
    e = np.pad(e[..., N:H], pad + [(N, 0)])
    for _ in range(n_iter):
        e = np.fft.hfft(e)                                        # fftcep.py:127
        e[e < 0] = 0
        e = np.fft.ihfft(e).real
        t = e[..., :N] * (1 + accel)
        v = v + t
        e = e - np.pad(t, pad + [(0, H - N)])
    idx = [0, N - 1] if H == N else [0]                           # fftcep.py:134-135
    v[..., idx] *= 0.5
    return v


# ----------------------------------------------------------------------------- SURVEY 8(f) rows 3-4
# Cepstrum conversions and mel-generalized cepstral analysis, float64 numpy (test infrastructure, pinned to
# tests/golden/synth.npz in tests/test_oracle_golden.py).
def mc2b(mc, alpha):
    """MelCepstrumToMLSADigitalFilterCoefficients (mc2b.py:95-99): b[M] = mc[M], b[m] = mc[m] - alpha b[m+1]."""
    mc = np.asarray(mc, dtype=np.float64)
    b = mc.copy()
    for m in range(mc.shape[-1] - 2, -1, -1):
        b[..., m] = mc[..., m] - alpha * b[..., m + 1]
    return b


def b2mc(b, alpha):
    """MLSADigitalFilterCoefficientsToMelCepstrum (b2mc.py): mc[m] = b[m] + alpha b[m+1]."""
    b = np.asarray(b, dtype=np.float64)
    mc = b.copy()
    mc[..., :-1] += alpha * b[..., 1:]
    return mc


def gnorm(x, gamma):
    """GeneralizedCepstrumGainNormalization (gnorm.py:102-112)."""
    x = np.asarray(x, dtype=np.float64)
    x0, x1 = x[..., :1], x[..., 1:]
    if gamma == 0:
        return np.concatenate((np.exp(x0), x1), -1)
    z = 1 + gamma * x0
    return np.concatenate((z ** (1 / gamma), x1 / z), -1)


def ignorm(y, gamma):
    """GeneralizedCepstrumInverseGainNormalization (ignorm.py:99-109)."""
    y = np.asarray(y, dtype=np.float64)
    K, y1 = y[..., :1], y[..., 1:]
    if gamma == 0:
        return np.concatenate((np.log(K), y1), -1)
    z = K ** gamma
    return np.concatenate(((z - 1) / gamma, y1 * z), -1)


def gc2gc(c1, out_order, in_gamma, out_gamma, n_fft=512):
    """GeneralizedCepstrumToGeneralizedCepstrum._forward (mgc2mgc.py:333-361), the FFT formulation."""
    c1 = np.asarray(c1, dtype=np.float64)
    c01 = c1.copy()
    c01[..., 0] = 0
    C1 = np.fft.fft(c01, n=n_fft)
    if in_gamma == 0:
        sC1 = np.exp(C1)
    else:
        z = 1 + in_gamma * C1
        sC1 = np.abs(z) ** (1 / in_gamma) * np.exp(1j * np.angle(z) / in_gamma)
    if out_gamma == 0:
        C2 = np.log(np.abs(sC1))                                 # clog of utils/private.py:318-319: the real log-magnitude
    else:
        C2 = (np.abs(sC1) ** out_gamma * np.cos(np.angle(sC1) * out_gamma) - 1) / out_gamma
    c02 = np.fft.ifft(C2).real[..., : out_order + 1]
    return np.concatenate((c1[..., :1], 2 * c02[..., 1:]), -1)


def _gamma_scale(c, s):
    out = np.asarray(c, dtype=np.float64).copy()
    out[..., 1:] *= s
    return out


def mgc2mgc(mc, out_order, in_alpha=0.0, out_alpha=0.0, in_gamma=0.0, out_gamma=0.0, in_norm=False, out_norm=False,
            in_mul=False, out_mul=False, n_fft=512):
    """MelGeneralizedCepstrumToMelGeneralizedCepstrum (mgc2mgc.py:176-300): the same sequence of elementary steps."""
    c = np.asarray(mc, dtype=np.float64)
    in_order = c.shape[-1] - 1
    if not in_norm and in_mul:                                   # ZerothGammaDivision
        c = c.copy()
        c[..., 0] = (c[..., 0] - 1) / in_gamma
    alpha = (out_alpha - in_alpha) / (1 - in_alpha * out_alpha)
    if alpha == 0:
        if in_order == out_order and in_gamma == out_gamma:
            if not in_mul and out_mul:
                c = _gamma_scale(c, in_gamma)
            if not in_norm and out_norm:
                c = gnorm(c, in_gamma)
            if in_norm and not out_norm:
                c = ignorm(c, out_gamma)
            if in_mul and not out_mul:
                c = _gamma_scale(c, 1 / out_gamma)
        else:
            if in_mul:
                c = _gamma_scale(c, 1 / in_gamma)
            if not in_norm:
                c = gnorm(c, in_gamma)
            c = gc2gc(c, out_order, in_gamma, out_gamma, n_fft)
            if not out_norm:
                c = ignorm(c, out_gamma)
            if out_mul:
                c = _gamma_scale(c, out_gamma)
    else:
        if in_mul:
            c = _gamma_scale(c, 1 / in_gamma)
        if in_norm:
            c = ignorm(c, in_gamma)
        c = c @ freqt_matrix(in_order, out_order, alpha)
        if out_norm or in_gamma != out_gamma:
            c = gnorm(c, in_gamma)
        if in_gamma != out_gamma:
            c = gc2gc(c, out_order, in_gamma, out_gamma, n_fft)
        if not out_norm and in_gamma != out_gamma:
            c = ignorm(c, out_gamma)
        if out_mul:
            c = _gamma_scale(c, out_gamma)
    if not out_norm and out_mul:                                 # ZerothGammaMultiplication
        c = c.copy()
        c[..., 0] = c[..., 0] * out_gamma + 1
    return c


def mgc2sp(mc, fft_length, alpha=0.0, gamma=0.0, norm=False, mul=False, n_fft=512, out_format="power"):
    """MelGeneralizedCepstrumToSpectrum (mgc2sp.py:150-202): cepstrum of order fft_length/2, rfft, formatter."""
    c = mgc2mgc(mc, fft_length // 2, in_alpha=alpha, out_alpha=0.0, in_gamma=gamma, out_gamma=0.0, in_norm=norm,
                out_norm=False, in_mul=mul, out_mul=False, n_fft=n_fft)
    sp = np.fft.rfft(c, n=fft_length)
    if out_format in (0, "db"):
        return sp.real * (20 / np.log(10))
    if out_format in (1, "log-magnitude"):
        return sp.real
    if out_format in (2, "magnitude"):
        return np.exp(sp.real)
    if out_format in (3, "power"):
        return np.exp(2 * sp.real)
    if out_format in (4, "cycle"):
        return sp.imag / np.pi
    if out_format in (5, "radian"):
        return sp.imag
    if out_format in (6, "degree"):
        return sp.imag * (180 / np.pi)
    if out_format == "complex":
        return np.exp(sp.real) * np.exp(1j * sp.imag)
    raise ValueError(f"out_format {out_format} is not supported.")


def coef_freqt_matrix(in_order, out_order, alpha):
    """CoefficientsFrequencyTransform of mgcep.py:255-282 (first row e_0, second row alpha^(j-1) (1 - alpha^2)):
    returns (in_order + 1, out_order + 1)."""
    L1, L2 = in_order + 1, out_order + 1
    A = np.zeros((L2, L1))
    A[0, 0] = 1
    if L2 > 1 and L1 > 1:
        A[1, 1:] = alpha ** np.arange(L1 - 1) * (1 - alpha * alpha)
    for i in range(2, L2):
        for j in range(1, L1):
            A[i, j] = A[i - 1, j - 1] + alpha * (A[i, j - 1] - A[i - 1, j])
    return A.T.copy()


def pq_matrices(order, alpha):
    """PTransform / QTransform of mgcep.py:285-331, as (order + 1, order + 1) right-multiplication matrices."""
    n = order + 1
    P = np.eye(n)
    P[np.arange(n - 1), np.arange(1, n)] = alpha
    P[0, 0] -= alpha * alpha
    P[0, 1] += alpha
    P[-1, -1] += alpha
    Q = np.eye(n)
    Q[np.arange(1, n), np.arange(n - 1)] = alpha
    Q[1, 0] = 0
    Q[1, 1] += alpha
    return P.T.copy(), Q.T.copy()


def mgcep(X, cep_order, alpha=0.0, gamma=0.0, n_iter=0):
    """MelGeneralizedCepstralAnalysis.forward (mgcep.py:181-249), gamma in [-1, 0)."""
    X = np.asarray(X, dtype=np.float64)
    if gamma == 0:
        return mcep(X, cep_order, alpha, n_iter)
    M = cep_order
    H = X.shape[-1] - 1
    L = 2 * H
    cfreqt = coef_freqt_matrix(M, L - 1, -alpha)
    pfreqt = coef_freqt_matrix(L - 1, 2 * M, alpha)
    rfreqt = coef_freqt_matrix(L - 1, M, alpha)
    P, Q = pq_matrices(2 * M, alpha)
    ii = np.arange(M)
    toep = np.abs(ii[:, None] - ii[None, :])
    hank = ii[:, None] + ii[None, :]

    def newton(gam, b1):
        b = np.concatenate((np.zeros_like(b1[..., :1]), b1), -1)
        C = np.fft.rfft(b @ cfreqt, n=L)
        if gam == -1:
            p = np.fft.irfft(X) @ pfreqt
            q, r = p, p[..., : M + 1]
        else:
            Xr, Y = 1 + gam * C.real, gam * C.imag
            D = Xr * Xr + Y * Y
            pp = X * D ** (-1 / gam) / D
            qq = pp / D
            p = np.fft.irfft(pp) @ pfreqt
            q = np.fft.irfft(qq * (Xr * Xr - Y * Y) + 1j * qq * (2 * Xr * Y)) @ pfreqt
            r = np.fft.irfft(pp * Xr + 1j * pp * Y) @ rfreqt
        p, q = p @ P, q @ Q
        if gam != -1:
            eps = r[..., 0] + gam * (r[..., 1:] * b1).sum(-1)
        pt, qt, rt = p[..., :M], q[..., 2:] * (1 + gam), r[..., 1:]
        A = pt[..., toep] + qt[..., hank]
        b1 = b1 + np.linalg.solve(A, rt[..., None])[..., 0]
        if gam == -1:
            eps = r[..., 0] + gam * (r[..., 1:] * b1).sum(-1)
        return np.sqrt(eps)[..., None], b1

    b1 = np.zeros(X.shape[:-1] + (M,))
    b0, b1 = newton(-1, b1)
    if gamma != -1:
        b = np.concatenate((b0, b1), -1)
        b = gnorm(mc2b(mgc2mgc(b2mc(ignorm(b, -1), alpha), M, in_gamma=-1, out_gamma=gamma), alpha), gamma)   # b2b
        b1 = b[..., 1:]
        for _ in range(n_iter):
            b0, b1 = newton(gamma, b1)
    return b2mc(ignorm(np.concatenate((b0, b1), -1), gamma), alpha)


# ----------------------------------------------------------------------------- MLSA synthesis filter
def linear_intpl(x, P):
    """LinearInterpolation._forward (linear_intpl.py:85-117): (..., N, D) -> (..., N P, D), frame n -> n + 1 over P
    samples, the last frame held."""
    x = np.asarray(x, dtype=np.float64)
    nxt = np.concatenate((x[..., 1:, :], x[..., -1:, :]), axis=-2)
    w = (np.arange(P) / P)[:, None]
    y = x[..., :, None, :] + w * (nxt - x)[..., :, None, :]
    return y.reshape(*x.shape[:-2], x.shape[-2] * P, x.shape[-1])


def zerodf(x, b, P, ignore_gain=False, zeroth_index=0):
    """AllZeroDigitalFilter (zerodf.py:184-243): y[t] = sum_k h_t[k] x[t - k + z0] with linearly interpolated taps."""
    x, b = np.asarray(x, dtype=np.float64), np.asarray(b, dtype=np.float64)
    M = b.shape[-1] - 1
    h = linear_intpl(b, P)                                       # (..., T, M+1)
    T = x.shape[-1]
    xp = np.concatenate((np.zeros(x.shape[:-1] + (M - zeroth_index,)), x, np.zeros(x.shape[:-1] + (zeroth_index,))), -1)
    y = np.zeros_like(x)
    for k in range(M + 1):                                       # x[t - k + z0] = xp[t + M - k]
        y += h[..., k] * xp[..., M - k: M - k + T]
    if ignore_gain:
        y = y / (h[..., 0] if zeroth_index != M or M == 0 else h[..., M])
    return y


def _mirror(x, half=False):
    x1 = x[..., 1:] * (0.5 if half else 1.0)
    return np.concatenate((x1[..., ::-1], x[..., :1], x1), -1)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def mlsa_mixed(x, mc, P, orders, alpha=0.0, gamma=0.0, ignore_gain=False, mode="multi-stage", **kw):
    """PseudoMGLSADigitalFilter(phase="mixed") (mglsadf.py:144-147, 240-246; multi-stage :356-365, single-stage :507-521,
    freq-domain :636-637).  mc:(.., N + M + 1) = c_{-N} .. c_{-1}, c_0 .. c_M; orders = (N, M) as the reference's
    filter_order; cep_order / ir_length pairs are (maximum-phase, minimum-phase) too."""
    x, mc = np.asarray(x, dtype=np.float64), np.asarray(mc, dtype=np.float64)
    N, M = _pair(orders)
    mc_min = mc[..., N:]
    mc_max = np.concatenate((np.zeros_like(mc[..., :1]), mc[..., :N][..., ::-1]), -1)   # (0, c_{-1}, .., c_{-N})
    if mode == "multi-stage":
        taylor_order, n_fft = kw.get("taylor_order", 20), kw.get("n_fft", 512)
        co_max, co_min = _pair(kw.get("cep_order", 199))
        c_min = mgc2mgc(mc_min, co_min, in_alpha=alpha, in_gamma=gamma, n_fft=n_fft)
        c_max = mgc2mgc(mc_max, co_max, in_alpha=alpha, in_gamma=gamma, n_fft=n_fft)
        c0 = c_min[..., :1] + c_max[..., :1]
        c = np.concatenate((c_max[..., 1:][..., ::-1], np.zeros_like(c0), c_min[..., 1:]), -1)
        y, cur = x.copy(), x
        for i in range(1, taylor_order + 1):
            cur = zerodf(cur, c, P, zeroth_index=co_max) * (1.0 / i)
            y = y + cur
        if not ignore_gain:
            y = y * np.exp(linear_intpl(c0, P))[..., 0]
        return y
    if mode == "single-stage":
        n_fft = kw.get("n_fft", 4096)
        il_max, il_min = _pair(kw.get("ir_length", 2000))
        c_min = mgc2mgc(mc_min, il_min - 1, in_alpha=alpha, in_gamma=gamma, n_fft=n_fft)
        c_max = mgc2mgc(mc_max, il_max - 1, in_alpha=alpha, in_gamma=gamma, n_fft=n_fft)
        c0 = np.zeros_like(c_min[..., :1]) if ignore_gain else c_min[..., :1] + c_max[..., :1]
        c = np.concatenate((c_max[..., 1:][..., ::-1], c0, c_min[..., 1:]), -1)
        c = np.concatenate((c, np.zeros(c.shape[:-1] + (n_fft - c.shape[-1],))), -1)
        shift = il_max - 1
        c = np.roll(c, -shift, -1)
        h = np.fft.ifft(np.exp(np.fft.fft(c, n=n_fft))).real          # c2mpir.py:98-101 with ir_length = n_fft
        h = np.roll(h, shift, -1)[..., : il_min + il_max - 1]
        return zerodf(x, h, P, zeroth_index=shift)
    if mode == "freq-domain":
        L, nfft, n_fft = kw.get("frame_length", 400), kw.get("fft_length", 512), kw.get("n_fft", 512)
        win = kw.get("window", "blackman")
        Hs = []
        for c in (mc_min, mc_max):
            if ignore_gain:
                bq = gnorm(mc2b(c, alpha), gamma)
                bq[..., 0] = 0
                c = b2mc(bq, alpha)
            Hs.append(mgc2sp(c, nfft, alpha, gamma, n_fft=n_fft, out_format="complex"))
        H = Hs[0] * np.conj(Hs[1])
        w = window_table(L, win)
        X = stft(x, L, P, nfft, window=win, out_format="complex")
        return istft(H * X, L, P, w=w, out_length=x.shape[-1])
    raise ValueError(f"mode {mode} is not supported.")


def mlsa(x, mc, P, alpha=0.0, gamma=0.0, ignore_gain=False, phase="minimum", mode="multi-stage", **kw):
    """PseudoMGLSADigitalFilter (mglsadf.py:126-253) for phase in {minimum, maximum, zero}:
    multi-stage (mglsadf.py:351-386), single-stage (:486-526), freq-domain (:613-644)."""
    x, mc = np.asarray(x, dtype=np.float64), np.asarray(mc, dtype=np.float64)
    M = mc.shape[-1] - 1
    if mode == "multi-stage":
        taylor_order, cep_order, n_fft = kw.get("taylor_order", 20), kw.get("cep_order", 199), kw.get("n_fft", 512)
        if alpha == 0 and gamma == 0:
            cep_order = M
        c = mgc2mgc(mc, cep_order, in_alpha=alpha, in_gamma=gamma, n_fft=n_fft)
        c0 = c[..., :1]
        c = np.concatenate((np.zeros_like(c0), c[..., 1:]), -1)
        z0 = 0
        if phase == "maximum":
            c, z0 = c[..., ::-1], cep_order
        elif phase == "zero":
            c, z0 = _mirror(c, half=True), cep_order
        y = x.copy()
        cur = x
        fact = 1.0
        for i in range(1, taylor_order + 1):
            cur = zerodf(cur, c, P, zeroth_index=z0) * (1.0 / i)   # weights[i] = (1/i!) / (1/(i-1)!)
            y = y + cur
        if not ignore_gain:
            y = y * np.exp(linear_intpl(c0, P))[..., 0]
        return y
    if mode == "single-stage":
        ir_length, n_fft = kw.get("ir_length", 2000), kw.get("n_fft", 4096)
        if phase in ("minimum", "maximum"):
            h = mgc2mgc(mc, ir_length - 1, in_alpha=alpha, in_gamma=gamma, out_gamma=1, out_mul=True, n_fft=n_fft)
            if ignore_gain:
                h = h / h[..., :1]
            z0 = 0
            if phase == "maximum":
                h, z0 = h[..., ::-1], ir_length - 1
        else:
            c = mgc2mgc(mc, ir_length - 1, in_alpha=alpha, in_gamma=gamma, n_fft=n_fft)
            c = np.concatenate((c[..., :1], 0.5 * c[..., 1:]), -1)
            if ignore_gain:
                c = np.concatenate((np.zeros_like(c[..., :1]), c[..., 1:]), -1)
            h = np.fft.ifft(np.exp(np.fft.hfft(c, n=n_fft))).real[..., :ir_length]
            h, z0 = _mirror(h), ir_length - 1
        return zerodf(x, h, P, zeroth_index=z0)
    if mode == "freq-domain":
        L, nfft, n_fft = kw.get("frame_length", 400), kw.get("fft_length", 512), kw.get("n_fft", 512)
        win = kw.get("window", "blackman")
        c = mc
        if ignore_gain:
            bq = gnorm(mc2b(mc, alpha), gamma)
            bq[..., 0] = 0
            c = b2mc(bq, alpha)
        H = mgc2sp(c, nfft, alpha, gamma, n_fft=n_fft, out_format="complex")
        if phase == "maximum":
            H = np.conj(H)
        elif phase == "zero":
            H = np.abs(H)
        w = window_table(L, win)
        X = stft(x, L, P, nfft, window=win, out_format="complex")
        return istft(H * X, L, P, w=w, out_length=x.shape[-1])
    raise ValueError(f"mode {mode} is not supported.")
