"""Numerical model of csrc/stft_mfma.hip: 512-point real DFT of a 400-sample windowed frame as
8 decimated 64-point real-input sub-DFTs (one 64 x 64 real matrix product per decimation phase, the
MFMA part) + per-bin twiddle and 8-point DFT across phases (the VALU part).  Checks the row/slot
layout, the two real-only rows of slot 0 and the bin mapping against numpy's rfft, and the
split-precision (binary16 hi/lo, 3 products) error against the parity tolerance of the tests."""
import numpy as np

L, NFFT, R, Q = 400, 512, 8, 64


def a_matrix():
    """rows rho = 2*sigma + part; sigma = 0: (Re S[0], Re S[32]); sigma >= 1: (Re S[sigma], Im S[sigma])."""
    m = np.arange(Q)
    A = np.zeros((64, Q))
    A[0] = 1.0
    A[1] = np.cos(np.pi * m)
    for s in range(1, 32):
        A[2 * s] = np.cos(2 * np.pi * m * s / Q)
        A[2 * s + 1] = -np.sin(2 * np.pi * m * s / Q)
    A[:, 50:] = 0.0  # samples n >= 400 do not exist
    return A


def split16(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def mm3(a, b):
    ah, al = split16(a.astype(np.float32))
    bh, bl = split16(b.astype(np.float32))
    return (ah @ bh + ah @ bl + al @ bh).astype(np.float32)


def dft8(u):  # u: (8, ...) complex, out[j] = sum_p W8^(p j) u[p]
    p = np.arange(8)
    W = np.exp(-2j * np.pi * np.outer(p, p) / 8)
    return np.tensordot(W, u, axes=(1, 0))


def stft_frame_power(xw, split=False, dtype=np.float64):
    """xw: (F, 400) windowed frames -> (F, 257) power."""
    F = xw.shape[0]
    xz = np.zeros((F, NFFT), dtype=np.float64)
    xz[:, :L] = xw
    A = a_matrix()
    out = np.zeros((F, 257))
    # B[m, (f, p)] = xz[f, 8 m + p]
    B = xz.reshape(F, Q, R)  # [f, m, p]
    if split:
        # per-frame power-of-two scale to 2^13 .. 2^14, matrix scaled by 2^10
        amax = np.abs(xw).max(axis=1)
        ex = np.frexp(amax)[1]
        sc = np.ldexp(1.0, 14 - ex)
        Bs = (B * sc[:, None, None]).astype(np.float32)
        S = np.stack([mm3((A * 1024).astype(np.float32), Bs[f]) for f in range(F)]).astype(np.float64)  # [f, rho, p]
        S = S / (1024 * sc[:, None, None])
        S = S.astype(np.float32).astype(np.float64)
    else:
        S = np.einsum("rm,fmp->frp", A, B)
    p = np.arange(8)
    for s in range(32):
        re, im = S[:, 2 * s, :], S[:, 2 * s + 1, :]  # (F, 8)
        if s == 0:
            G = dft8((re + 1j * im).T)  # (8, F)
            Aj = 0.5 * (G + np.conj(G[(-np.arange(8)) % 8]))  # DFT8 of the real sequence a_p = Re S_p[0]
            for j in range(5):
                out[:, 64 * j] = np.abs(Aj[j]) ** 2
            H = dft8((np.exp(-2j * np.pi * p / 16)[None, :] * im).T)
            for j in range(4):
                out[:, 32 + 64 * j] = np.abs(H[j]) ** 2
        else:
            tw = np.exp(-2j * np.pi * p * s / NFFT)
            Y = dft8((tw[None, :] * (re + 1j * im)).T)
            for j in range(4):
                out[:, s + 64 * j] = np.abs(Y[j]) ** 2
            for j in range(4, 8):
                out[:, (64 - s) + 64 * (7 - j)] = np.abs(Y[j]) ** 2
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    z = np.load("tests/golden/tables.npz")
    w = [z[k] for k in z.files if "blackman" in k.lower() or "window" in k.lower()]
    win = w[0].astype(np.float64) if w else np.blackman(L)
    x = rng.standard_normal((64, L))
    x[1] *= 1e-6
    x[2] *= 1e6
    x[3, 200:] *= 1e-5
    xw = x * win[:L]
    ref = np.abs(np.fft.rfft(xw, n=NFFT)) ** 2
    y = stft_frame_power(xw)
    print("exact model  max rel-to-rowmax err", np.max(np.abs(y - ref) / ref.max(1, keepdims=True)))
    ys = stft_frame_power(xw.astype(np.float32).astype(np.float64), split=True)
    err = np.abs(ys - ref)
    tol = 1e-4 * ref + 2e-6 * ref.max(1, keepdims=True)
    print("split model  max err/tol", np.max(err / tol), " max err/rowmax", np.max(err / ref.max(1, keepdims=True)))
    y32 = np.abs(np.fft.rfft(xw.astype(np.float32), n=NFFT).astype(np.complex64)) ** 2
    print("numpy f32 fft max err/rowmax", np.max(np.abs(y32 - ref) / ref.max(1, keepdims=True)))
