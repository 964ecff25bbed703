"""numpy model of the big STFT backward (fft_length 1024 / 2048): the forward's decimated 256-point transforms, radix-S combine and real-FFT
split (csrc/stft_pk_big.h), then the adjoint -- pack (csrc/stft_bwd_pk.h), inverse combine, inverse 256-point transforms -- against the analytic
gradient of sum_k gy_k |rfft(xw)_k|^2.  Pins the scale factors and the special pairs of the kernel (csrc/stft_bwd_pk_big.h)."""
import numpy as np
rng = np.random.default_rng(0)
for S in (2, 4):
    C = 256 * S; N = 2 * C; K = C + 1
    L = N - 8
    xw = np.zeros(N); xw[:L] = rng.standard_normal(L)
    gy = rng.standard_normal(K)
    X_ref = np.fft.rfft(xw)
    # analytic gradient: d/dxw_n sum_k gy_k |X_k|^2 = 2 Re sum_k gy_k X_k e^{+i theta k n}
    n = np.arange(N)
    g_ref = np.array([2 * np.real(np.sum(gy * X_ref * np.exp(2j * np.pi * np.arange(K) * nn / N))) for nn in n])
    # ---- forward as the kernel does it ----
    c = xw[0::2] + 1j * xw[1::2]                       # C complex points
    Y = np.stack([np.fft.fft(c[r::S]) * 0.5 for r in range(S)])   # halved (the table)
    WC = lambda e: np.exp(-2j * np.pi * e / C)
    W2C = lambda e: np.exp(-2j * np.pi * e / N)
    WS = lambda e: np.exp(-2j * np.pi * e / S)
    Z = np.zeros(C, complex)
    for kp in range(256):
        for q in range(S):
            Z[kp + 256 * q] = sum(WS(r * q) * WC(r * kp) * Y[r][kp] for r in range(S))
    def split(a, b, W):          # a = Z[k], b = Z[C - k] (halved): X[k], conj(X[C - k])
        Ss, Dd = a + np.conj(b), a - np.conj(b)
        Pp = W * Dd
        return Ss - 1j * Pp, Ss + 1j * Pp
    X = np.zeros(K, complex)
    for k in range(1, C):
        X[k], _ = split(Z[k], Z[C - k], W2C(k))
    X[0] = 2 * (Z[0].real + Z[0].imag); X[C] = 2 * (Z[0].real - Z[0].imag)
    assert np.allclose(X, X_ref), np.abs(X - X_ref).max()
    # ---- backward: pack per pair (k, C - k) ----
    def pack(a, b, W, g1, g2):   # stft_bwd_pk.h: A = g1 X[k], B = conj(g2 X[C-k]); Zin[k] = (A + B) + i Q, Zin[C-k] = conj((A + B) - i Q), Q = conj(W)(A - B)
        X1, Y2 = split(a, b, W)
        A, B = g1 * X1, g2 * Y2
        Q = np.conj(W) * (A - B)
        return (A + B) + 1j * Q, np.conj((A + B) - 1j * Q)
    gsc, gsc0 = 2.0, 4.0
    Zb = np.zeros(C, complex)
    done = set()
    for k in range(1, C):
        if k in done or (C - k) in done: continue
        if k == C - k:
            zk, _ = pack(Z[k], Z[k], W2C(k), gy[k] * gsc, gy[k] * gsc)
            Zb[k] = zk
        else:
            zk, zm = pack(Z[k], Z[C - k], W2C(k), gy[k] * gsc, gy[C - k] * gsc)
            Zb[k], Zb[C - k] = zk, zm
        done.add(k); done.add(C - k)
    zk, _ = pack(Z[0], Z[0], 1.0, gy[0] * gsc0, gy[C] * gsc0)     # the pair (0, C): real-valued bins, the kernel's lane-0 special
    Zb[0] = zk
    # ---- inverse combine, inverse 256-point transforms (unnormalised, through the halved table: x 1/2) ----
    Yb = np.zeros((S, 256), complex)
    for kp in range(256):
        for r in range(S):
            Yb[r][kp] = np.conj(WC(r * kp)) * sum(np.conj(WS(r * q)) * Zb[kp + 256 * q] for q in range(S))
    g = np.zeros(N)
    for r in range(S):
        yb = np.fft.ifft(Yb[r]) * 256 * 0.5           # unnormalised inverse, halved table
        m = np.arange(256)
        g[2 * (S * m + r)] = yb.real
        g[2 * (S * m + r) + 1] = yb.imag
    print(S, "max |g - g_ref| / max |g_ref| =", np.abs(g - g_ref).max() / np.abs(g_ref).max())
