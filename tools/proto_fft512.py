"""Lane-level numpy prototype of the tuned rFFT-512 kernel (16 lanes per frame).

Mirrors diffsptk_amd/csrc/stft.hip step by step so the index math can be checked on CPU
(there is no GPU in the build container).  Run: python tools/proto_fft512.py
"""
import numpy as np


def dft4(a0, a1, a2, a3):
    # forward, W4 = -i
    s02, d02 = a0 + a2, a0 - a2
    s13, d13 = a1 + a3, a1 - a3
    return s02 + s13, d02 - 1j * d13, s02 - s13, d02 + 1j * d13


W16 = np.exp(-2j * np.pi * np.arange(16) / 16)


def fft16(v):
    """v: list of 16 complex -> list of 16 complex X[k] in natural order (static indices)."""
    B = [[None] * 4 for _ in range(4)]
    for n0 in range(4):
        B[n0] = list(dft4(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]))  # B[n0][q]
    for n0 in range(1, 4):
        for q in range(1, 4):
            B[n0][q] = B[n0][q] * W16[n0 * q]
    X = [None] * 16
    for q in range(4):
        y = dft4(B[0][q], B[1][q], B[2][q], B[3][q])
        for p in range(4):
            X[q + 4 * p] = y[p]
    return X


def rfft512_lanes(xw):
    """xw: 512 real samples (windowed, zero padded).  Emulates 16 lanes."""
    z = xw[0::2] + 1j * xw[1::2]  # 256 complex
    # step 1: lane j holds z[j + 16*m1]; 16-point DFT over m1
    A = np.zeros((16, 16), complex)  # A[j][k1]
    for j in range(16):
        A[j] = fft16([z[j + 16 * m1] for m1 in range(16)])
    # step 2: twiddle W256^(j*k1)
    for j in range(16):
        for k1 in range(16):
            A[j, k1] *= np.exp(-2j * np.pi * j * k1 / 256)
    # step 3: transpose through LDS: lane k1 gets A[j][k1] for j = 0..15
    Tm = A.T.copy()  # Tm[k1][j]
    # step 4: lane k1: DFT over j -> Z[k1 + 16*k0]
    Z = np.zeros(256, complex)
    for k1 in range(16):
        out = fft16(list(Tm[k1]))
        for k0 in range(16):
            Z[k1 + 16 * k0] = out[k0]
    # step 5: real post-processing, k = 0..256
    k = np.arange(257)
    Zk = Z[k % 256]
    Zm = np.conj(Z[(256 - k) % 256])
    W = np.exp(-2j * np.pi * k / 512)
    X = 0.5 * (Zk + Zm) - 0.5j * W * (Zk - Zm)
    return X


rng = np.random.default_rng(0)
for trial in range(3):
    v = rng.standard_normal(16) + 1j * rng.standard_normal(16)
    assert np.allclose(fft16(list(v)), np.fft.fft(v)), "fft16"
    x = rng.standard_normal(512)
    assert np.allclose(rfft512_lanes(x), np.fft.rfft(x)), "rfft512"
print("proto_fft512 OK")
