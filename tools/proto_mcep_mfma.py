"""Lane-level numpy prototype of the MFMA mel-cepstral kernel (16 frames per wave64).

Emulates v_mfma_f32_16x16x4_f32 operand/result layouts (cdna_hip_programming.md section 3):
  A: lane l holds A[i = l & 15][k = l >> 4]      B: lane l holds B[k = l >> 4][j = l & 15]
  C/D: lane l, reg r holds D[row = (l >> 4) * 4 + r][col = l & 15]
and checks the whole per-wave dataflow of diffsptk_amd/csrc/mcep_mfma.hip against a plain
numpy evaluation.  Run: python tools/proto_mcep_mfma.py
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from diffsptk_amd.utils import tables

LANES = np.arange(64)
N_OF = LANES & 15   # frame within the wave
G_OF = LANES >> 4   # lane group


def mfma(a, b, c):
    """a, b: (64,) per-lane operands; c: (64, 4) accumulators -> (64, 4)."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    A[LANES & 15, LANES >> 4] = a
    B[LANES >> 4, LANES & 15] = b
    Dm = A @ B
    out = c.copy()
    for r in range(4):
        out[:, r] += Dm[(LANES >> 4) * 4 + r, LANES & 15]
    return out


def run_wave(X16, G, D, E, av, n_iter, M):
    """X16: (16, 257) power spectra of the wave's 16 frames."""
    M1, M2, H = M + 1, 2 * M + 1, 256
    assert M1 == 25
    KS = 7
    # ---- operand images (what the kernel builds in LDS / fetches from L2) ----
    Dp = np.zeros((KS * 4, 257)); Dp[:M1] = D
    Dt = np.zeros((16, KS, 64))       # Dt[mt][ks][lane] = D[4ks + (l>>4)][mt*16 + (l&15)]
    for mt in range(16):
        for ks in range(KS):
            Dt[mt, ks] = Dp[4 * ks + G_OF, mt * 16 + N_OF]
    Et = np.zeros((3, 16, 4, 64))     # Et[it][mt][r][lane] = E[mt*16 + (l>>4)*4 + r][it*16 + (l&15)]
    for it in range(3):
        for mt in range(16):
            for r in range(4):
                Et[it, mt, r] = E[mt * 16 + G_OF * 4 + r, it * 16 + N_OF]
    E256t = np.zeros((3, 64))         # extra K-step: only k-slot 0 carries bin 256
    for it in range(3):
        E256t[it] = np.where(G_OF == 0, E[256, it * 16 + N_OF], 0.0)
    E48 = np.zeros((16, 4, 64))       # E48[mt][r][lane] = E[mt*16 + g*4 + r][48]
    for mt in range(16):
        for r in range(4):
            E48[mt, r] = E[mt * 16 + G_OF * 4 + r, 48]
    Gp = np.zeros((257, 32)); Gp[:, :M1] = G
    Gt = np.zeros((2, 16, 4, 64))
    for it in range(2):
        for mt in range(16):
            for r in range(4):
                Gt[it, mt, r] = Gp[mt * 16 + G_OF * 4 + r, it * 16 + N_OF]
    D256 = np.zeros(KS * 4); D256[:M1] = D[:, 256]
    # ---- per-lane registers ----
    logx = np.zeros((16, 4, 64))
    for mt in range(16):
        for r in range(4):
            logx[mt, r] = np.log(X16[N_OF, mt * 16 + G_OF * 4 + r])
    logx256 = np.log(X16[N_OF, 256])
    # mc0^T = G^T logx^T : 2 tiles of 16 outputs
    accG = np.zeros((2, 64, 4))
    for it in range(2):
        for mt in range(16):
            for r in range(4):
                accG[it] = mfma(Gt[it, mt, r], logx[mt, r], accG[it])
        accG[it] = mfma(np.where(G_OF == 0, Gp[256, it * 16 + N_OF], 0.0), np.where(G_OF == 0, logx256, 0.0), accG[it])
    # accG[it][lane][r] = mc0[frame n][coef it*16 + 4g + r]  -> through LDS -> mcB[ks] = mc[4ks + g]
    lds_mc = np.zeros((16, 32))
    for it in range(2):
        for r in range(4):
            lds_mc[N_OF, it * 16 + 4 * G_OF + r] = accG[it][:, r]
    mcB = np.zeros((KS, 64))
    for ks in range(KS):
        mcB[ks] = lds_mc[N_OF, 4 * ks + G_OF]
    mcB[6] = np.where(4 * 6 + G_OF < M1, mcB[6], 0.0)
    hist = [mcB.copy()]
    for _ in range(n_iter):
        # step A: d^T = D^T mc^T
        e = np.zeros((16, 64, 4))
        for mt in range(16):
            acc = np.zeros((64, 4))
            for ks in range(KS):
                acc = mfma(Dt[mt, ks], mcB[ks], acc)
            for r in range(4):
                e[mt][:, r] = np.exp(logx[mt, r] - 2.0 * acc[:, r])
        part = sum(mcB[ks] * D256[4 * ks + G_OF] for ks in range(KS))
        d256 = np.zeros(64)
        for l in range(64):  # xor-16 / xor-32 butterfly = sum over the 4 lanes of a frame
            d256[l] = part[l & 15] + part[(l & 15) + 16] + part[(l & 15) + 32] + part[(l & 15) + 48]
        e256 = np.exp(logx256 - 2.0 * d256)
        # step B: rt^T = E^T e^T (48 outputs on MFMA, output 48 on VALU)
        accB = np.zeros((3, 64, 4))
        for it in range(3):
            for mt in range(16):
                for r in range(4):
                    accB[it] = mfma(Et[it, mt, r], e[mt][:, r], accB[it])
            accB[it] = mfma(E256t[it], np.where(G_OF == 0, e256, 0.0), accB[it])
        p48 = sum(e[mt][:, r] * E48[mt, r] for mt in range(16) for r in range(4))
        rt48 = np.zeros(64)
        for l in range(64):
            rt48[l] = sum(p48[(l & 15) + 16 * q] for q in range(4)) + e256[l] * E[256, 48]
        # rt to LDS (per frame 49 values + zero pad), reflected copy rr[27 + d] = r[|d|]
        S = 68
        rt_lds = np.zeros((16, S))
        for it in range(3):
            for r in range(4):
                rt_lds[N_OF, it * 16 + 4 * G_OF + r] = accB[it][:, r]
        rt_lds[N_OF, 48] = rt48
        rr_lds = np.zeros((16, S))
        for n in range(16):
            for d in range(-27, 28):
                rr_lds[n, 27 + d] = rt_lds[n, abs(d)]
        # build local rows i = g + 4m: A[i][j] = r[|i-j|] + rt[i+j]
        a = np.zeros((7, 25, 64)); b = np.zeros((7, 64))
        for m in range(7):
            i = G_OF + 4 * m
            valid = i < M1
            for j in range(25):
                t = 4 * m - j
                val = rr_lds[N_OF, 27 + G_OF + t] + rt_lds[N_OF, G_OF + 4 * m + j]
                a[m, j] = np.where(valid, val, 0.0)
            avg = np.array([av[ii] if ii < M1 else 0.0 for ii in i])
            b[m] = np.where(valid, rt_lds[N_OF, np.minimum(i, 67)] - avg, 0.0)
        # forward elimination, row-cyclic over the 4 lane groups
        for k in range(25):
            gk, mk = k & 3, k >> 2
            src = N_OF + 16 * gk
            prow = {j: a[mk, j][src] for j in range(k, 25)}
            pb = b[mk][src]
            inv = 1.0 / prow[k]
            for m in range(mk, 7):
                f = a[m, k] * inv
                if m == mk:
                    f = np.where(G_OF > gk, f, 0.0)
                for j in range(k + 1, 25):
                    a[m, j] = a[m, j] - f * prow[j]
                b[m] = b[m] - f * pb
        # back substitution: x_k from the owner lane, broadcast, column update
        x = [None] * 25
        for k in range(24, -1, -1):
            gk, mk = k & 3, k >> 2
            src = N_OF + 16 * gk
            xk = (b[mk] / a[mk, k])[src]
            x[k] = xk
            for m in range(0, mk + 1):
                b[m] = b[m] - a[m, k] * xk
        for ks in range(KS):
            sel = np.zeros(64)
            for gg in range(4):
                if 4 * ks + gg < M1:
                    sel = np.where(G_OF == gg, x[4 * ks + gg], sel)
            mcB[ks] = mcB[ks] + sel
        hist.append(mcB.copy())
    mc = np.zeros((16, 28))
    for ks in range(KS):
        mc[N_OF, 4 * ks + G_OF] = mcB[ks]
    return mc[:, :M1]


def reference(X, G, D, E, av, n_iter, M):
    logx = np.log(X)
    mc = logx @ G
    for _ in range(n_iter):
        e = np.exp(logx - 2 * (mc @ D))
        rt = e @ E
        out = np.zeros_like(mc)
        for f in range(X.shape[0]):
            A = np.array([[rt[f, abs(i - j)] + rt[f, i + j] for j in range(M + 1)] for i in range(M + 1)])
            out[f] = np.linalg.solve(A, rt[f, : M + 1] - av)
        mc = mc + out
    return mc


if __name__ == "__main__":
    G, D, E, av = tables.mcep_matrices(512, 24, 0.42)[:4]
    g = np.load("tests/golden/datawav.npz")
    X = g["stft_power_f64"][100:116]
    got = run_wave(X, G, D, E, av, 3, 24)
    want = reference(X, G, D, E, av, 3, 24)
    print("max abs diff vs plain numpy:", np.abs(got - want).max())
    assert np.abs(got - want).max() < 1e-9
    full = reference(X, G, D, E, av, 10, 24)
    print("vs reference golden (10 iters):", np.abs(full - g["mcep_f64"][100:116]).max())
    print("proto_mcep_mfma OK")
