"""Numerical model of the f16-split (hi/lo, 3-product) matrix chains of the mcep Newton step.

Emulates csrc/mcep_mfma.hip variant 16 at matrix level: operands are split into two binary16 pieces,
products hi*hi + hi*lo + lo*hi are accumulated in float32 (float64 matmul of exactly representable
products, rounded once: slightly optimistic on accumulation order only).  Prints the error of the
converged mel-cepstrum against the committed goldens with the parity tolerance of tests/.
"""
import sys
import numpy as np

sys.path.insert(0, ".")
from diffsptk_amd.utils.tables import mcep_matrices

f32, f16 = np.float32, np.float16
L2E2 = -2.885390081777926815


def split(x):
    x = x.astype(f32)
    hi = x.astype(f16)
    lo = (x - hi.astype(f32)).astype(f16)
    return hi.astype(np.float64), lo.astype(np.float64)


def mm3(a, b, terms=3):
    ah, al = split(a)
    bh, bl = split(b)
    r = ah @ bh + ah @ bl + al @ bh
    if terms == 4:
        r = r + al @ bl
    return r.astype(f32)


def solve_rows(rt, av):
    M1 = 25
    F = rt.shape[0]
    i = np.arange(M1)
    T = rt[:, np.abs(i[:, None] - i[None, :])]
    Hk = rt[:, i[:, None] + i[None, :]]
    A = (T + Hk).astype(f32)
    b = (rt[:, :M1] - av[None, :]).astype(f32)
    return np.linalg.solve(A.astype(np.float64), b.astype(np.float64)[..., None])[..., 0].astype(f32)


def mcep(X, n_iter=10, mode="f16x3", c0_special=True, sD=2.0 ** 4, sM=2.0 ** 10, sE=2.0 ** 16):
    G, D, E, av, *_ = mcep_matrices(512, 24, 0.42)
    G32, D32, E32, av32 = (m.astype(f32) for m in (G, D, E, av))
    logx = np.log2(X.astype(f32)).astype(f32)
    mc = ((logx.astype(np.float64) @ (np.log(2.0) * G32.astype(np.float64)))).astype(f32)
    Dn = (f32(L2E2) * D32).astype(f32)
    for _ in range(n_iter):
        if mode == "f32":
            d = (mc.astype(np.float64) @ Dn.astype(np.float64)).astype(f32)
            t = logx + d
            e = np.exp2(t).astype(f32)
            rt = (e.astype(np.float64) @ E32.astype(np.float64)).astype(f32)
        else:
            Ds = (Dn * f32(sD)).astype(f32)
            mcs = (mc * f32(sM)).astype(f32)
            if c0_special:
                c0 = (mc[:, :1].astype(np.float64) * Ds[0:1, :].astype(np.float64))  # fp32 FMA into acc init
                Dz = Ds.copy(); Dz[0] = 0
                acc = (mm3(mcs, Dz).astype(np.float64) + c0 * sM).astype(f32)
            else:
                acc = mm3(mcs, Ds)
            t = (acc.astype(np.float64) / (sD * sM) + logx).astype(f32)
            mi = np.ceil(t.max(axis=1, keepdims=True))
            ep = np.exp2((t + (15 - mi)).astype(f32)).astype(f32)
            rts = mm3(ep, (E32 * f32(sE)).astype(f32))
            rt = np.ldexp(rts, (mi - 15 - 16).astype(np.int32)).astype(f32)
        x = solve_rows(rt, av32)
        mc = (mc + x).astype(f32)
    return mc


def report(name, y, ref64):
    err = np.abs(y - ref64)
    tol = 2e-5 + 1e-4 * np.abs(ref64)
    print(f"{name:28s} max|err| {err.max():.3e}  max err/tol {np.max(err / tol):.3f}  rms {np.sqrt((err**2).mean()):.2e}")


if __name__ == "__main__":
    for fn, key in (("tests/golden/datawav.npz", "stft_power_f32"), ("tests/golden/randn.npz", "stft_power_f32")):
        z = np.load(fn)
        X = z[key].reshape(-1, 257)
        ref = z["mcep_f64"].reshape(-1, 25)
        print(fn, X.shape)
        report("golden f32 vs f64", z["mcep_f32"].reshape(-1, 25), ref)
        report("model f32 chain", mcep(X, mode="f32"), ref)
        report("model f16x3 (c0 in acc)", mcep(X), ref)
        report("model f16x3 (c0 in mfma)", mcep(X, c0_special=False), ref)
