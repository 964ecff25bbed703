"""GPU parity tests of SURVEY.md section 8(f) rows 3-4 (run with -m gpu on an MI355X): mel-generalized cepstral
analysis and the cepstrum conversions around it, against outputs of the reference (tests/golden/synth.npz) and the
numpy oracle.  Tolerances: float64 rtol 1e-5 / atol 1e-8 (tests/utils.py:66-72 of the reference); float32 as stated
per test."""
import math

import numpy as np
import pytest
import scipy.signal
import torch

import diffsptk_amd as dsp
from diffsptk_amd import _lib, functional as F, ops
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
F64 = dict(rtol=1e-5, atol=1e-8)


def dev(a, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def close(a, b, rtol, atol):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_mc2b_b2mc_gnorm_ignorm(golden):
    g = golden("synth")
    mc = dev(g["mc"])
    for a in (0.1, 0.42):
        b = dsp.MelCepstrumToMLSADigitalFilterCoefficients(8, a, dtype=torch.float64, device=DEV)(mc)
        close(host(b), g[f"mc2b_{a}"], **F64)
        close(host(F.mc2b(mc, a)), g[f"mc2b_{a}"], **F64)
        close(host(dsp.MLSADigitalFilterCoefficientsToMelCepstrum(8, a, dtype=torch.float64, device=DEV)(b)), g[f"b2mc_{a}"], **F64)
        close(host(F.b2mc(F.mc2b(mc, a), a)), g["mc"], **F64)          # round trip
    x = dev(g["gn_in"])
    for gam in (0.0, -0.5, -1.0):
        y = dsp.GeneralizedCepstrumGainNormalization(8, gam)(x)
        close(host(y), g[f"gnorm_{gam}"], **F64)
        close(host(F.ignorm(y, gam)), g[f"ignorm_{gam}"], **F64)
    close(host(F.gnorm(x, c=2)), g["gnorm_-0.5"], **F64)               # c = 2 stages <=> gamma = -1/2
    b = F.mc2b(dev(g["mcep512"]), 0.42)
    close(host(b), g["mc2b512"], **F64)
    with pytest.raises(ValueError):
        dsp.MelCepstrumToMLSADigitalFilterCoefficients(8, 1.0)


def test_mgc2mgc_and_mgc2sp(golden):
    g = golden("synth")
    x = dev(g["mgc_in"])
    for i, c in enumerate(g["mgc2mgc_cases"]):
        ia, oa, ig, og, inn, on, im, om, oo = c
        m = dsp.MelGeneralizedCepstrumToMelGeneralizedCepstrum(8, int(oo), in_alpha=ia, out_alpha=oa, in_gamma=ig, out_gamma=og,
                                                                in_norm=bool(inn), out_norm=bool(on), in_mul=bool(im),
                                                                out_mul=bool(om), n_fft=128, dtype=torch.float64, device=DEV)
        close(host(m(x)), g[f"mgc2mgc_{i}"], **F64)
    for gam in (0.0, -0.5):
        for fmt in (0, 1, 2, 3, 4, 5, 6, "complex"):
            y = F.mgc2sp(x, 32, alpha=0.1, gamma=gam, n_fft=128, out_format=fmt)
            ref = g[f"mgc2sp_{gam}_{fmt}"]
            if fmt == "complex":
                y = torch.view_as_real(y)
            close(host(y), ref, **F64)
    xg = x.clone().requires_grad_(True)
    F.mgc2sp(xg, 32, alpha=0.1, gamma=-0.5, n_fft=128, out_format=3).sum().backward()
    ref = g["mgc2sp_grad_-0.5_3"]
    assert np.abs(host(xg.grad) - ref).max() < 1e-8 * np.abs(ref).max()
    # the BASELINE geometry: spectral envelope of the mel-cepstra of data.wav, float64 and float32
    mc = dev(g["mcep512"])
    close(host(F.mgc2sp(mc, 512, alpha=0.42)), g["mgc2sp512_power"], **F64)
    y32 = host(F.mgc2sp(mc.float(), 512, alpha=0.42)).astype(np.float64)
    assert np.abs(y32 / g["mgc2sp512_power"] - 1).max() < 2e-4        # exp of a float32 log spectrum of magnitude <= 20
    # analysis -> synthesis consistency on the device: exp(2 Re rfft(freqt(mc))) follows the spectrum the cepstra came from
    stft = dsp.STFT(400, 80, 512, device=DEV)
    mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=DEV)
    X = stft(torch.randn(2, 8000, generator=torch.Generator().manual_seed(2)).to(DEV))
    env = F.mgc2sp(mcep(X), 512, alpha=0.42)
    assert env.shape == X.shape and bool(torch.isfinite(env).all())
    assert float((torch.log(env).mean() - torch.log(X).mean()).abs()) < 0.7   # unbiased log-spectral fit up to the gain convention


def test_thsolve_kernel_vs_numpy_and_gradcheck():
    rng = np.random.default_rng(0)
    for n, Fr in ((8, 37), (24, 200), (1, 3), (64, 5)):
        p = rng.standard_normal((Fr, n))
        p[:, 0] += 2.0 * n
        q = 0.3 * rng.standard_normal((Fr, 2 * n - 1))
        r = rng.standard_normal((Fr, n))
        ii = np.arange(n)
        A = p[:, np.abs(ii[:, None] - ii[None, :])] + q[:, ii[:, None] + ii[None, :]]
        ref = np.linalg.solve(A, r[..., None])[..., 0]
        for dt, tol in ((torch.float64, 1e-9), (torch.float32, 2e-4)):
            gsol = ops.ThSolveFn.apply(dev(p, dt), dev(q, dt), dev(r, dt))
            # cepstral order 24 in float32: the unpivoted quad-layout solve shared with the mel-cepstral kernels
            # (round 6: the kernel depends on (order, dtype) ALONE -- float32 orders 2 .. 55 run the quad-layout template whatever the
            #  number of systems; float64, order 1 and orders above 55 keep the one-wave-per-system pivoted kernel)
            want = "th_solve_fwd"
            if dt == torch.float32 and n == 24:
                want = "th_solve_quad_fwd"
            elif dt == torch.float32 and 2 <= n <= 55:
                want = "th_solve_quadn_fwd"
            assert _lib.last_kernel() == want, (n, dt, _lib.last_kernel())
            assert np.abs(host(gsol) - ref).max() < tol * max(1.0, np.abs(ref).max()), (n, dt)
    # bench-size batch of 24 x 24 systems with the spectrum-like structure of the analysis (positive definite, condition
    # numbers up to 1e4): every system against float64
    Fr, n = 51200 + 7, 24
    w = np.exp(rng.standard_normal((Fr, 64)) * 1.5)                       # positive "spectra"
    om = np.pi * (np.arange(64) + 0.5) / 64
    ii = np.arange(n)
    pk = (w[:, None, :] * np.cos(om[None, None, :] * ii[None, :, None])).sum(-1)            # toeplitz part
    qk = 0.5 * (w[:, None, :] * np.cos(om[None, None, :] * np.arange(2 * n - 1)[None, :, None])).sum(-1)   # hankel part
    rk = rng.standard_normal((Fr, n))
    gsol = ops.ThSolveFn.apply(dev(pk, torch.float32), dev(qk, torch.float32), dev(rk, torch.float32))
    assert _lib.last_kernel() == "th_solve_quad_fwd"
    sel = rng.choice(Fr, 400, replace=False)
    sel[:3] = [0, Fr - 1, Fr - 7]
    A = pk[sel][:, np.abs(ii[:, None] - ii[None, :])] + qk[sel][:, ii[:, None] + ii[None, :]]
    ref = np.linalg.solve(A, rk[sel][..., None])[..., 0]
    err = np.abs(host(gsol)[sel] - ref).max(-1) / np.abs(ref).max(-1)
    assert err.max() < 5e-3 and np.median(err) < 2e-5, (err.max(), np.median(err))
    assert bool(torch.isfinite(gsol).all())
    p = dev(p[:4, :6] + 3.0).requires_grad_(True)
    q = dev(q[:4, :11]).requires_grad_(True)
    r = dev(r[:4, :6]).requires_grad_(True)
    assert torch.autograd.gradcheck(ops.ThSolveFn.apply, (p, q, r), eps=1e-6, atol=1e-7, rtol=1e-5)


@pytest.mark.parametrize("gamma", [-0.5, -1.0])
@pytest.mark.parametrize("n_iter", [0, 3])
def test_mgcep_reference_grid(golden, gamma, n_iter):
    """tests/test_mgcep.py:24-49 of the reference: M = 8, L = 32, alpha = 0.1 -- outputs and gradients."""
    g = golden("synth")
    X = dev(g["mgcep_X"]).requires_grad_(True)
    m = dsp.MelGeneralizedCepstralAnalysis(fft_length=32, cep_order=8, alpha=0.1, gamma=gamma, n_iter=n_iter,
                                           dtype=torch.float64, device=DEV)
    y = m(X)
    close(host(y), g[f"mgcep_{gamma}_{n_iter}"], **F64)
    close(host(F.mgcep(X.detach(), 8, alpha=0.1, gamma=gamma, n_iter=n_iter)), g[f"mgcep_{gamma}_{n_iter}"], **F64)
    (y * torch.linspace(-1, 1, 9, dtype=torch.float64, device=DEV)).sum().backward()
    ref = g[f"mgcep_grad_{gamma}_{n_iter}"]
    assert np.abs(host(X.grad) - ref).max() < 1e-6 * np.abs(ref).max()


@pytest.mark.parametrize("M,alpha,gamma", [(4, 0.5, -0.3), (6, 0.42, -0.5), (4, 0.5, -1.0)])
@pytest.mark.parametrize("n_iter", [0, 1])
def test_mgcep_zero_iterations_keeps_the_gain_of_the_first_step(golden, M, alpha, gamma, n_iter):
    """mgcep.py:240-249 of the reference: with n_iter = 0 (the reference default) and gamma not in {-1, 0} the zeroth coefficient is
    the gain of the gamma = -1 Newton step, only b1 comes from b2b.  A small order with a strong alpha makes the difference 1.5e-3
    (the reference's own grid, M = 8 / alpha = 0.1, hides it at 7.6e-11).  Both the path without a gradient (the gain and the
    joining as one launch) and the differentiable path, outputs and gradients (tests/golden/make_golden_r5.py)."""
    g = golden("r5")
    ref = g[f"mgcep0_{M}_{alpha}_{gamma}_{n_iter}"]
    for dt, tol in ((torch.float64, F64), (torch.float32, dict(rtol=2e-4, atol=2e-5))):
        m = dsp.MelGeneralizedCepstralAnalysis(fft_length=32, cep_order=M, alpha=alpha, gamma=gamma, n_iter=n_iter, dtype=dt, device=DEV)
        with torch.no_grad():
            close(host(m(dev(g["mgcep0_X"], dt))), ref, **tol)
        X = dev(g["mgcep0_X"], dt).requires_grad_(True)
        y = m(X)
        close(host(y), ref, **tol)
        if dt == torch.float64:
            (y * torch.linspace(-1, 1, M + 1, dtype=dt, device=DEV)).sum().backward()
            gref = g[f"mgcep0_grad_{M}_{alpha}_{gamma}_{n_iter}"]
            assert np.abs(host(X.grad) - gref).max() < 1e-6 * np.abs(gref).max()


def test_mgcep_speech_512_and_gamma0_route(golden):
    g = golden("synth")
    X = dev(g["mgcep512_X"])
    m = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, c=3, n_iter=5, dtype=torch.float64, device=DEV)
    close(host(m(X)), g["mgcep512_c3_5"], **F64)
    m32 = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, c=3, n_iter=5, device=DEV)
    y32 = host(m32(X.float())).astype(np.float64)
    # float32 through pow(D, 3) on speech spectra: measured 2.9e-4 (coefficients up to 53; tools/measure_tolerances.py)
    assert np.abs(y32 - g["mgcep512_c3_5"]).max() < 1e-3
    # gamma = 0 routes to the tuned mel-cepstral kernel (mgcep.py:97-105)
    m0 = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=0, n_iter=10, device=DEV)
    y0 = m0(X.float())
    assert _lib.last_kernel().startswith("mcep_mfma_fwd")
    close(host(y0), g["mcep512"], 1e-4, 5e-5)
    Xs = (torch.randn(2, 17, dtype=torch.float64, generator=torch.Generator().manual_seed(0)).square() + 0.1).to(DEV).requires_grad_(True)
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=32, cep_order=4, alpha=0.2, gamma=-0.5, n_iter=2, dtype=torch.float64, device=DEV)
    assert torch.autograd.gradcheck(mg, (Xs,), eps=1e-6, atol=1e-6, rtol=1e-4)
    with pytest.raises(ValueError):
        dsp.MelGeneralizedCepstralAnalysis(fft_length=32, cep_order=4, gamma=0.5)


# ----------------------------------------------------------------------------- MLSA synthesis filter (8(f) row 4)
def test_zerodf_and_linear_intpl(golden):
    g = golden("mlsa")
    for z0 in (0, 2, 4):
        for ig in (0, 1):
            if ig and z0 == 2:
                continue
            x, b = dev(g["zdf_x"]).requires_grad_(True), dev(g["zdf_b"]).requires_grad_(True)
            m = dsp.AllZeroDigitalFilter(4, 4, ignore_gain=bool(ig), zeroth_index=z0, mode="efficient")
            y = m(x, b)
            assert _lib.last_kernel() == "zerodf_fwd"
            close(host(y), g[f"zdf_{z0}_{ig}"], **F64)
            (y * torch.linspace(-1, 1, 24, dtype=torch.float64, device=DEV)).sum().backward()
            close(host(x.grad), g[f"zdf_gx_{z0}_{ig}"], 1e-8, 1e-10)
            close(host(b.grad), g[f"zdf_gb_{z0}_{ig}"], 1e-8, 1e-10)
    close(host(dsp.LinearInterpolation(3)(dev(g["intpl_in"]))), g["intpl_out"], **F64)
    x = torch.randn(2, 12, dtype=torch.float64, generator=torch.Generator().manual_seed(1)).to(DEV).requires_grad_(True)
    b = (torch.randn(2, 4, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(2)) + torch.tensor([2.0, 0, 0])).to(DEV).requires_grad_(True)
    for z0, ig in ((0, True), (1, False), (2, True)):
        assert torch.autograd.gradcheck(lambda a, c: ops.ZerodfFn.apply(a, c, 3, z0, ig), (x, b), eps=1e-6, atol=1e-7, rtol=1e-5)
    # a long filter in float32 against the float64 oracle (the single-stage filter runs 200 .. 2000 taps)
    rng = np.random.default_rng(5)
    xl, bl = rng.standard_normal((3, 800)), 0.05 * rng.standard_normal((3, 10, 400))
    yl = host(ops.ZerodfFn.apply(dev(xl, torch.float32), dev(bl, torch.float32), 80, 0, False))
    assert _lib.last_kernel() == "zerodf_rows_fwd"   # M >= 16, P % 4 == 0: several frames per workgroup, packed multiply-adds
    ref = O.zerodf(xl, bl, 80)
    assert np.abs(yl - ref).max() < 2e-5 * np.abs(ref).max()
    # every shape class of the long-filter kernels: tap counts that do / do not divide by 4, P in {5, 80, 128, 160}, look-ahead
    # taps, gain normalisation on either end tap, more / fewer frames than a workgroup takes, float64; a frame period that
    # is not a multiple of 4 keeps the one-frame-per-workgroup blocked kernel, short filters the one-thread-per-sample one
    for M_, P_, z0, ig, nfr, kern in ((64, 80, 0, False, 4, "rows"), (199, 80, 199, True, 31, "rows"), (301, 128, 100, False, 5, "rows"),
                                      (70, 5, 0, True, 4, "blocked"), (1999, 80, 0, False, 4, "rows"), (1998, 80, 7, True, 3, "rows"),
                                      (17, 8, 3, False, 70, "rows"), (100, 160, 0, True, 2, "rows"), (202, 80, 0, False, 13, "rows")):
        xs_, bs_ = rng.standard_normal((2, nfr * P_)), 0.05 * rng.standard_normal((2, nfr, M_ + 1))
        bs_[..., 0] += 1.5
        bs_[..., -1] += 1.5
        for dt, tol in ((torch.float64, 1e-11), (torch.float32, 3e-5)):
            out = host(ops.ZerodfFn.apply(dev(xs_, dt), dev(bs_, dt), P_, z0, ig))
            assert _lib.last_kernel() == f"zerodf_{kern}_fwd", (M_, P_, dt)
            ref = O.zerodf(xs_, bs_, P_, ig, z0)
            assert np.abs(out - ref).max() < tol * np.abs(ref).max(), (M_, P_, z0, ig, dt)
    # NaN containment: a NaN sample reaches exactly the outputs whose taps touch it (zero PADDING taps must not multiply it)
    M_, P_, z0 = 17, 8, 3
    xs_, bs_ = rng.standard_normal((1, 40 * P_)), rng.standard_normal((1, 40, M_ + 1))
    xs_[0, 100] = np.nan
    out = host(ops.ZerodfFn.apply(dev(xs_, torch.float32), dev(bs_, torch.float32), P_, z0, False))
    assert _lib.last_kernel() == "zerodf_rows_fwd"
    bad = np.flatnonzero(np.isnan(out[0]))
    assert bad.min() == 100 - z0 and bad.max() == 100 - z0 + M_ and len(bad) == M_ + 1
    # one Taylor stage of the multi-stage filter in one launch: rounded like filter, scale, add
    xt, bt = dev(rng.standard_normal((3, 20 * 80)), torch.float32), dev(0.05 * rng.standard_normal((3, 20, 200)), torch.float32)
    acc = dev(rng.standard_normal((3, 20 * 80)), torch.float32)
    want_cur = ops.ZerodfFn.apply(xt, bt, 80, 0, False) * (1.0 / 3)
    want_sum = acc + want_cur
    cur, ysum = ops.zerodf_taylor(xt, bt, 80, 0, 1.0 / 3, acc.clone())
    assert torch.equal(cur, want_cur) and torch.equal(ysum, want_sum)
    assert ops.zerodf_taylor(xt, bt, 80, 0, 1.0 / 3, acc.clone(), want_y=False)[0] is None
    out = host(ops.ZerodfFn.apply(dev(rng.standard_normal((1, 12))), dev(rng.standard_normal((1, 2, 11))), 6, 0, False))
    assert _lib.last_kernel() == "zerodf_fwd" and np.isfinite(out).all()


@pytest.mark.parametrize("M,P,z0,N", [(199, 80, 0, 31), (64, 80, 64, 4), (17, 8, 3, 70), (301, 128, 100, 5), (199, 80, 199, 13),
                                      (23, 4, 2, 9), (1999, 80, 0, 3), (70, 5, 0, 6),
                                      (24, 256, 0, 6), (24, 252, 24, 5), (24, 124, 0, 7)])   # gb rows that needed fewer frames per workgroup
def test_zerodf_backward_kernels_against_autograd_of_the_definition(M, P, z0, N):
    """The backward of the time-variant FIR (packed multi-frame kernels for M >= 16 and P % 4 == 0, the round-2 kernels
    otherwise) against autograd through the definition written with tensor operations (zerodf.py:207-243: interpolated taps
    times the unfolded signal), float64 to 1e-11 and float32 to 2e-5 of the largest entry: look-ahead taps that are / are not a
    multiple of four, more / fewer frames than a workgroup takes, the clamped last frame."""
    g = torch.Generator().manual_seed(M + P + z0)
    B, T = 2, N * P
    x = torch.randn(B, T, generator=g, dtype=torch.float64)
    b = 0.1 * torch.randn(B, N, M + 1, generator=g, dtype=torch.float64)
    gy = torch.randn(B, T, generator=g, dtype=torch.float64)

    def definition(x, b):
        t = torch.arange(T, device=x.device)
        n0, w = t // P, (t % P).to(x.dtype) / P
        n1 = torch.clamp(n0 + 1, max=N - 1)
        h = b[:, n0] * (1 - w)[None, :, None] + b[:, n1] * w[None, :, None]          # (B, T, M + 1)
        xp = torch.nn.functional.pad(x, (M, M))
        idx = t[:, None] - torch.arange(M + 1, device=x.device)[None, :] + z0 + M    # x[t - k + z0]
        return (h * xp[:, idx]).sum(-1)

    xr, br = x.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    (definition(xr, br) * gy.to(DEV)).sum().backward()
    for dt, tol in ((torch.float64, 1e-11), (torch.float32, 2e-5)):
        xk, bk = x.to(DEV, dt).requires_grad_(True), b.to(DEV, dt).requires_grad_(True)
        y = ops.ZerodfFn.apply(xk, bk, P, z0, False)
        assert float((y.double() - definition(xr, br)).abs().max()) < (1e-11 if dt == torch.float64 else 3e-5) * float(y.abs().max())
        (y * gy.to(DEV, dt)).sum().backward()
        for got, ref in ((xk.grad, xr.grad), (bk.grad, br.grad)):
            assert float((got.double() - ref).abs().max()) < tol * float(ref.abs().max()), (M, P, z0, N, dt)


@pytest.mark.parametrize("M,P,z0,N,order", [(199, 80, 0, 31, 20), (60, 80, 60, 5, 7), (300, 16, 100, 40, 3), (24, 256, 0, 6, 5),
                                            (24, 252, 0, 4, 3)])
def test_taylor_stages_function_equals_the_stage_by_stage_graph(M, P, z0, N, order):
    """ops.ZerodfTaylorFn (one launch per stage forward, one call per stage backward: dsa_zerodf_taylor_fwd / _bwd) against the
    graph autograd builds from the differentiable filter and two element-wise operations per stage (mglsadf.py:356-365):
    outputs bit-identical, gradients with respect to the signal and the coefficients to float64 / float32 rounding."""
    g = torch.Generator().manual_seed(M + order)
    x = torch.randn(2, N * P, generator=g, dtype=torch.float64)
    b = 0.05 * torch.randn(2, N, M + 1, generator=g, dtype=torch.float64)
    gy = torch.randn(2, N * P, generator=g, dtype=torch.float64)
    for dt, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
        xa, ba = x.to(DEV, dt).requires_grad_(True), b.to(DEV, dt).requires_grad_(True)
        assert ops.zerodf_taylor_shapes_ok(xa, ba, P)
        ya = ops.ZerodfTaylorFn.apply(xa, ba, P, z0, order)
        (ya * gy.to(DEV, dt)).sum().backward()
        xb, bb = x.to(DEV, dt).requires_grad_(True), b.to(DEV, dt).requires_grad_(True)
        yb, cur = xb, xb
        for i in range(1, order + 1):
            cur = ops.ZerodfFn.apply(cur, bb, P, z0, False) * (1.0 / i)
            yb = yb + cur
        (yb * gy.to(DEV, dt)).sum().backward()
        assert torch.equal(ya.detach(), yb.detach())
        for got, ref in ((xa.grad, xb.grad), (ba.grad, bb.grad)):
            assert float((got - ref).abs().max()) <= tol * float(ref.abs().max()), (M, P, dt)


@pytest.mark.parametrize("mode", ["multi-stage", "single-stage", "freq-domain"])
def test_mlsa_filter_golden(golden, mode):
    """tests/test_mglsadf.py of the reference: M = 24, P = 80, alpha = 0.42, c in {0, 2}, with / without the gain."""
    g = golden("mlsa")
    params = {"multi-stage": {"taylor_order": 7, "cep_order": 100}, "single-stage": {"ir_length": 200, "n_fft": 512},
              "freq-domain": {"frame_length": 512, "fft_length": 512, "window": "hamming"}}[mode]
    x = dev(g["mlsa_x"])
    for c in (0, 2):
        mc = dev(g[f"mlsa_mc_c{c}"])
        for ig in (0, 1):
            m = dsp.MLSA(24, 80, alpha=0.42, c=c, ignore_gain=bool(ig), phase="minimum", mode=mode, dtype=torch.float64, device=DEV, **params)
            ref = g[f"mlsa_{mode}_c{c}_{ig}"]
            y = host(m(x, mc))
            assert np.abs(y - ref).max() < 1e-7 * np.abs(ref).max(), (mode, c, ig, np.abs(y - ref).max())
    m32 = dsp.MLSA(24, 80, alpha=0.42, mode=mode, device=DEV, **params)
    y32 = host(m32(x.float(), dev(g["mlsa_mc_c0"], torch.float32))).astype(np.float64)
    ref = g[f"mlsa_{mode}_c0_0"]
    assert np.abs(y32 - ref).max() < 2e-4 * np.abs(ref).max()
    p2 = {"multi-stage": {"cep_order": 60, "taylor_order": 7}, "single-stage": {"ir_length": 120, "n_fft": 512},
          "freq-domain": {"frame_length": 512, "fft_length": 512, "window": "hamming"}}[mode]
    for ph in ("maximum", "zero"):
        m = dsp.MLSA(24, 80, alpha=0.42, phase=ph, mode=mode, dtype=torch.float64, device=DEV, **p2)
        ref = g[f"mlsa_{mode}_{ph}"]
        assert np.abs(host(m(x, dev(g["mlsa_mc_c0"]))) - ref).max() < 1e-7 * np.abs(ref).max(), (mode, ph)


def test_mlsa_learnable_taylor_weights(golden):
    """learnable=True of the multi-stage filter (mglsadf.py:344-349, 376-379): one weight per Taylor term, initialised to ones --
    same output as the fixed filter, and a gradient with respect to the weights: d/da_i of sum(gy * y) = sum(gy * K * x_i)."""
    g = golden("mlsa")
    x, mc = dev(g["mlsa_x"]), dev(g["mlsa_mc_c0"])
    kw = dict(alpha=0.42, mode="multi-stage", taylor_order=5, cep_order=60, dtype=torch.float64, device=DEV)
    fixed = dsp.MLSA(24, 80, **kw)
    learn = dsp.MLSA(24, 80, learnable=True, **kw)
    assert isinstance(learn.a, torch.nn.Parameter) and learn.a.shape == (6,) and bool((learn.a == 1).all())
    y0, y1 = fixed(x, mc), learn(x, mc)
    close(host(y1), host(y0), 1e-12, 1e-12)
    gy = torch.randn(y1.shape, generator=torch.Generator().manual_seed(0), dtype=torch.float64).to(DEV)
    (y1 * gy).sum().backward()
    with torch.no_grad():   # the terms x_i K by differences of truncated filters
        ys = [dsp.MLSA(24, 80, **{**kw, "taylor_order": k})(x, mc) for k in range(6)]
        terms = [ys[0]] + [ys[k] - ys[k - 1] for k in range(1, 6)]
        want = torch.stack([(gy * t).sum() for t in terms])
    close(host(learn.a.grad), host(want), 1e-8, 1e-10)
    with torch.no_grad():
        learn.a[2] = 0.5
    assert float((learn(x, mc) - y0).abs().max()) > 0


def test_mlsa_gradients_and_analysis_synthesis(golden):
    g = golden("mlsa")
    x, mc = dev(g["mlsa_x"]).requires_grad_(True), dev(g["mlsa_mc_c0"]).requires_grad_(True)
    m = dsp.MLSA(24, 80, alpha=0.42, mode="multi-stage", taylor_order=7, cep_order=100, dtype=torch.float64, device=DEV)
    m(x, mc).square().sum().backward()
    for got, name in ((x.grad, "mlsa_gx"), (mc.grad, "mlsa_gmc")):
        ref = g[name]
        assert np.abs(host(got) - ref).max() < 1e-6 * np.abs(ref).max(), name
    with pytest.raises(NotImplementedError):
        dsp.MLSA(24, 80, mode="pade-approx")
    with pytest.raises(ValueError):
        dsp.MLSA((24, 24), 80, phase="minimum")
    # README.md:91-93 of the reference: analysis -> synthesis.  White excitation through the filter of the analysed
    # cepstra must carry the spectral envelope the cepstra describe (mgc2sp), frame by frame, within the variance of a
    # periodogram: compared on mel-cepstra of the synthesised signal.
    stft = dsp.STFT(400, 80, 512, device=DEV)
    mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=DEV)
    src = torch.randn(1, 16000, generator=torch.Generator().manual_seed(3)).to(DEV)
    shape = torch.tensor([0.0, 1.2, -0.6, 0.3] + [0.0] * 21, device=DEV)
    mc_t = shape.expand(1, 200, 25).contiguous()
    y = dsp.MLSA(24, 80, alpha=0.42, mode="freq-domain", frame_length=400, fft_length=512, device=DEV)(src, mc_t)
    mc_hat = mcep(stft(y))[:, 20:180].mean(1)
    assert float((mc_hat[0, 1:4] - shape[1:4]).abs().max()) < 0.15


@pytest.mark.parametrize("mode", ["multi-stage", "single-stage", "freq-domain"])
def test_mlsa_mixed_phase_golden(golden, mode):
    """PseudoMGLSADigitalFilter(phase="mixed") (mglsadf.py:144-147, 240-246): mc = c_{-12} .. c_{-1}, c_0 .. c_24, against the
    reference's outputs, float64 and float32; equal orders given as one integer, with the gradients."""
    g, b = golden("mlsa_mixed"), golden("mlsa")
    params = {"multi-stage": {"taylor_order": 7, "cep_order": (40, 60)}, "single-stage": {"ir_length": (80, 120), "n_fft": 512},
              "freq-domain": {"frame_length": 512, "fft_length": 512, "window": "hamming"}}[mode]
    x = dev(b["mlsa_x"])
    for c in (0, 2):
        mc = dev(g[f"mc_c{c}"])
        for ig in (0, 1):
            m = dsp.MLSA((12, 24), 80, alpha=0.42, c=c, ignore_gain=bool(ig), phase="mixed", mode=mode, dtype=torch.float64, device=DEV, **params)
            ref = g[f"{mode}_c{c}_{ig}"]
            y = host(m(x, mc))
            assert np.abs(y - ref).max() < 1e-7 * np.abs(ref).max(), (mode, c, ig, np.abs(y - ref).max())
    m32 = dsp.MLSA((12, 24), 80, alpha=0.42, phase="mixed", mode=mode, device=DEV, **params)
    y32 = host(m32(x.float(), dev(g["mc_c0"], torch.float32))).astype(np.float64)
    ref = g[f"{mode}_c0_0"]
    assert np.abs(y32 - ref).max() < 3e-4 * np.abs(ref).max()
    if mode == "multi-stage":
        xg, mg = dev(b["mlsa_x"]).requires_grad_(True), dev(g["mc_equal"]).requires_grad_(True)
        m = dsp.MLSA(24, 80, alpha=0.42, phase="mixed", mode="multi-stage", taylor_order=6, cep_order=50, dtype=torch.float64, device=DEV)
        y = m(xg, mg)
        assert np.abs(host(y) - g["multi_equal"]).max() < 1e-7 * np.abs(g["multi_equal"]).max()
        y.square().sum().backward()
        for got, name in ((xg.grad, "gx"), (mg.grad, "gmc")):
            assert np.abs(host(got) - g[name]).max() < 1e-6 * np.abs(g[name]).max(), name
    with pytest.raises(ValueError):
        dsp.MLSA((12, 24), 80, phase="mixed", mode=mode, device=DEV, **params)(x.float(), dev(g["mc_c0"], torch.float32)[..., :-1])


def test_gc2gc_wraps_large_phases(golden):
    """mgc2mgc.py:349-355: the phase is wrapped to (-pi, pi] before it is scaled by out_gamma; reference outputs for cepstra
    whose unwrapped phase reaches 4 pi (tests/golden/gc2gc_phase.npz), and the gradient against float64 autograd of the
    reference's formulation written with stock operators."""
    from diffsptk_amd.modules.mgc2mgc import gc2gc

    g = golden("gc2gc_phase")
    c = dev(g["c"])
    for ig, og, oo, nf in g["cases"]:
        key = f"gc2gc_{ig}_{og}_{int(oo)}_{int(nf)}"
        y = gc2gc(c, int(oo), float(ig), float(og), int(nf))
        close(host(y), g[key], rtol=1e-9, atol=1e-11)

    def ref(c1, oo, ig, og, nf):   # the reference's op sequence (mgc2mgc.py:333-361)
        c01 = torch.nn.functional.pad(c1[..., 1:], (1, 0))
        C1 = torch.fft.fft(c01, n=nf)
        if ig == 0:
            s = torch.polar(torch.exp(C1.real), C1.imag)
        else:
            z = 1 + ig * C1
            s = torch.polar(z.abs() ** (1 / ig), z.angle() / ig)
        C2 = torch.log(s.abs()) if og == 0 else ((s.abs() ** og) * torch.cos(s.angle() * og) - 1) / og
        c02 = torch.fft.ifft(C2).real[..., : oo + 1]
        return torch.cat((c1[..., :1], 2 * c02[..., 1:]), -1)

    ig, og, oo, nf = -0.25, -0.4, 12, 64
    w = torch.randn(4, oo + 1, dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    c1 = c.clone().requires_grad_(True)
    (gk,) = torch.autograd.grad((gc2gc(c1, oo, ig, og, nf) * w).sum(), c1)
    c2 = c.clone().requires_grad_(True)
    (gr,) = torch.autograd.grad((ref(c2, oo, ig, og, nf) * w).sum(), c2)
    close(host(gk), host(gr), rtol=1e-7, atol=1e-9 * float(gr.abs().max()))


def test_zerodf_broadcasts_leading_dims_and_rejects_mismatches():
    """The reference's direct mode takes x:(B, T) with an unbatched b:(N, M+1) (tensor broadcasting, zerodf.py:207-243); the
    kernels index b by (utterance, frame), so the host layer expands first -- and a shape that does not broadcast raises
    instead of reading out of bounds."""
    P, M, N, B = 10, 3, 6, 3
    gen = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(B, N * P, dtype=torch.float64, device=DEV, generator=gen)
    b = torch.randn(N, M + 1, dtype=torch.float64, device=DEV, generator=gen)
    y = F.zerodf(x, b, P)
    y_each = torch.stack([F.zerodf(x[i], b, P) for i in range(B)])
    close(host(y), host(y_each), rtol=0, atol=0)
    close(host(y), O.zerodf(host(x), np.broadcast_to(host(b), (B, N, M + 1)), P), rtol=1e-10, atol=1e-12)
    # gradient of the broadcast operand = the sum over the batch
    bg = b.clone().requires_grad_(True)
    (gb,) = torch.autograd.grad(F.zerodf(x, bg, P).square().sum(), bg)
    be = b.expand(B, N, M + 1).contiguous().requires_grad_(True)
    (gbe,) = torch.autograd.grad(F.zerodf(x, be, P).square().sum(), be)
    close(host(gb), host(gbe.sum(0)), rtol=1e-12, atol=1e-12)
    assert gb.shape == b.shape
    # one signal, a batch of filters
    y2 = F.zerodf(x[0], b.expand(2, N, M + 1) * torch.tensor([1.0, 2.0], dtype=torch.float64, device=DEV)[:, None, None], P)
    assert y2.shape == (2, N * P)
    with pytest.raises(ValueError):
        F.zerodf(x, torch.randn(2, N, M + 1, dtype=torch.float64, device=DEV), P)       # 3 signals, 2 filter sets
    with pytest.raises(ValueError):
        ops.ZerodfFn.apply(x, b, P, 0, False)                                             # the raw op refuses the mismatch
    with pytest.raises(ValueError):
        ops.ThSolveFn.apply(torch.rand(4, 5, dtype=torch.float64, device=DEV) + 2, torch.rand(5 * 2 - 1, dtype=torch.float64, device=DEV),
                            torch.rand(4, 5, dtype=torch.float64, device=DEV))


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-10), (torch.float32, 2e-5)])
def test_gc2gc_fused_kernel_equals_the_operator_chain(golden, dt, tol):
    """dsa_gc2gc_fwd (one launch, the n_fft-point spectrum in LDS) against the differentiable composition it replaces when no
    gradient is wanted (row transform -> element-wise operators -> inverse transform), the numpy oracle and the reference's
    outputs; every gamma case incl. wrapped phases, rows longer and shorter than the output, n_fft up to 4096."""
    from diffsptk_amd.modules.mgc2mgc import gc2gc

    g = golden("gc2gc_phase")
    c = dev(g["c"], dt)
    for ig, og, oo, nf in g["cases"]:
        y = gc2gc(c, int(oo), float(ig), float(og), int(nf))
        assert _lib.last_kernel() == "gc2gc_fused" and not y.requires_grad
        ref = g[f"gc2gc_{ig}_{og}_{int(oo)}_{int(nf)}"]
        assert np.abs(host(y) - ref).max() <= (1e-9 if dt == torch.float64 else 5e-5) * max(1.0, np.abs(ref).max())
    gen = torch.Generator().manual_seed(21)
    for n_in, oo, ig, og, nf in ((25, 1999, 0.0, 1.0, 4096), (25, 24, -0.5, 0.0, 512), (40, 12, -1.0, -0.25, 256), (9, 30, 0.0, -1 / 3, 64),
                                 (300, 10, -0.5, -0.5, 256)):
        cr = (torch.randn(6, n_in, dtype=torch.float64, generator=gen) * 0.3 / (1 + torch.arange(n_in, dtype=torch.float64) * 0.3)).to(DEV, dt)
        fused = gc2gc(cr, oo, ig, og, nf)
        assert _lib.last_kernel() == "gc2gc_fused"
        chain = gc2gc(cr.clone().requires_grad_(True), oo, ig, og, nf)      # the differentiable operator chain
        assert chain.requires_grad and chain.shape == fused.shape == (6, oo + 1)
        scale = max(1.0, float(chain.abs().max()))
        assert float((fused - chain.detach()).abs().max()) <= tol * scale, (n_in, oo, ig, og, nf)
        ref = O.gc2gc(host(cr).astype(np.float64), oo, ig, og, nf)
        assert np.abs(host(fused) - ref).max() <= (1e-9 if dt == torch.float64 else 5e-5) * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("g1,g2,n_fft,m1,m2", [(0.0, -0.5, 128, 24, 30), (-0.5, 0.0, 64, 8, 12), (-1.0, -1.0 / 3, 512, 24, 24),
                                              (-0.25, 1.0, 4096, 24, 1999), (0.0, 1.0, 256, 30, 100), (0.0, 0.0, 64, 10, 20)])
def test_gc2gc_backward_kernel_against_autograd_of_the_definition(g1, g2, n_fft, m1, m2):
    """dsa_gc2gc_bwd (ops.Gc2gcFn: one launch forward, one backward) against autograd through the definition written with
    torch.fft (mgc2mgc.py:333-361), float64 to 1e-10 and float32 to 2e-4 of the largest entry; every gamma branch, the
    4096-point transform of the single-stage MLSA filter's impulse responses included."""
    import math
    gen = torch.Generator().manual_seed(n_fft + m1)
    c1 = 0.3 * torch.randn(37, m1 + 1, generator=gen, dtype=torch.float64)
    gy = torch.randn(37, m2 + 1, generator=gen, dtype=torch.float64)

    def definition(c):
        c01 = torch.cat((torch.zeros_like(c[..., :1]), c[..., 1:]), -1)
        C1 = torch.fft.fft(c01, n=n_fft)
        if g1 == 0:
            mag, ang = torch.exp(C1.real), C1.imag
        else:
            z = 1 + g1 * C1
            mag, ang = z.abs() ** (1 / g1), z.angle() / g1
        if g2 == 0:
            C2 = torch.log(mag)
        else:
            ang = torch.remainder(ang + math.pi, 2 * math.pi) - math.pi
            C2 = (mag ** g2 * torch.cos(ang * g2) - 1) / g2
        c02 = torch.fft.ifft(C2).real[..., : m2 + 1]
        return torch.cat((c[..., :1], 2 * c02[..., 1:]), -1)

    cr = c1.to(DEV).requires_grad_(True)
    yr = definition(cr)
    (yr * gy.to(DEV)).sum().backward()
    from diffsptk_amd.modules.spec import device_twiddle
    for dt, tol in ((torch.float64, 1e-10), (torch.float32, 2e-4)):
        ck = c1.to(DEV, dt).requires_grad_(True)
        y = ops.gc2gc_fn(ck, m2, g1, g2, n_fft, device_twiddle(n_fft, ck.device, dt))
        assert y is not None and _lib.last_kernel() == "gc2gc_fused"
        assert float((y.double() - yr).abs().max()) < tol * float(yr.abs().max())
        (y * gy.to(DEV, dt)).sum().backward()
        err = float((ck.grad.double() - cr.grad).abs().max() / cr.grad.abs().max())
        assert err < tol, (g1, g2, n_fft, dt, err)


def test_thsolve_order24_float32_falls_back_to_pivoting_on_indefinite_systems():
    """The order-24 float32 path eliminates WITHOUT pivoting (sound for the analysis' positive definite systems).  A system whose
    elimination meets a non-positive pivot is marked and re-solved with row pivoting by the second launch -- what the reference's
    torch.linalg.solve (LAPACK, pivoted) does for every system: indefinite and zero-leading-pivot systems scattered among positive
    definite ones, each against numpy."""
    rng = np.random.default_rng(5)
    F, n = 2048 + 5, 24
    ii = np.arange(n)
    w = np.exp(rng.standard_normal((F, 48)))
    om = np.pi * (np.arange(48) + 0.5) / 48
    p = (w[:, None, :] * np.cos(om[None, None, :] * ii[None, :, None])).sum(-1)
    q = 0.5 * (w[:, None, :] * np.cos(om[None, None, :] * np.arange(2 * n - 1)[None, :, None])).sum(-1)
    r = rng.standard_normal((F, n))
    bad = rng.choice(F, 97, replace=False)
    p[bad[:40]] = rng.standard_normal((40, n)) * 3.0          # symmetric, indefinite
    q[bad[:40]] = rng.standard_normal((40, 2 * n - 1))
    p[bad[40:70], 0] = -q[bad[40:70], 0]                       # leading pivot p0 + q0 exactly zero: needs a row exchange
    p[bad[70:]] *= -1.0                                        # negative definite Toeplitz part
    A = p[:, np.abs(ii[:, None] - ii[None, :])] + q[:, ii[:, None] + ii[None, :]]
    ref = np.linalg.solve(A, r[..., None])[..., 0]
    g = host(ops.ThSolveFn.apply(dev(p, torch.float32), dev(q, torch.float32), dev(r, torch.float32)))
    assert _lib.last_kernel() == "th_solve_quad_fwd" and np.isfinite(g).all()
    cond = np.linalg.cond(A)
    err = np.abs(g - ref).max(-1) / np.abs(ref).max(-1)
    ok = cond < 1e4                                             # float32: the bound scales with the condition number
    assert ok[bad].sum() > 40 and err[ok].max() < 2e-2 and np.median(err[bad][ok[bad]]) < 1e-3, (err[ok].max(), np.median(err[bad]))
    good = np.setdiff1d(np.arange(F), bad)
    assert np.median(err[good]) < 2e-5
    # the Newton update in one call (dsa_thsolve_update_fwd): right-hand side read in place from the step's (F, 25) vector, the
    # solution added to b -- bit-identical to solve + addition, on the marked systems too
    r25 = torch.cat((torch.full((F, 1), 7.0), torch.from_numpy(r).float()), -1).to(DEV).contiguous()
    b_in = torch.randn(F, n, generator=torch.Generator().manual_seed(3)).to(DEV)
    out = ops.thsolve_update(dev(p, torch.float32), dev(q, torch.float32), r25, b_in)
    assert out is not None and torch.equal(out, b_in + torch.from_numpy(g).to(DEV))
    assert ops.thsolve_update(dev(p[:, :8], torch.float32), dev(q[:, :15], torch.float32), r25[:, :9].contiguous(), b_in[:, :8].contiguous()) is None


def test_f_rows_at_the_bench_size_against_the_oracle():
    """SURVEY 8(f) rows 3-4 at the size bench.py times them (256 utterances x 1 s = 51 200 frames, float32, default options):
    frames / utterances sampled across the batch against the float64 oracle.  Every row is independent per frame (analysis,
    conversions) or per utterance (the filters), so a sample checks the whole launch geometry: first / last workgroups, the
    ragged last chunk of frames, every utterance offset.  Tolerances: 3 x the error measured on this input
    (tools/measure_tolerances.py prints them)."""
    B, N, P, M, L, alpha = 256, 200, 80, 24, 512, 0.42
    g = torch.Generator().manual_seed(11)
    # coloured noise: white noise through a fixed one-pole smoother, so that the spectra have a 40 dB tilt like speech
    x = torch.randn(B, N * P, generator=g, dtype=torch.float64)
    x = torch.from_numpy(np.ascontiguousarray(scipy.signal.lfilter([1.0], [1.0, -0.9], x.numpy(), axis=-1)))
    xd = x.float().to(DEV)
    with torch.no_grad():
        X = dsp.STFT(400, P, L, device=DEV)(xd)[:, :N].contiguous()                      # (B, N, 257) power spectra
        frames = torch.tensor([0, 1, 199, 200, 12345, 25599, 25600, 40000, 51198, 51199])
        Xs = host(X.reshape(-1, L // 2 + 1)[frames]).astype(np.float64)
        # f3: mel-generalized cepstral analysis, gamma = -0.5, 10 Newton steps
        mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=L, cep_order=M, alpha=alpha, gamma=-0.5, n_iter=10, device=DEV)
        y = host(mg(X).reshape(-1, M + 1)[frames]).astype(np.float64)
        ref = O.mgcep(Xs, M, alpha=alpha, gamma=-0.5, n_iter=10)
        err_mgcep = np.abs(y - ref).max() / np.abs(ref).max()
        # the mel-cepstra the synthesis rows start from
        mc = dsp.MelCepstralAnalysis(fft_length=L, cep_order=M, alpha=alpha, n_iter=10, device=DEV)(X)
        mcs = host(mc.reshape(-1, M + 1)[frames]).astype(np.float64)
        # f4: mgc2sp, mc2b
        sp = host(dsp.MelGeneralizedCepstrumToSpectrum(M, L, alpha=alpha, device=DEV)(mc).reshape(-1, L // 2 + 1)[frames]).astype(np.float64)
        ref = O.mgc2sp(mcs, L, alpha=alpha)
        err_sp = (np.abs(sp - ref) / np.abs(ref)).max()
        bb = host(dsp.MelCepstrumToMLSADigitalFilterCoefficients(M, alpha, device=DEV)(mc).reshape(-1, M + 1)[frames]).astype(np.float64)
        ref = O.mc2b(mcs, alpha)
        err_b = np.abs(bb - ref).max() / np.abs(ref).max()
        # f4: the MLSA filter, three modes: excitation through the filters of the analysed cepstra; utterances 0, 131, 255
        exc = torch.randn(B, N * P, generator=g).to(DEV)
        utt = [0, 131, 255]
        errs = {}
        for mode, kw in (("multi-stage", {}), ("single-stage", {}), ("freq-domain", dict(frame_length=400, fft_length=L))):
            out = dsp.MLSA(M, P, alpha=alpha, mode=mode, device=DEV, **kw)(exc, mc)
            assert torch.isfinite(out).all()
            refm = O.mlsa(host(exc[utt]).astype(np.float64), host(mc[utt]).astype(np.float64), P, alpha=alpha, mode=mode, **kw)
            errs[mode] = np.abs(host(out[utt]).astype(np.float64) - refm).max() / np.abs(refm).max()
    print(f"bench-size f rows, float32 vs float64 oracle: mgcep {err_mgcep:.2e} mgc2sp {err_sp:.2e} mc2b {err_b:.2e} MLSA {errs}")
    # measured: mgcep 1.6e-7, mgc2sp 1.2e-6 (relative, per bin), mc2b 1.3e-7, MLSA 4.0e-7 / 3.6e-7 / 2.8e-7 of the largest sample
    assert err_mgcep < 5e-7
    assert err_sp < 4e-6
    assert err_b < 4e-7
    assert all(e < 1.5e-6 for e in errs.values()), errs


@pytest.mark.parametrize("n,F", [(2, 64), (5, 1000), (23, 77), (25, 300), (30, 1025), (35, 64), (36, 65), (43, 130), (44, 1000), (50, 1000),
                                 (51, 67), (52, 200), (55, 257)])
def test_thsolve_quad_layout_solver_for_general_orders(n, F):
    """The Toeplitz-plus-Hankel solve for orders other than 24 (csrc/thsolve_quad.hip: 16 systems per wave -- from order 36 on 8, the
    block columns dealt to the two quads of an octet --, 4 x 4 x 1 matrix products, no pivoting, marked systems re-solved with pivoting) against float64 LAPACK on positive definite systems (the
    Hessians of the analysis: T + H = 2 sum_w e(w) c(w) c(w)^T), ragged batch sizes; then systems that are NOT positive definite
    (the fallback must give the pivoted answer); and the fused Newton update of the untuned mel-cepstral step."""
    g = torch.Generator().manual_seed(n * 1000 + F)
    K = 4 * n
    w = torch.arange(K, dtype=torch.float64) * (math.pi / K)
    cw = torch.cos(torch.arange(2 * n - 1, dtype=torch.float64)[:, None] * w[None, :])        # (2n-1, K)
    e = torch.rand(F, K, generator=g, dtype=torch.float64) + 0.05
    rtl = (e @ cw.t()) / K                                                                     # rt[m] = mean_w e(w) cos(m w)
    p64, q64 = rtl[:, :n].contiguous(), rtl
    r64 = torch.randn(F, n, generator=g, dtype=torch.float64)

    def dense(p, q):
        i = torch.arange(n)
        return p[:, (i[:, None] - i[None, :]).abs()] + q[:, i[:, None] + i[None, :]]

    ref = torch.linalg.solve(dense(p64, q64), r64)
    pd, qd, rd = p64.float().to(DEV), q64.float().to(DEV), r64.float().to(DEV)
    got = ops.ThSolveFn.apply(pd, qd, rd)
    # order 24: the tuned kernel; up to 35: 16 systems per wave (a quad each); from 36: 8 systems per wave (an octet each)
    want_kernel = "th_solve_quad_fwd" if n == 24 else ("th_solve_quadn_fwd" if n <= 35 else "th_solve_octn_fwd")
    assert _lib.last_kernel() == want_kernel, _lib.last_kernel()
    ref32 = torch.linalg.solve(dense(pd.double().cpu(), qd.double().cpu()), rd.double().cpu())   # the float32 inputs' own solution
    cond = torch.linalg.cond(dense(p64, q64)).max()
    err = float((got.double().cpu() - ref32).abs().max() / ref32.abs().max())
    assert err < 2e-6 * float(cond) ** 0.5 + 2e-5, (err, float(cond))
    # not positive definite: random symmetric Toeplitz + Hankel
    F2 = min(F, 130)
    p2 = torch.randn(F2, n, generator=g, dtype=torch.float64)
    q2 = torch.randn(F2, 2 * n - 1, generator=g, dtype=torch.float64)
    r2 = torch.randn(F2, n, generator=g, dtype=torch.float64)
    A2 = dense(p2.float().double(), q2.float().double())
    ref2 = torch.linalg.solve(A2, r2.float().double())
    got2 = ops.ThSolveFn.apply(p2.float().to(DEV), q2.float().to(DEV), r2.float().to(DEV)).double().cpu()
    assert bool(torch.isfinite(got2).all())
    c2 = torch.linalg.cond(A2)
    ok = c2 < 1e3
    assert int(ok.sum()) > 0
    assert float(((got2 - ref2).abs().amax(-1) / ref2.abs().amax(-1))[ok].max()) < 1e-3
    # the fused Newton update: mc + solve(T(rt[:n]) + H(rt), rt[:n] - av)
    if 2 <= n <= 55:
        av = torch.randn(n, generator=g, dtype=torch.float64) * 0.1
        mc = torch.randn(F, n, generator=g, dtype=torch.float64)
        upd = ops.mcep_newton_update(qd, av.float().to(DEV), mc.float().to(DEV))
        ref_u = mc.float().double() + torch.linalg.solve(dense(pd.double().cpu(), qd.double().cpu()), pd.double().cpu() - av.float().double())
        assert float((upd.double().cpu() - ref_u).abs().max() / ref_u.abs().max()) < 2e-6 * float(cond) ** 0.5 + 2e-5
    # the backward: gr = A^-1 gbar on the same batched solve (A is symmetric), gp / gq = the diagonal / anti-diagonal sums of
    # -gr g^T (dsa_thsolve_bwd), against float64 autograd through the dense solve; 5e-5 of the largest entry (condition numbers
    # of a few hundred, float32)
    if n != 24:
        pg, qg, rg = (t.clone().requires_grad_(True) for t in (pd, qd, rd))
        w = torch.randn(F, n, generator=g, dtype=torch.float64)
        (ops.ThSolveFn.apply(pg, qg, rg) * w.float().to(DEV)).sum().backward()
        p6, q6, r6 = (t.detach().double().cpu().requires_grad_(True) for t in (pd, qd, rd))
        (torch.linalg.solve(dense(p6, q6), r6) * w.float().double()).sum().backward()
        for got_g, ref_g, name in ((pg.grad, p6.grad, "p"), (qg.grad, q6.grad, "q"), (rg.grad, r6.grad, "r")):
            e = float((got_g.double().cpu() - ref_g).abs().max() / ref_g.abs().max())
            assert e < 5e-6 * float(cond) ** 0.5 + 5e-5, (name, e, float(cond))


@pytest.mark.parametrize("gamma", [0.0, -0.5, -1.0, 0.5, 1.0])
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_gain_normalisation_kernels_against_the_stock_composition(gamma, dt):
    """dsa_gnorm_fwd (one launch, no gradient) against the stock tensor operations the modules keep for the differentiable case
    (gnorm.py:102-112, ignorm.py:99-109), both directions, ragged leading shapes; float32 2e-6, float64 1e-13 relative."""
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(3, 77, 9, generator=g, dtype=torch.float64) * 0.4 + 0.1).to(dt).to(DEV)    # 1 + gamma x0 stays positive
    tol = 2e-6 if dt == torch.float32 else 1e-13
    y = F.gnorm(x, gamma)
    assert _lib.last_kernel() == "gnorm_fwd"
    xr = x.clone().requires_grad_(True)
    y_ref = F.gnorm(xr, gamma)                      # a gradient is wanted: the composition
    assert float(((y - y_ref.detach()) / y_ref.detach()).abs().max()) < tol
    z = F.ignorm(y, gamma)
    assert _lib.last_kernel() == "ignorm_fwd"
    z_ref = F.ignorm(y.clone().requires_grad_(True), gamma)
    assert float(((z - z_ref.detach()) / z_ref.detach().abs().clamp_min(1e-3)).abs().max()) < tol * 10
    assert float((z - x).abs().max()) < (2e-6 if dt == torch.float32 else 1e-12)      # the round trip
