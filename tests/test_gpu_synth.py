"""GPU parity tests of SURVEY.md section 8(f) rows 3-4 (run with -m gpu on an MI355X): mel-generalized cepstral
analysis and the cepstrum conversions around it, against outputs of the reference (tests/golden/synth.npz) and the
numpy oracle.  Tolerances: float64 rtol 1e-5 / atol 1e-8 (tests/utils.py:66-72 of the reference); float32 as stated
per test."""
import numpy as np
import pytest
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
            assert _lib.last_kernel() == "th_solve_fwd"
            assert np.abs(host(gsol) - ref).max() < tol * max(1.0, np.abs(ref).max()), (n, dt)
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


def test_mgcep_speech_512_and_gamma0_route(golden):
    g = golden("synth")
    X = dev(g["mgcep512_X"])
    m = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, c=3, n_iter=5, dtype=torch.float64, device=DEV)
    close(host(m(X)), g["mgcep512_c3_5"], **F64)
    m32 = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, c=3, n_iter=5, device=DEV)
    y32 = host(m32(X.float())).astype(np.float64)
    assert np.abs(y32 - g["mgcep512_c3_5"]).max() < 5e-3              # float32 through pow(D, 3) on speech spectra (reference alike)
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
