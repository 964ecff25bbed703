"""The 48 kHz mel-cepstral analysis WITH a gradient (round 6): McepNewtonStepsHFn -- per Newton step dsa_mcep_newton_update_bwd and
dsa_mcep_newton_resid_h_bwd (csrc/mcep_resid_bwd_f16.h) -- against float64 autograd of the ATen port of mcep.py:189-224
(oracle/torch_port.py), against the composed differentiable pieces it replaces, and the raw entry against float64 autograd of its own
formula.  Tolerances: gradients 2e-5 of the frame's largest entry (measured 0.7e-6 .. 6.2e-6; the composed path's own error is
3e-6 .. 3.4e-5)."""
import numpy as np
import pytest
import torch

import diffsptk_amd as dsp
from diffsptk_amd import _lib, ops
from oracle import torch_port as TP

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel_rows(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float(((a - b).abs().amax(-1) / b.abs().amax(-1).clamp_min(1e-300)).max())


def _grads(m, X, w):
    Xg = X.detach().clone().requires_grad_(True)
    y = m(Xg)
    (y * w).sum().backward()
    return y.detach(), Xg.grad


@pytest.mark.parametrize("nfft,M,alpha", [(2048, 49, 0.55), (1024, 34, 0.55), (2048, 40, 0.5), (2048, 54, 0.55), (2048, 32, 0.55), (2048, 48, 0.46)])
def test_48khz_gradient_one_node_vs_float64_autograd_and_the_composed_pieces(nfft, M, alpha, monkeypatch):
    K = nfft // 2 + 1
    g = torch.Generator().manual_seed(1000 + M)
    for F, n_iter in ((70, 2), (333, 10)):
        X = (torch.randn(F, K, generator=g).square() + 0.05).to(DEV)
        w = torch.randn(F, M + 1, generator=g).to(DEV)
        m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=alpha, n_iter=n_iter, device=DEV)
        monkeypatch.setenv("DSA_MCEP_GRAD_H", "1")
        assert ops.mcep_newton_steps_grad_applies(M + 1, m.D, m.E, m.alpha_vector)
        y1, g1 = _grads(m, X, w)
        monkeypatch.setenv("DSA_MCEP_GRAD_H", "0")
        y0, g0 = _grads(m, X, w)
        tab = TP.McepTables(nfft, M, alpha, torch.float64)
        Xs = X.double().cpu().requires_grad_(True)
        yr = TP.mcep(Xs, tab, n_iter)
        (yr * w.double().cpu()).sum().backward()
        assert torch.isfinite(g1).all()
        np.testing.assert_allclose(y1.cpu().numpy(), yr.detach().numpy(), rtol=1e-4, atol=5e-6)
        np.testing.assert_allclose(y1.cpu().numpy(), y0.cpu().numpy(), rtol=1e-4, atol=5e-6)   # (the composed forward runs float32 matrix products)
        assert _rel_rows(g1, Xs.grad) < 2e-5, (F, n_iter, _rel_rows(g1, Xs.grad))
        assert _rel_rows(g1, g0) < 1e-4
        # with a gradient wanted or not: the same values (the no-gradient path is the one persistent launch)
        with torch.no_grad():
            assert torch.equal(m(X), y1)


@pytest.mark.parametrize("K,n", [(1025, 50), (513, 35), (1025, 33), (1025, 55), (1024, 48), (129, 40), (501, 41), (37, 33)])
def test_resid_h_bwd_entry_vs_float64_autograd_of_its_formula(K, n):
    """dsa_mcep_newton_resid_h_bwd alone: glogx += (grt E^T) * e, gmc = -2 ((grt E^T) * e) D^T with e = exp(logx - 2 mc D), ragged batch,
    accumulation into a non-zero glogx, rows past the batch untouched."""
    g = torch.Generator().manual_seed(K + n)
    F = 150
    D = (torch.randn(n, K, generator=g) * 0.3).to(DEV)
    E = (torch.randn(K, 2 * n - 1, generator=g) * 0.01).to(DEV)
    logx = torch.randn(F, K, generator=g).to(DEV)
    mc = (torch.randn(F, n, generator=g) * 0.2).to(DEV)
    grt = torch.randn(F, 2 * n - 1, generator=g).to(DEV)
    grt[7] *= 1e6          # per-frame scales
    grt[8] *= 1e-6
    grt[9] = 0.0
    glogx0 = torch.randn(F + 3, K, generator=g).to(DEV)
    glogx0[:F] = 0.0       # (the sums start at zero, as the reverse sweep starts them; the rows past the batch keep their noise)
    images = ops.mcep_resid_bwd_images(D, E)
    assert images is not None
    glogx = glogx0.clone()
    gmc = torch.full((F + 3, n), 7.0, device=DEV)
    ops._call("dsa_mcep_newton_resid_h_bwd", ops._p(logx), F, K, ops._p(mc), n, ops._p(grt), ops._p(images), ops._dtype_code(logx), ops._p(glogx),
              ops._p(gmc), ops._stream())
    assert _lib.last_kernel() == "mcep_resid_bwd_h"
    lx, mcd = logx.double().requires_grad_(True), mc.double().requires_grad_(True)
    rt = torch.exp(lx - 2.0 * mcd @ D.double()) @ E.double()
    (rt * grt.double()).sum().backward()
    assert torch.equal(glogx[F:], glogx0[F:]) and bool((gmc[F:] == 7.0).all())      # rows past the batch: untouched
    assert _rel_rows(glogx[:F], lx.grad) < 5e-6, _rel_rows(glogx[:F], lx.grad)
    assert _rel_rows(gmc[:F], mcd.grad) < 5e-6, _rel_rows(gmc[:F], mcd.grad)
    # a second call ADDS into glogx (read-modify-write) and overwrites gmc
    gmc_b = torch.empty(F, n, device=DEV)
    glogx_b = glogx.clone()
    ops._call("dsa_mcep_newton_resid_h_bwd", ops._p(logx), F, K, ops._p(mc), n, ops._p(grt), ops._p(images), ops._dtype_code(logx), ops._p(glogx_b),
              ops._p(gmc_b), ops._stream())
    assert torch.equal(glogx_b[:F], glogx[:F] + glogx[:F]) and torch.equal(gmc_b, gmc[:F]) and torch.equal(glogx_b[F:], glogx0[F:])
    # every frame's arithmetic depends on the frame alone: a slice gives the same bits
    gl2 = torch.zeros(40, K, device=DEV)
    gm2 = torch.empty(40, n, device=DEV)
    ops._call("dsa_mcep_newton_resid_h_bwd", ops._p(logx[60:100].contiguous()), 40, K, ops._p(mc[60:100].contiguous()), n, ops._p(grt[60:100].contiguous()),
              ops._p(images), ops._dtype_code(logx), ops._p(gl2), ops._p(gm2), ops._stream())
    assert torch.equal(gm2, gmc[60:100]) and torch.equal(gl2, glogx[60:100])
    # a non-finite frame stays in its rows
    lx_bad = logx.clone()
    lx_bad[5, 3] = float("nan")
    gl3, gm3 = glogx0.clone(), torch.empty(F, n, device=DEV)
    ops._call("dsa_mcep_newton_resid_h_bwd", ops._p(lx_bad), F, K, ops._p(mc), n, ops._p(grt), ops._p(images), ops._dtype_code(logx), ops._p(gl3),
              ops._p(gm3), ops._stream())
    keep = torch.ones(F, dtype=torch.bool, device=DEV)
    keep[5] = False
    assert torch.equal(gm3[keep], gmc[:F][keep]) and torch.equal(gl3[:F][keep], glogx[:F][keep]) and not torch.isfinite(gm3[5]).all()


def test_48khz_gradient_any_bin_count_and_the_fallback_outside_the_orders(monkeypatch):
    """K = fft_length / 2 + 1 with a ragged last stage of several bins (fft_length 1000: K = 501 = 15 x 32 + 21) runs the one node;
    orders outside 32 .. 54 keep the composed gradient."""
    g = torch.Generator().manual_seed(5)
    m = dsp.MelCepstralAnalysis(fft_length=1000, cep_order=40, alpha=0.5, n_iter=3, device=DEV)
    assert ops.mcep_newton_steps_grad_applies(41, m.D, m.E, m.alpha_vector)
    X = (torch.randn(50, 501, generator=g).square() + 0.05).to(DEV)
    w = torch.randn(50, 41, generator=g).to(DEV)
    _, g1 = _grads(m, X, w)
    tab = TP.McepTables(1000, 40, 0.5, torch.float64)
    Xs = X.double().cpu().requires_grad_(True)
    (TP.mcep(Xs, tab, 3) * w.double().cpu()).sum().backward()
    assert _rel_rows(g1, Xs.grad) < 2e-5
    m2 = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=30, alpha=0.5, n_iter=2, device=DEV)
    assert not ops.mcep_newton_steps_grad_applies(31, m2.D, m2.E, m2.alpha_vector)


@pytest.mark.parametrize("nfft,M", [(2048, 49), (1024, 34)])
def test_48khz_gradient_is_batch_invariant(nfft, M):
    """A frame's value and gradient depend on the frame alone: the same bits alone, in a batch of 3, of 70 and of 20 000 frames (every kernel of the
    node is chosen by the order, none by the batch size; above 16 384 frames the NO-gradient path plans wide tiles -- not this path)."""
    K = nfft // 2 + 1
    g = torch.Generator().manual_seed(77 + M)
    X = (torch.randn(20000, K, generator=g).square() + 0.05).to(DEV)
    w = torch.randn(20000, M + 1, generator=g).to(DEV)
    m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=4, device=DEV)
    yb, gb = _grads(m, X, w)
    for lo, hi in ((0, 1), (5, 8), (100, 170), (19990, 20000)):
        ys, gs = _grads(m, X[lo:hi], w[lo:hi])
        assert torch.equal(ys, yb[lo:hi]) and torch.equal(gs, gb[lo:hi]), (lo, hi)


@pytest.mark.parametrize("nfft,M,n_iter", [(2048, 49, 10), (1024, 34, 10), (2048, 54, 12), (2048, 32, 3), (1000, 40, 1), (2048, 48, 15)])
def test_48khz_glogx_in_one_pass_equals_the_in_place_accumulation_bit_for_bit(nfft, M, n_iter, monkeypatch):
    """0.2.2: the spectrum's gradient formed AFTER the reverse sweep, in one pass over the bins from the saved iterates and the steps'
    cotangents (dsa_mcep_newton_glogx_h; the sweep's launches pass glogx = NULL), against the sweep's in-place accumulation
    (DSA_MCEP_GLOGX_PASS=0): the same values summed in the same order -- the same bits, at ragged batch sizes and bin counts; n_iter
    beyond what the pass holds on chip (13 at orders >= 48) keeps the accumulation."""
    K = nfft // 2 + 1
    g = torch.Generator().manual_seed(2000 + M + n_iter)
    m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.5, n_iter=n_iter, device=DEV)
    assert ops.mcep_newton_steps_grad_applies(M + 1, m.D, m.E, m.alpha_vector)
    for F in (1, 17, 333):
        X = (torch.randn(F, K, generator=g).square() + 0.05).to(DEV)
        w = torch.randn(F, M + 1, generator=g).to(DEV)
        monkeypatch.setenv("DSA_MCEP_GLOGX_PASS", "1")
        y1, g1 = _grads(m, X, w)
        k1 = _lib.last_kernel()
        monkeypatch.setenv("DSA_MCEP_GLOGX_PASS", "0")
        y0, g0 = _grads(m, X, w)
        assert torch.equal(y1, y0) and torch.isfinite(g1).all()
        assert torch.equal(g1, g0), (F, float((g1 - g0).abs().max()))
    fits = n_iter * ((4 + 2 * ((2 * (M + 1) - 1 + 31) // 32)) * 1024 + 128) <= 156 * 1024
    assert fits == (n_iter <= (12 if M + 1 >= 49 else 15))
    # a non-finite frame stays in its own row
    # by default the pass takes over from ops.MCEP_GLOGX_MIN_FRAMES frames on -- the same bits either side of the threshold
    monkeypatch.delenv("DSA_MCEP_GLOGX_PASS")
    monkeypatch.setattr(ops, "MCEP_GLOGX_MIN_FRAMES", 300)
    assert torch.equal(_grads(m, X, w)[1], g1)
    monkeypatch.setattr(ops, "MCEP_GLOGX_MIN_FRAMES", 400)
    assert torch.equal(_grads(m, X, w)[1], g1)
    Xb = X.clone()
    Xb[5, 3] = float("nan")
    monkeypatch.setenv("DSA_MCEP_GLOGX_PASS", "1")
    _, gb = _grads(m, Xb, w)
    keep = torch.ones(F, dtype=torch.bool, device=DEV)
    keep[5] = False
    assert torch.equal(gb[keep], g1[keep]) and not torch.isfinite(gb[5]).all()


@pytest.mark.parametrize("K,n,n_iter,F", [(1025, 50, 4, 37), (513, 35, 10, 100), (37, 33, 3, 5), (129, 40, 1, 16), (501, 41, 7, 33), (1024, 55, 12, 3)])
def test_glogx_entry_against_float64_of_its_formula_and_null_glogx_in_the_sweep(K, n, n_iter, F):
    """dsa_mcep_newton_glogx_h alone against float64 of  sum_s (grt_s E^T) * exp(logx - 2 mc_s D); dsa_mcep_newton_resid_h_bwd with
    glogx = NULL returns the gmc of the accumulating call, bit for bit; ragged bin counts (a last unit of 5, 1, 16 bins), one step."""
    g = torch.Generator().manual_seed(9 + K)
    m = dsp.MelCepstralAnalysis(fft_length=2 * (K - 1), cep_order=n - 1, alpha=0.55, n_iter=1, device=DEV)
    images = ops.mcep_resid_bwd_images(m.D, m.E)
    logx = (torch.randn(F, K, generator=g) * 0.7).to(DEV)
    mcs = (torch.randn(n_iter, F, n, generator=g) * 0.02).to(DEV)
    mcs[:, :, 0] += 0.3
    grts = torch.randn(n_iter, F, 2 * n - 1, generator=g).to(DEV)
    glogx = torch.empty(F, K, device=DEV)
    ops._call("dsa_mcep_newton_glogx_h", ops._p(logx), F, K, ops._p(mcs), n, ops._p(grts), n_iter, ops._p(images), ops._dtype_code(logx),
              ops._p(glogx), ops._stream())
    D, E = m.D.double(), m.E.double()
    ref = sum((grts[s].double() @ E.T) * torch.exp(logx.double() - 2 * mcs[s].double() @ D) for s in range(n_iter))
    assert _rel_rows(glogx, ref) < 2e-6, _rel_rows(glogx, ref)
    acc = torch.zeros(F, K, device=DEV)
    gm_a, gm_n = torch.empty(F, n, device=DEV), torch.empty(F, n, device=DEV)
    for s in range(n_iter - 1, -1, -1):
        ops._call("dsa_mcep_newton_resid_h_bwd", ops._p(logx), F, K, ops._p(mcs[s]), n, ops._p(grts[s]), ops._p(images), ops._dtype_code(logx),
                  ops._p(acc), ops._p(gm_a), ops._stream())
        ops._call("dsa_mcep_newton_resid_h_bwd", ops._p(logx), F, K, ops._p(mcs[s]), n, ops._p(grts[s]), ops._p(images), ops._dtype_code(logx),
                  None, ops._p(gm_n), ops._stream())
        assert torch.equal(gm_a, gm_n), s
    assert torch.equal(acc, glogx)
