"""fuse(frame, window, lpc): the LPC branch of the reference's README (README.md:198-201; BASELINE configs[3]) through the
reference's own modules, ONE launch forward and ONE launch backward (dsa_frame_window_lpc_fwd / _bwd, csrc/lpc.hip) -- against the
reference's exported outputs and gradients (tests/golden/randn.npz: lpc_f64, grad_lpc_wsum_f64), the module chain on the float64
kernels, float64 gradcheck-grade finite differences of the chain, and the size-independent properties of the partition."""
import numpy as np
import pytest
import torch

import diffsptk_amd as dsp
from diffsptk_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mods(L, P, M=24, dt=torch.float32, center=True, eps=1e-5, window="blackman"):
    return dsp.Frame(L, P, center=center), dsp.Window(L, window=window, dtype=dt, device=DEV), dsp.LPC(L, M, eps=eps, dtype=dt, device=DEV)


def _chain64(x, L, P, center=True, eps=1e-5, gy=None):
    """outputs and input gradient of the module chain on the float64 kernels"""
    f, w, l = _mods(L, P, dt=torch.float64, center=center, eps=eps)
    xd = x.double().clone().requires_grad_(True)
    y = l(w(f(xd)))
    (y * (gy.double() if gy is not None else 1.0)).sum().backward()
    return y.detach(), xd.grad


def test_fused_lpc_golden_forward_and_gradient(golden):
    g = golden("randn")
    x = torch.from_numpy(g["x"]).float().to(DEV).requires_grad_(True)
    f, w, l = _mods(400, 80)
    fl = dsp.fuse(f, w, l)
    y = fl(x)
    assert fl.last_path == "fused" and _lib.last_kernel() == "frame_window_lpc24_mfma_fwd"
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["lpc_f64"], rtol=1e-4, atol=1e-4)
    wts = torch.linspace(-1, 1, 25, device=DEV)
    (y * wts).sum().backward()
    ref = g["grad_lpc_wsum_f64"]
    err = np.abs(x.grad.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 1e-4, err          # (the float32 module chain of the reference is held to 1e-3 by test_lpc_backward_golden)
    with torch.no_grad():
        y2 = fl(x)
    assert fl.last_path == "fused-forward" and torch.equal(y2, y.detach())


@pytest.mark.parametrize("L,P,T,B,center", [(400, 80, 16000, 3, True), (400, 80, 1234, 2, True), (400, 160, 4001, 2, True), (400, 80, 977, 2, False),
                                            (25, 7, 300, 2, True), (512, 128, 3000, 2, True), (401, 100, 2500, 1, True), (127, 64, 1000, 3, False),
                                            (400, 80, 80, 2, True), (400, 80, 1, 1, True), (256, 300, 2000, 2, True), (400, 400, 4000, 1, False)])
def test_fused_lpc_backward_against_the_float64_chain(L, P, T, B, center):
    """every frame-length class, ragged lengths, utterances shorter than a frame, hops longer than the overlap (samples no frame
    covers: zero gradient), both centring modes: outputs and gradients against the module chain on the float64 kernels"""
    gen = torch.Generator().manual_seed(L * 131 + P * 7 + T)
    x = torch.randn(B, T, generator=gen).to(DEV)
    N = (T - 1) // P + 1
    gy = torch.randn(B, N, 25, generator=gen).to(DEV)
    f, w, l = _mods(L, P, center=center)
    fl = dsp.fuse(f, w, l)
    xg = x.clone().requires_grad_(True)
    y = fl(xg)
    assert fl.last_path == "fused", fl.last_path
    (y * gy).sum().backward()
    assert _lib.last_kernel() in ("frame_window_lpc24_bwd_mfma", "frame_window_lpc24_mfma_fwd")
    y64, g64 = _chain64(x, L, P, center=center, gy=gy)
    # (a frame of 25 samples carries an order-24 Toeplitz system that eps alone conditions: float32 lag sums move it by 1e-4)
    tol = 5e-4 if L < 64 else 2e-5
    assert float((y.detach().double() - y64).abs().max()) < tol * max(1.0, float(y64.abs().max()))
    scale = float(g64.abs().max())
    assert float((xg.grad.double() - g64).abs().max()) < tol * scale, (float((xg.grad.double() - g64).abs().max()), scale)
    assert bool(torch.isfinite(xg.grad).all())


def test_fused_lpc_gradient_is_partition_invariant_and_reproducible():
    """every sample's sum runs over its frames in increasing order whatever run of hops its wave owns: an utterance alone, inside
    a batch, or as the head of a longer one (up to the frames that reach past the cut) gives the same bits; launches repeat"""
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(5, 16000, generator=gen).to(DEV)
    gy = torch.randn(5, 200, 25, generator=gen).to(DEV)
    fl = dsp.fuse(*_mods(400, 80))

    def grad(xs, gs):
        xg = xs.clone().requires_grad_(True)
        (fl(xg) * gs).sum().backward()
        return xg.grad

    g_all = grad(x, gy)
    for _ in range(3):
        assert torch.equal(grad(x, gy), g_all)
    assert torch.equal(grad(x[2:3], gy[2:3]), g_all[2:3])
    # 6000 samples = 75 hops: another split of the utterance into runs; samples whose frames all lie below frame 75 - 3 agree
    g_head = grad(x[:, :6000].contiguous(), gy[:, :75].contiguous())
    assert torch.equal(g_head[:, :5500], g_all[:, :5500])


def test_fused_lpc_bench_size_properties():
    """config 4 at full size (1024 utterances x 1 s): finite, sampled utterances against the float64 chain, zero cotangent -> zero"""
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(1024, 16000, generator=gen).to(DEV)
    gy = torch.randn(1024, 200, 25, generator=gen).to(DEV)
    fl = dsp.fuse(*_mods(400, 80))
    xg = x.clone().requires_grad_(True)
    y = fl(xg)
    (y * gy).sum().backward()
    assert bool(torch.isfinite(xg.grad).all())
    idx = [0, 1, 511, 1023]
    y64, g64 = _chain64(x[idx], 400, 80, gy=gy[idx])
    # measured (tools/measure_tolerances.py): outputs 2.1e-7, gradient 4.3e-7 of its maximum; bounds at three times that
    assert float((y[idx].double() - y64).abs().max()) < 1e-6
    assert float((xg.grad[idx].double() - g64).abs().max()) < 1.5e-6 * float(g64.abs().max())
    (g0,) = torch.autograd.grad((fl(xg) * 0.0).sum(), xg)
    assert float(g0.abs().max()) == 0.0


def test_fused_lpc_fallbacks_and_errors():
    f, w, l = _mods(400, 80)
    with pytest.raises(ValueError):
        dsp.fuse(f, w)
    with pytest.raises(ValueError):
        dsp.fuse(f, dsp.Window(300, device=DEV), l)
    x = torch.randn(2, 3000, device=DEV)
    # float64: the three modules
    f64 = dsp.fuse(*_mods(400, 80, dt=torch.float64))
    xd = x.double().requires_grad_(True)
    y = f64(xd)
    assert f64.last_path == "three-stage"
    y.sum().backward()
    # reflect padding: one launch forward, the module chain with a gradient
    fr = dsp.fuse(dsp.Frame(400, 80, mode="reflect"), w, l)
    with torch.no_grad():
        y0 = fr(x)
    assert fr.last_path == "fused-forward"
    xg = x.clone().requires_grad_(True)
    y1 = fr(xg)
    assert fr.last_path == "three-stage"
    np.testing.assert_allclose(y1.detach().cpu().numpy(), y0.cpu().numpy(), rtol=2e-4, atol=2e-4)
    # a learnable window: the differentiable modules
    wl = dsp.Window(400, learnable=True, device=DEV)
    fw = dsp.fuse(f, wl, l)
    fw(x).sum().backward()
    assert fw.last_path == "three-stage" and wl.window.grad is not None
    # other orders: forward fused (generic fused kernel), backward through the chain
    fo = dsp.fuse(f, w, dsp.LPC(400, 12, eps=1e-5, device=DEV))
    xg = x.clone().requires_grad_(True)
    fo(xg).sum().backward()
    assert fo.last_path == "three-stage"


def test_exact_lag_sums_flag_on_a_near_singular_frame():
    """a sinusoid plus tiny noise with a small eps: the Toeplitz system amplifies the lag sums' error; exact float64 sums
    (DSA_LPC_EXACT_LAGSUMS through the API, not an environment variable) against the float64 chain"""
    gen = torch.Generator().manual_seed(3)
    t = torch.arange(4000, dtype=torch.float64)
    x = (torch.sin(0.3 * t) + 1e-3 * torch.randn(4000, generator=gen, dtype=torch.float64)).float().unsqueeze(0).to(DEV)
    f, w, l = _mods(400, 80, eps=1e-9)
    y64, _ = _chain64(x, 400, 80, eps=1e-9)
    with torch.no_grad():
        ya = dsp.fuse(f, w, l)(x)
        assert _lib.last_kernel() == "frame_window_lpc24_mfma_fwd"
        ye = dsp.fuse(f, w, l, exact_lag_sums=True)(x)
        assert _lib.last_kernel() == "frame_window_lpc24_fwd"
    ea = float((ya.double() - y64).abs().max())
    ee = float((ye.double() - y64).abs().max())
    print(f"near-singular frames: matrix-pipe lag sums {ea:.3e}, exact lag sums {ee:.3e} from the float64 chain")
    assert ee <= ea + 1e-6 and ee < 5e-3


def test_fused_lpc_nan_containment_and_empty_batch():
    """a non-finite sample poisons the frames that contain it and the gradient of the samples those frames cover -- nothing else
    (the per-frame power-of-two scales of the matrix-pipe products are taken with NaN-ignoring maxima); an empty batch is a no-op"""
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(3, 8000, generator=gen).to(DEV)
    gy = torch.randn(3, 100, 25, generator=gen).to(DEV)
    fl = dsp.fuse(*_mods(400, 80))

    def run(xs):
        xg = xs.clone().requires_grad_(True)
        y = fl(xg)
        (y * gy).sum().backward()
        return y.detach(), xg.grad

    y0, g0 = run(x)
    xb = x.clone()
    xb[1, 4000] = float("nan")
    y1, g1 = run(xb)
    bad_frames = torch.isnan(y1).any(-1)
    # frames n with 80 n - 200 <= 4000 < 80 n + 200: n = 48 .. 52 of utterance 1
    assert bad_frames[1, 48:53].all() and int(bad_frames.sum()) == 5
    ok = ~bad_frames
    assert torch.equal(y1[ok], y0[ok])
    bad_samples = torch.isnan(g1)
    assert not bad_samples[0].any() and not bad_samples[2].any()
    lo, hi = 48 * 80 - 200, 52 * 80 + 200
    assert not bad_samples[1, :lo].any() and not bad_samples[1, hi:].any() and bad_samples[1, lo:hi].all()
    assert torch.equal(g1[0], g0[0]) and torch.equal(g1[2], g0[2])
    assert torch.equal(g1[1, :lo], g0[1, :lo]) and torch.equal(g1[1, hi:], g0[1, hi:])
    xe = torch.zeros(0, 8000, device=DEV, requires_grad=True)
    ye = fl(xe)
    assert ye.shape == (0, 100, 25)
    ye.sum().backward()
    assert xe.grad.shape == (0, 8000)
