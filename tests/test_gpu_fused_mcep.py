"""STFT -> mel-cepstrum in ONE launch (dsa_stft_mcep_fwd, `diffsptk_amd.fuse(stft, mcep)`; SURVEY.md 8(d): 420 bytes per
frame instead of 1348 + 1128) against the two-kernel path and the oracle.

The fused wave runs the packed STFT kernel's instructions on its tile's frames, so the power spectrogram it can leave behind
(X_out) must equal the stand-alone kernel's bit for bit; the mel-cepstra go through the same Newton code and are held to the
two-kernel result (bitwise in practice, asserted to 1e-6 of the largest coefficient) and to the float64 oracle at the
tolerance of tests/test_gpu_parity.py (rtol 1e-4, atol 5e-6)."""
import numpy as np
import pytest
import torch

import diffsptk_amd as dsp
from diffsptk_amd import _lib, ops
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
MC32 = dict(rtol=1e-4, atol=5e-6)


def host(t):
    return t.detach().cpu().numpy()


def _modules(P=80, center=True, n_iter=10, **kw):
    stft = dsp.STFT(400, P, 512, center=center, device=DEV, **kw)
    mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=n_iter, device=DEV)
    return stft, mcep, dsp.fuse(stft, mcep)


def _fused_with_spectrogram(stft, mcep, x, pad_mode=0):
    """the raw launch with both side products (what the autograd Function keeps for the backward)"""
    T = x.size(-1)
    B = x.numel() // T
    N = ops.num_frames(T, stft.frame_period)
    mc = torch.empty(B, N, 25, device=DEV)
    X = torch.full((B, N, 257), float("nan"), device=DEV)
    hist = torch.empty(mcep.n_iter + 1, B * N, 25, device=DEV)
    images = ops.mcep_images(mcep.G, mcep.D, mcep.E, 512, 24)
    scratch = torch.zeros(_lib.SCRATCH_BYTES, dtype=torch.uint8, device=DEV)
    ops._call("dsa_stft_mcep_fwd", x.data_ptr(), B, T, 400, stft.frame_period, 512, stft.window.data_ptr(), stft.twiddle.data_ptr(),
              int(stft.center), float(stft.eps), 24, mcep.n_iter, mcep.G.data_ptr(), mcep.D.data_ptr(), mcep.E.data_ptr(),
              mcep.alpha_vector.data_ptr(), _lib.F32, _lib.ALGO_AUTO | (pad_mode << 12), images.data_ptr(), scratch.data_ptr(), mc.data_ptr(),
              hist.data_ptr(), X.data_ptr(), ops._stream())
    assert _lib.last_kernel() == "stft512_mcep_fused_fwd"
    return mc, X, hist


@pytest.mark.parametrize("B,T,P,center", [(3, 16000, 80, True), (2, 15997, 80, True), (5, 1234, 80, True), (1, 1, 80, True),
                                          (2, 4000, 160, True), (3, 2000, 50, False), (1, 401, 7, True), (7, 3211, 80, True)])
def test_fused_equals_two_kernels_and_oracle(B, T, P, center):
    x = torch.randn(B, T, generator=torch.Generator().manual_seed(B * 7 + T)).to(DEV)
    stft, mcep, fused = _modules(P, center)
    with torch.no_grad():
        X2 = stft(x)
        mc2 = mcep(X2)
        mc1 = fused(x)
    assert fused.last_path == "fused" and _lib.last_kernel() == "stft512_mcep_fused_fwd"
    assert mc1.shape == mc2.shape and bool(torch.isfinite(mc1).all())
    assert float((mc1 - mc2).abs().max()) <= 1e-6 * float(mc2.abs().max())
    mcf, Xf, hist = _fused_with_spectrogram(stft, mcep, x)
    # the spectrogram side product: where the packed kernel serves the geometry, its own values bit for bit
    with torch.no_grad():
        stft(x)
    if _lib.last_kernel() == "stft512_fwd" and P % 2 == 0 and center and P in (80, 160):
        assert torch.equal(Xf, X2)
    else:
        assert float(((Xf - X2).abs() / X2.amax(-1, keepdim=True)).max()) < 2e-6
    assert torch.equal(mcf, mc1) and torch.equal(hist[-1].view_as(mc1), mc1)
    X_ref = O.stft(host(x).astype(np.float64), 400, P, 512, center=center)
    np.testing.assert_allclose(host(mc1), O.mcep(X_ref, 24, 0.42, 10), **MC32)


def test_fused_config5_shard_against_the_oracle_and_bitwise_properties():
    """BASELINE configs[4], one rank's shard (1024 utterances x 1 s) through the one-launch path."""
    B = 1024
    x = torch.randn(B, 16000, generator=torch.Generator().manual_seed(11))
    xd = x.to(DEV)
    stft, mcep, fused = _modules()
    with torch.no_grad():
        mc = fused(xd)
        assert fused.last_path == "fused"
        two = mcep(stft(xd))
    assert mc.shape == (B, 200, 25) and bool(torch.isfinite(mc).all())
    assert float((mc - two).abs().max()) <= 1e-6 * float(two.abs().max())
    sel = [0, 171, 342, 513, 684, 855, 1023]
    X_ref = O.stft(x[sel].double().numpy(), 400, 80, 512)
    np.testing.assert_allclose(host(mc)[sel], O.mcep(X_ref, 24, 0.42, 10), **MC32)
    idx = torch.randperm(B, generator=torch.Generator().manual_seed(12)).to(DEV)
    with torch.no_grad():
        assert torch.equal(fused(xd[idx]), mc[idx])       # utterances are independent: bitwise
        assert torch.equal(fused(xd), mc)                 # and the launch is reproducible (dynamic tile queue or not)
        assert torch.equal(fused(xd[:100]), mc[:100])     # independent of the batch it sits in


def test_fused_gradient_equals_the_two_stage_gradient():
    x = torch.randn(6, 8000, generator=torch.Generator().manual_seed(5)).to(DEV)
    stft, mcep, fused = _modules()
    w = torch.randn(6, 100, 25, generator=torch.Generator().manual_seed(6)).to(DEV)
    xa = x.clone().requires_grad_(True)
    ya = fused(xa)
    assert fused.last_path == "fused"
    (ya * w).sum().backward()
    xb = x.clone().requires_grad_(True)
    yb = mcep(stft(xb))
    (yb * w).sum().backward()
    assert float((ya - yb).abs().max()) <= 1e-6 * float(yb.abs().max())
    assert float((xa.grad - xb.grad).abs().max()) <= 1e-5 * float(xb.grad.abs().max())


@pytest.mark.parametrize("mode", ["reflect", "replicate", "circular"])
@pytest.mark.parametrize("B,T,P,center", [(3, 16000, 80, True), (2, 15997, 80, True), (5, 1234, 80, True), (2, 4000, 160, True),
                                          (3, 2000, 50, False), (4, 401, 80, True), (2, 640, 80, False)])
def test_fused_pad_modes_equal_the_two_kernels_and_the_oracle(mode, B, T, P, center):
    """ShortTimeFourierTransform(mode=...) (stft.py:86-104, frame.py:130-137) through the ONE launch (round 6; DSA_ALGO_PAD_MODE):
    `last_path == "fused"`, the spectrogram side product bit-identical to the two-kernel path's -- which runs the packed kernel
    for every pad mode now --, the mel-cepstra at 1e-6 of the two-kernel result and at the goldens' tolerance against the float64
    oracle, and the gradient equal to the two-stage gradient."""
    x = torch.randn(B, T, generator=torch.Generator().manual_seed(B * 11 + T)).to(DEV)
    stft, mcep, fused = _modules(P, center, mode=mode)
    with torch.no_grad():
        X2 = stft(x)
        k2 = _lib.last_kernel()
        mc2 = mcep(X2)
        mc1 = fused(x)
    assert fused.last_path == "fused" and _lib.last_kernel() == "stft512_mcep_fused_fwd"
    assert float((mc1 - mc2).abs().max()) <= 1e-6 * float(mc2.abs().max())
    mcf, Xf, _ = _fused_with_spectrogram(stft, mcep, x, {"reflect": 1, "replicate": 2, "circular": 3}[mode])
    assert torch.equal(mcf, mc1)
    if P % 2 == 0:
        assert k2 == "stft512_fwd" and torch.equal(Xf, X2)     # the packed kernel on both sides: bit for bit
    else:
        assert float(((Xf - X2).abs() / X2.amax(-1, keepdim=True)).max()) < 2e-6
    X_ref = O.stft(host(x).astype(np.float64), 400, P, 512, center=center, mode=mode)
    err = np.abs(host(X2) - X_ref) / X_ref.max(-1, keepdims=True)
    assert err.max() < 2e-6, err.max()
    np.testing.assert_allclose(host(mc1), O.mcep(X_ref, 24, 0.42, 10), **MC32)
    w = torch.randn(mc1.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
    xa = x.clone().requires_grad_(True)
    (fused(xa) * w).sum().backward()
    assert fused.last_path == "fused"
    xb = x.clone().requires_grad_(True)
    (mcep(stft(xb)) * w).sum().backward()
    assert float((xa.grad - xb.grad).abs().max()) <= 1e-5 * float(xb.grad.abs().max())


@pytest.mark.parametrize("kw", [dict(zmean=True), dict(relative_floor=-40.0), dict(zmean=True, relative_floor=-25.0, mode="reflect"),
                                dict(zmean=True, mode="circular"), dict(relative_floor=-60.0, mode="replicate")])
@pytest.mark.parametrize("B,T,P,center", [(3, 16000, 80, True), (2, 15997, 80, True), (5, 1234, 160, True), (3, 2000, 80, False)])
def test_fused_zmean_and_relative_floor_equal_the_two_kernels_and_the_oracle(kw, B, T, P, center):
    """zmean (frame.py:139-140) and relative_floor (spec.py:174-176) -- with any pad mode -- through the ONE launch (round 6;
    dsa_stft_mcep_opts_fwd): `last_path == "fused"`, the spectrogram side product bit-identical to the two-kernel path's (the packed
    kernel serves these options now, with the same summation for the mean), mel-cepstra against the two kernels and the float64 oracle,
    the gradient against the two-stage gradient."""
    x = torch.randn(B, T, generator=torch.Generator().manual_seed(B * 13 + T)).to(DEV)
    x[0, : min(T, 500)] *= 1e-3                                   # a quiet stretch: the floor bites there
    # (no offset on top: a frame of 400 standard normal samples has a mean of ~0.05, which zmean removes; a large common offset is
    #  removed in float32 with an error that shows at the DC bins against a float64 reference -- in the reference's own float32 too)
    stft, mcep, fused = _modules(P, center, **kw)
    with torch.no_grad():
        X2 = stft(x)
        assert _lib.last_kernel() == "stft512_fwd"
        mc2 = mcep(X2)
        mc1 = fused(x)
    assert fused.last_path == "fused" and _lib.last_kernel() == "stft512_mcep_fused_fwd"
    assert float((mc1 - mc2).abs().max()) <= 1e-6 * float(mc2.abs().max())
    T_ = x.size(-1)
    N = ops.num_frames(T_, P)
    mcf = torch.empty(B, N, 25, device=DEV)
    Xf = torch.full((B, N, 257), float("nan"), device=DEV)
    images = ops.mcep_images(mcep.G, mcep.D, mcep.E, 512, 24)
    scratch = torch.zeros(_lib.SCRATCH_BYTES, dtype=torch.uint8, device=DEV)
    rf = kw.get("relative_floor")
    ops._call("dsa_stft_mcep_opts_fwd", x.data_ptr(), B, T_, 400, P, 512, stft.window.data_ptr(), stft.twiddle.data_ptr(), int(center),
              int(kw.get("zmean", False)), ops.pad_mode_code(kw.get("mode", "constant")), float(stft.eps), int(rf is not None),
              float(rf or 0.0), 24, mcep.n_iter, mcep.G.data_ptr(), mcep.D.data_ptr(), mcep.E.data_ptr(), mcep.alpha_vector.data_ptr(),
              _lib.F32, _lib.ALGO_AUTO, images.data_ptr(), scratch.data_ptr(), mcf.data_ptr(), None, Xf.data_ptr(), ops._stream())
    assert torch.equal(Xf, X2) and torch.equal(mcf, mc1)
    X_ref = O.stft(host(x).astype(np.float64), 400, P, 512, center=center, zmean=kw.get("zmean", False), mode=kw.get("mode", "constant"),
                   relative_floor=rf)
    err = np.abs(host(X2) - X_ref) / X_ref.max(-1, keepdims=True)
    # (2e-6 of the row maximum as everywhere; with zmean 6e-6: the float32 mean of 400 samples carries ~1e-7 of their size, which
    #  shows at the DC bins -- measured 3.1e-6)
    assert err.max() < (6e-6 if kw.get("zmean") else 2e-6), err.max()
    np.testing.assert_allclose(host(mc1), O.mcep(X_ref, 24, 0.42, 10), **MC32)
    w = torch.randn(mc1.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
    xa = x.clone().requires_grad_(True)
    (fused(xa) * w).sum().backward()
    assert fused.last_path == "fused"
    xb = x.clone().requires_grad_(True)
    (mcep(stft(xb)) * w).sum().backward()
    assert float((xa.grad - xb.grad).abs().max()) <= 1e-5 * float(xb.grad.abs().max())


def test_fused_contains_non_finite_samples_to_their_frames():
    x = torch.randn(2, 4000, generator=torch.Generator().manual_seed(9))
    x[0, 1000] = float("nan")
    x[1, 3999] = float("inf")
    xd = x.to(DEV)
    stft, mcep, fused = _modules()
    with torch.no_grad():
        mc = fused(xd)
        two = mcep(stft(xd))
    bad = ~torch.isfinite(two).all(-1)
    assert torch.equal(~torch.isfinite(mc).all(-1), bad) and 0 < int(bad.sum()) < 12
    assert float((mc[~bad] - two[~bad]).abs().max()) <= 1e-6 * float(two[~bad].abs().max())


def test_fused_routes_other_configurations_to_the_two_modules():
    x = torch.randn(2, 4000, generator=torch.Generator().manual_seed(1)).to(DEV)
    for kw in (dict(out_format="magnitude"),):
        stft = dsp.STFT(400, 80, 512, device=DEV, **kw)
        mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=2, device=DEV)
        f = dsp.fuse(stft, mcep)
        with torch.no_grad():
            y = f(x)
            assert f.last_path == "two-stage" and torch.equal(y, mcep(stft(x)))
    stft = dsp.STFT(320, 80, 512, device=DEV)
    mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=20, alpha=0.42, n_iter=2, device=DEV)
    f = dsp.fuse(stft, mcep)
    with torch.no_grad():
        y = f(x)
    assert f.last_path == "two-stage" and y.shape == (2, 50, 21)
    f64 = dsp.fuse(dsp.STFT(400, 80, 512, device=DEV, dtype=torch.float64),
                   dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=2, device=DEV, dtype=torch.float64))
    with torch.no_grad():
        f64(x.double())
    assert f64.last_path == "two-stage"
    with pytest.raises(ValueError):
        dsp.fuse(dsp.STFT(400, 80, 1024, device=DEV), mcep)


def test_fused_replays_from_a_hip_graph():
    stft, mcep, fused = _modules()
    x0 = torch.randn(8, 16000, generator=torch.Generator().manual_seed(2)).to(DEV)
    x1 = torch.randn(8, 16000, generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        g = dsp.Graphed(fused, x0)
        ref = mcep(stft(x1))
        out = g(x1)
        eager = fused(x1)     # an eager call next to the captured one: they do not share tile counters
        out2 = g(x1)
    assert torch.equal(out, eager) and torch.equal(out2, eager)
    assert float((out - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


@pytest.mark.parametrize("n_iter", [1, 3, 10])
@pytest.mark.parametrize("B,T", [(1024, 16000), (777, 12345), (163, 16000)])
def test_both_paths_are_deterministic_under_full_load(n_iter, B, T):
    """Round 4 met non-deterministic wrong frames in the fused launch when its STFT prologue ran on packed float32 instructions;
    round 5 reduced it from the failing kernel (tools/hazard/): the instruction is v_pk_add_f32 with the halves of
    src1 crossed (the +-i rotation of the radix-4 butterfly), lanes 48..63, a transient -- 1..170 bad frames per launch of 204 800
    depending on what else is packed, none with the rotations on scalar instructions, none in the stand-alone STFT kernels.  The guard:
    both shipped paths, three iteration counts (the effect needs a wave in its Newton phase next to one in its prologue; its rate
    changes with the phase mix) x full and ragged batches, 50 launches each -- every launch bit-identical to the first, the fused
    launch's spectrogram bit-identical to the stand-alone kernel's, the two paths' mel-cepstra equal to 1e-6."""
    stft = dsp.STFT(400, 80, 512, device=DEV)
    mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=n_iter, device=DEV)
    fused = dsp.fuse(stft, mcep)
    x = torch.randn(B, T, generator=torch.Generator().manual_seed(77 + n_iter)).to(DEV)
    launches = 50 if B == 1024 else 20
    with torch.no_grad():
        X2 = stft(x)
        ref2 = mcep(X2)
        ref1 = fused(x)
        assert fused.last_path == "fused"
        bad2 = bad1 = 0
        for _ in range(launches):
            bad2 += int((mcep(stft(x)) != ref2).any(-1).sum())
            bad1 += int((fused(x) != ref1).any(-1).sum())
        assert bad2 == 0 and bad1 == 0, (bad2, bad1)
    assert float((ref1 - ref2).abs().max()) <= 1e-6 * float(ref2.abs().max())
    # the spectrogram the fused launch computes (its side product when a gradient is wanted) against the stand-alone kernel, bit for bit
    xg = x.clone().requires_grad_(True)
    yg = fused(xg)
    Xs = [t for t in yg.grad_fn.saved_tensors if t.numel() == X2.numel() and t.size(-1) == 257]
    assert Xs and all(torch.equal(t.reshape(X2.shape), X2) for t in Xs)
