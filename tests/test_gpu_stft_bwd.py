"""GPU parity of the packed STFT backward / inverse-STFT kernel (csrc/stft_bwd_pk.h: one launch, the overlap-add of
the frame cotangents carried in registers along a run of passes, no workspace).

Checker: autograd of the float64 ATen port of the reference (oracle/torch_port.py: frame.py:130-140 -> window.py:185-193
-> fftr.py:117,145 -> spec.py:173) for the power format; the float64 numpy oracle (oracle.istft: istft.py:141-146) for
the inverse transform.  Tolerance (float32 kernel against float64): 2e-6 of the largest gradient entry of the utterance
-- the bound the two-kernel path it replaces was held to (test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

import diffsptk_amd as dsp
from diffsptk_amd import _lib
from oracle import oracle as O
from oracle import torch_port as TP

pytestmark = pytest.mark.gpu
DEV = "cuda"


def host(t):
    return t.detach().cpu().numpy()


def grad_and_kernel(y, x, cot, **kw):
    """Gradient of y w.r.t. x and the name of the kernel that produced it.  dsa_last_kernel() is per thread and the
    backward runs on autograd's worker thread: a hook on x sees it there, right after the backward function."""
    names = []
    h = x.register_hook(lambda g: names.append(_lib.last_kernel()))
    (gx,) = torch.autograd.grad(y, x, cot, **kw)
    h.remove()
    return gx, names[-1]


def ref_grad_power(x64, wt64, center):
    xr = x64.clone().requires_grad_(True)
    (TP.stft_power(xr, 400, 80, 512, center=center) * wt64).sum().backward()
    return xr.grad


@pytest.mark.parametrize("shape,center", [
    ((3, 16000), True),      # runs of the bench geometry (one item per (utterance, run))
    ((2, 5, 4001), True),    # odd length: utterance bases off the 16-byte grid (the synchronous staging path)
    ((7, 803), False),       # no centring; ragged last pass
    ((1, 401), True),        # N = 6: two passes, the second with two frames
    ((5, 79), True),         # a single frame per utterance
    ((2, 48000), False),     # long utterances: many runs per utterance
])
def test_power_gradient_matches_float64_autograd(shape, center):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    st = dsp.STFT(400, 80, 512, center=center, device=DEV)
    xd = x.to(DEV).requires_grad_(True)
    y = st(xd)
    wt = torch.randn(y.shape, generator=g)
    gx, kern = grad_and_kernel(y, xd, wt.to(DEV))
    assert kern == "stft512_bwd_pk"
    ref = ref_grad_power(x.double(), wt.double(), center)
    err = (gx.cpu().double() - ref).abs().amax(-1) / ref.abs().amax(-1)
    assert float(err.max()) < 2e-6


def test_magnitude_gradient_matches_float64_autograd():
    """out_format "magnitude" (spec.py:129: sqrt(|X|^2 + eps)) on the same kernel with one more factor per bin."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 8000, generator=g) * torch.tensor([1e-3, 1.0, 100.0]).view(3, 1)
    st = dsp.STFT(400, 80, 512, out_format="magnitude", device=DEV)
    xd = x.to(DEV).requires_grad_(True)
    y = st(xd)
    wt = torch.randn(y.shape, generator=g)
    gx, kern = grad_and_kernel(y, xd, wt.to(DEV))
    assert kern == "stft512_bwd_pk"
    xr = x.double().clone().requires_grad_(True)
    (torch.sqrt(TP.stft_power(xr, 400, 80, 512)) * wt.double()).sum().backward()
    err = (gx.cpu().double() - xr.grad).abs().amax(-1) / xr.grad.abs().amax(-1)
    assert float(err.max()) < 2e-6


@pytest.mark.parametrize("shape,center", [((3, 16000), True), ((2, 4001), True), ((5, 1283), False), ((2, 159), True)])
def test_ten_millisecond_period_runs_the_same_kernel(shape, center):
    """frame_period 160 (25 ms window, 10 ms hop at 16 kHz): its own instantiation (two carried / ten stored samples per
    lane and pass): power and magnitude gradients against float64 autograd, the inverse STFT against the oracle, and the
    run-partition invariance."""
    g = torch.Generator().manual_seed(shape[-1])
    x = torch.randn(*shape, generator=g)
    for fmt in ("power", "magnitude"):
        st = dsp.STFT(400, 160, 512, center=center, out_format=fmt, device=DEV)
        xd = x.to(DEV).requires_grad_(True)
        y = st(xd)
        wt = torch.randn(y.shape, generator=g)
        gx, kern = grad_and_kernel(y, xd, wt.to(DEV))
        assert kern == "stft512_bwd_pk"
        xr = x.double().clone().requires_grad_(True)
        Pw = TP.stft_power(xr, 400, 160, 512, center=center)
        ((Pw if fmt == "power" else torch.sqrt(Pw)) * wt.double()).sum().backward()
        err = (gx.cpu().double() - xr.grad).abs().amax(-1) / xr.grad.abs().amax(-1)
        assert float(err.max()) < 2e-6, fmt
    if center:
        stc = dsp.STFT(400, 160, 512, out_format="complex", device=DEV)
        ist = dsp.ISTFT(400, 160, 512, device=DEV)
        with torch.no_grad():
            yc = stc(x.to(DEV))
            xr32 = ist(yc, out_length=shape[-1])
        assert _lib.last_kernel() == "stft512_bwd_pk"
        ref = O.istft(host(yc).astype(np.complex128), 400, 160, w=O.window_table(400, "blackman", "power", True), out_length=shape[-1])
        np.testing.assert_allclose(host(xr32), ref, rtol=0, atol=2e-6 * np.abs(ref).max())
    if shape == (3, 16000):   # alone (100 frames = 25 passes in 6 runs) and among 1024 copies (4 runs): bit-identical
        st = dsp.STFT(400, 160, 512, device=DEV)
        wt1 = torch.randn(1, 100, 257, generator=g).to(DEV)
        outs = []
        for B in (1, 1024):
            xd = x[:1].expand(B, -1).contiguous().to(DEV).requires_grad_(True)
            (gx,) = torch.autograd.grad(st(xd), xd, wt1.expand(B, -1, -1).contiguous())
            assert torch.equal(gx, gx[:1].expand_as(gx))
            outs.append(gx[0].clone())
        assert torch.equal(outs[0], outs[1])


def test_gradient_does_not_depend_on_the_run_partition():
    """An utterance alone or among 3 / 300 copies (its 50 passes cut into 12 runs) and among 1024 copies (4 runs):
    bit-identical gradients -- the carried partial sums of a run's warm-up pass are the same
    arithmetic as the previous run's own."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 16000, generator=g)
    st = dsp.STFT(400, 80, 512, device=DEV)
    wt = torch.randn(1, 200, 257, generator=g).to(DEV)
    outs = []
    for B in (1, 3, 300, 1024):
        xd = x.expand(B, -1).contiguous().to(DEV).requires_grad_(True)
        gx, kern = grad_and_kernel(st(xd), xd, wt.expand(B, -1, -1).contiguous())
        assert kern == "stft512_bwd_pk"
        assert torch.equal(gx, gx[:1].expand_as(gx))
        outs.append(gx[0].clone())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # and two launches agree bitwise
    xd = x.expand(64, -1).contiguous().to(DEV).requires_grad_(True)
    y = st(xd)
    w64 = wt.expand(64, -1, -1).contiguous()
    a, = torch.autograd.grad(y, xd, w64, retain_graph=True)
    b, = torch.autograd.grad(y, xd, w64)
    assert torch.equal(a, b)


def test_bench_size_gradient_sampled_against_float64():
    """BASELINE configs[2] at the configs[4] shard size: 1024 utterances x 1 s; eight sampled utterances against float64."""
    g = torch.Generator().manual_seed(12)
    x = torch.randn(1024, 16000, generator=g)
    st = dsp.STFT(400, 80, 512, device=DEV)
    xd = x.to(DEV).requires_grad_(True)
    y = st(xd)
    wt = torch.randn(y.shape, generator=g)
    gx, kern = grad_and_kernel(y, xd, wt.to(DEV))
    assert kern == "stft512_bwd_pk" and torch.isfinite(gx).all()
    sel = [0, 1, 255, 256, 511, 700, 1022, 1023]
    ref = ref_grad_power(x[sel].double(), wt[sel].double(), True)
    err = (gx[sel].cpu().double() - ref).abs().amax(-1) / ref.abs().amax(-1)
    assert float(err.max()) < 2e-6


def test_complex_cotangent_and_inverse_stft_against_the_oracle():
    g = torch.Generator().manual_seed(13)
    for shape in ((3, 16000), (2, 4000), (1, 801)):
        x = torch.randn(*shape, generator=g)
        xd = x.to(DEV).requires_grad_(True)
        stc = dsp.STFT(400, 80, 512, out_format="complex", device=DEV)
        yc = stc(xd)
        wc = torch.randn(yc.shape + (2,), generator=g)
        gc, kern = grad_and_kernel(yc, xd, torch.view_as_complex(wc.to(DEV)))
        assert kern == "stft512_bwd_pk"
        # float64 reference: the adjoint of frame -> window -> rfft
        xr = x.double().clone().requires_grad_(True)
        fr = TP.frame(xr, 400, 80) * TP.window_table(400)
        Y = torch.fft.rfft(torch.nn.functional.pad(fr, (0, 112)), n=512)
        (torch.view_as_real(Y) * wc.double()).sum().backward()
        err = (gc.cpu().double() - xr.grad).abs().amax(-1) / xr.grad.abs().amax(-1)
        assert float(err.max()) < 2e-6
        # inverse STFT of the same spectra (divides by the overlap-added squared window as it stores)
        ist = dsp.ISTFT(400, 80, 512, device=DEV)
        with torch.no_grad():
            xr32 = ist(yc.detach(), out_length=shape[-1])
        assert _lib.last_kernel() == "stft512_bwd_pk"
        ref = O.istft(host(yc).astype(np.complex128), 400, 80, w=O.window_table(400, "blackman", "power", True), out_length=shape[-1])
        np.testing.assert_allclose(host(xr32), ref, rtol=0, atol=2e-6 * np.abs(ref).max())


def test_other_geometries_keep_their_kernels():
    x = torch.randn(2, 4000, generator=torch.Generator().manual_seed(14)).to(DEV)
    for st in (dsp.STFT(320, 80, 512, device=DEV), dsp.STFT(400, 100, 512, device=DEV), dsp.STFT(400, 80, 512, zmean=True, device=DEV),
               dsp.STFT(400, 80, 512, out_format="log-magnitude", device=DEV)):
        xd = x.clone().requires_grad_(True)
        y = st(xd)
        gx, kern = grad_and_kernel(y, xd, torch.ones_like(y))
        assert kern == "stft512_bwd" and torch.isfinite(gx).all()


def test_nonfinite_sample_stays_in_the_frames_that_contain_it():
    """A NaN sample poisons the spectra of the five frames that contain it, hence the gradient over their support
    (80 n - 200 <= t < 80 n + 200 for n = 48 .. 52) -- and nothing else (the reference's dense autograd behaves the same)."""
    x = torch.randn(1, 8000, generator=torch.Generator().manual_seed(15))
    x[0, 4000] = float("nan")
    xd = x.to(DEV).requires_grad_(True)
    st = dsp.STFT(400, 80, 512, device=DEV)
    st(xd).sum().backward()
    bad = ~torch.isfinite(xd.grad[0]).cpu()
    lo, hi = 80 * 48 - 200, 80 * 52 + 200
    assert not bad[:lo].any() and not bad[hi:].any() and bad[lo + 100:hi - 100].all()
