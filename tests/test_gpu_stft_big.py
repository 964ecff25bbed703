"""The packed STFT kernel for fft_length 1024 / 2048 (csrc/stft_pk_big.h, round 6; stft.py:86-104 at the 44.1 / 48 kHz set-ups of
utils/public.py:61-104) against the float64 oracle, the generic kernel and float64 autograd.  Tolerance of float32 spectra as
everywhere (tests/test_gpu_parity.py): |y - y64| <= 1e-4 |y64| + 2e-6 max_k y64[frame]."""
import numpy as np
import pytest
import torch

import diffsptk_amd as dsp
from diffsptk_amd import _lib
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def spec_close(y, y64, rtol=1e-4, rel_rowmax=2e-6):
    y, y64 = np.asarray(y, np.float64), np.asarray(y64, np.float64)
    bound = rtol * np.abs(y64) + rel_rowmax * np.abs(y64).max(-1, keepdims=True)
    bad = np.abs(y - y64) > bound
    assert not bad.any(), f"{bad.sum()} bins out of tolerance, worst {np.abs(y - y64).max():.3e}"


GEOMETRIES = [
    # (frame_length, frame_period, fft_length, kernel): every (S, NR) instantiation, frame_length = fft_length, short frames
    (1200, 240, 2048, "stft2048_fwd"), (800, 200, 1024, "stft1024_fwd"), (1024, 256, 1024, "stft1024_fwd"),
    (2048, 512, 2048, "stft2048_fwd"), (1600, 400, 2048, "stft2048_fwd"), (600, 150, 1024, "stft1024_fwd"),
    (882, 220, 1024, "stft1024_fwd"), (1102, 220, 2048, "stft2048_fwd"), (64, 16, 1024, "stft1024_fwd"),
]


@pytest.mark.parametrize("fl,fp,nfft,kernel", GEOMETRIES)
@pytest.mark.parametrize("center", [True, False])
def test_against_the_oracle_and_the_generic_kernel(fl, fp, nfft, kernel, center):
    g = torch.Generator().manual_seed(fl + nfft + center)
    for B, T in ((3, 6 * nfft), (2, 2 * fl + 2 * fp + 2), (1, fl // 2 * 2)):
        if T % 2:
            T += 1
        x = torch.randn(B, T, generator=g)
        x[0, : min(T, 300)] *= 1e-3                      # a quiet stretch: bins far below the row maximum
        y64 = O.stft(x.double().numpy(), fl, fp, nfft, center=center)
        st = dsp.STFT(fl, fp, nfft, center=center, device=DEV)
        y = st(x.to(DEV))
        # the packed kernel needs an even left pad: frame_length % 4 == 0 when centred (882, 1102: the generic kernel)
        even_left = (not center) or (fl // 2) % 2 == 0
        assert _lib.last_kernel() == (kernel if even_left else "row_fft_generic"), (_lib.last_kernel(), fl, center)
        assert y.shape == y64.shape
        spec_close(y.cpu().numpy(), y64)
        # ... and the generic kernel on the same input (DSA_ALGO_GENERIC): same tolerance against each other's float64 value
        from diffsptk_amd import ops

        yg = ops.StftFn.apply(x.to(DEV), st.window, st.twiddle, fl, fp, nfft, center, False, "constant", 1e-9, None, 3, _lib.ALGO_GENERIC)
        spec_close(yg.cpu().numpy(), y64)


def test_bench_size_properties_parseval_and_partition_invariance():
    """512 utterances x 1 s @ 48 kHz (the size of bench.py's 48 kHz rows): Parseval per frame against the windowed frame's energy,
    bit-identical results whether an utterance travels alone or in the batch (a pass never mixes utterances), repeat launches
    bit-identical, and a sample of frames against the oracle."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(512, 48000, generator=g).to(DEV)
    for fl, fp, nfft in ((1200, 240, 2048), (800, 200, 1024)):
        st = dsp.STFT(fl, fp, nfft, device=DEV)
        y = st(x)
        assert _lib.last_kernel() == f"stft{nfft}_fwd"
        assert torch.equal(st(x), y)
        for u in (0, 17, 511):
            assert torch.equal(st(x[u:u + 1]), y[u:u + 1]), u
        assert torch.equal(st(x[100:300]), y[100:300])
        # Parseval: sum_k c_k |X_k|^2 = nfft * sum_n (w x)_n^2, c = 1 at DC / Nyquist, 2 in between
        fr = dsp.Window(fl, nfft, device=DEV)(dsp.Frame(fl, fp)(x[:8]))
        e_time = fr.double().square().sum(-1) * nfft
        yy = (y[:8].double() - 1e-9)
        e_freq = 2 * yy.sum(-1) - yy[..., 0] - yy[..., -1]
        assert float(((e_freq - e_time).abs() / e_time).max()) < 2e-5
        sel = [0, 255, 511]
        y64 = O.stft(x[sel].double().cpu().numpy(), fl, fp, nfft)
        spec_close(y[sel].cpu().numpy(), y64)


def test_non_finite_samples_stay_in_their_frames():
    fl, fp, nfft = 1200, 240, 2048
    x = torch.randn(2, 24000, generator=torch.Generator().manual_seed(5))
    xb = x.clone()
    xb[1, 7000] = float("nan")
    xb[0, 23999] = float("inf")
    st = dsp.STFT(fl, fp, nfft, device=DEV)
    y, yb = st(x.to(DEV)), st(xb.to(DEV))
    N = y.shape[1]
    t0 = torch.arange(N) * fp - fl // 2                      # first sample of frame n (centred)
    for (u, pos) in ((1, 7000), (0, 23999)):
        hit = (t0 <= pos) & (pos < t0 + fl)
        assert hit.any()
        assert not torch.isfinite(yb[u][hit.to(DEV)]).any(dim=-1).any()      # every bin of a frame that contains it
        assert torch.equal(yb[u][~hit.to(DEV)], y[u][~hit.to(DEV)])           # nothing else moved, bit for bit


@pytest.mark.parametrize("fl,fp,nfft", [(1200, 240, 2048), (800, 200, 1024)])
def test_gradient_against_float64_autograd(fl, fp, nfft):
    """The packed backward of stft_bwd_pk_big.h behind the packed forward (see the tests below for the kernel against the generic one)."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 6000, generator=g)
    cot = torch.randn(2, (6000 - 1) // fp + 1, nfft // 2 + 1, generator=g)
    xs = x.to(DEV).requires_grad_(True)
    y = dsp.STFT(fl, fp, nfft, device=DEV)(xs)
    assert _lib.last_kernel() == f"stft{nfft}_fwd"
    (y * cot.to(DEV)).sum().backward()
    xd = x.double().to(DEV).requires_grad_(True)
    yd = dsp.STFT(fl, fp, nfft, device=DEV, dtype=torch.float64)(xd)
    (yd * cot.double().to(DEV)).sum().backward()
    err = float((xs.grad.double() - xd.grad).abs().max() / xd.grad.abs().max())
    assert err < 3e-6, err


def _grad(st, x, cot):
    xs = x.detach().clone().requires_grad_(True)
    (st(xs) * cot).sum().backward()
    return xs.grad


BWD_GEOMETRIES = [(1200, 240, 2048), (800, 200, 1024), (1024, 256, 1024), (2048, 512, 2048), (1600, 400, 2048), (600, 150, 1024),
                  (600, 150, 2048), (64, 16, 1024), (1200, 1300, 2048)]


@pytest.mark.parametrize("fl,fp,nfft", BWD_GEOMETRIES)
@pytest.mark.parametrize("center", [True, False])
def test_packed_backward_against_float64_autograd_and_the_generic_backward(fl, fp, nfft, center, monkeypatch):
    """stft_big_bwd_pk_kernel (round 6): every (S, NR) instantiation, frame_length = fft_length, a period longer than the frame
    (samples no frame reaches: zero gradient), ragged utterances (several runs per utterance, one-frame utterances), against float64 autograd of
    the generic float64 kernels and the generic float32 backward."""
    g = torch.Generator().manual_seed(fl + nfft + 7 * center)
    for B, T in ((3, 12 * nfft), (2, 2 * fl + 2 * fp + 2), (70, 3 * nfft), (1, max(fl // 2 * 2, 2))):
        x = torch.randn(B, T, generator=g).to(DEV)
        st = dsp.STFT(fl, fp, nfft, center=center, device=DEV)
        st64 = dsp.STFT(fl, fp, nfft, center=center, device=DEV, dtype=torch.float64)
        N = (T - 1) // fp + 1
        cot = torch.randn(B, N, nfft // 2 + 1, generator=g).to(DEV)
        monkeypatch.setenv("DSA_STFT_BIG_BWD", "1")
        gp = _grad(st, x, cot)
        monkeypatch.setenv("DSA_STFT_BIG_BWD", "0")
        gg = _grad(st, x, cot)
        g64 = _grad(st64, x.double(), cot.double())
        scale = float(g64.abs().max())
        assert torch.isfinite(gp).all()
        assert float((gp.double() - g64).abs().max()) < 3e-6 * scale, (B, T, float((gp.double() - g64).abs().max()) / scale)
        assert float((gp - gg).abs().max()) < 3e-6 * scale


def test_packed_backward_bench_size_partition_invariance_and_non_finite_containment(monkeypatch):
    """512 utterances x 1 s @ 48 kHz: an utterance's gradient is the same bits alone or in the batch and however the chip cuts it into runs
    (1 utterance: many runs; 512: four), repeat launches identical; a non-finite sample touches only the samples of the frames that contain it."""
    g = torch.Generator().manual_seed(11)
    fl, fp, nfft = 1200, 240, 2048
    x = torch.randn(512, 48000, generator=g).to(DEV)
    st = dsp.STFT(fl, fp, nfft, device=DEV)
    cot = torch.randn(512, (48000 - 1) // fp + 1, 1025, generator=g).to(DEV)
    gb = _grad(st, x, cot)
    assert torch.equal(gb, _grad(st, x, cot))
    for u in (0, 17, 511):
        assert torch.equal(_grad(st, x[u:u + 1], cot[u:u + 1])[0], gb[u])
    assert torch.equal(_grad(st, x[100:103], cot[100:103]), gb[100:103])
    xb = x[:4].clone()
    xb[1, 7000] = float("nan")
    gn = _grad(st, xb, cot[:4])
    t = torch.arange(48000, device=DEV)
    n_lo, n_hi = (7000 + fl // 2 - fl) // fp + 1, (7000 + fl // 2) // fp          # frames that contain sample 7000
    touched = (t >= n_lo * fp - fl // 2) & (t < n_hi * fp - fl // 2 + fl)
    assert torch.equal(gn[[0, 2, 3]], gb[[0, 2, 3]])
    assert torch.equal(gn[1][~touched], gb[1][~touched]) and not torch.isfinite(gn[1][touched]).all()
