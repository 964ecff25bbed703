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
    """The backward of these geometries stays on the generic kernels; the forward that feeds it is the packed one."""
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
