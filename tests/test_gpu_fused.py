"""GPU parity of the fused STFT -> mel filter bank (-> MFCC) kernel (SURVEY 8(f) row 1 in one launch:
stft.py:237-241 + fbank.py:306-321 / mfcc.py:244-256; csrc/stft_pk.h, FBM variants).

The checker is the float64 oracle (oracle.stft -> oracle.fbank / oracle.mfcc) on seeded inputs and on data.wav, and
the library's own two-stage path (already pinned to the reference's goldens in test_gpu_parity.py).
Tolerance (float32 in, float32 out): the filter-bank outputs are LOGS of sums of positive float32 power values, so
|y - y64| <= 2e-5 absolute (a relative 2e-5 of the channel sum: the float32 STFT's own rounding, cf. the spectra
tolerance in test_gpu_parity.py; the segmented scan adds 2e-7) -- the same bound the two-stage path meets."""
import numpy as np
import pytest
import torch

import diffsptk_amd as dsp
from conftest import wav_float
from diffsptk_amd import _lib, ops
from diffsptk_amd.utils import tables
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def host(t):
    return t.detach().cpu().numpy()


def oracle_fbank(x, P, C, sr, *, center=True, use_power=False, floor=1e-5, gamma=0.0, f_min=0.0, f_max=None):
    X = O.stft(np.asarray(x, np.float64), 400, P, 512, center=center, eps=1e-9)
    H = np.asarray(tables.fbank_matrix(512, C, sr, f_min, f_max, "htk", None))
    y, _ = O.fbank(X, H, floor, gamma, use_power)
    return y


@pytest.mark.parametrize("C,sr,P,use_power,gamma,center", [
    (40, 16000, 80, True, 0.0, True),
    (40, 16000, 80, False, 0.0, True),
    (80, 16000, 160, True, 0.0, True),
    (80, 22050, 128, False, -0.5, False),
    (24, 16000, 100, True, 0.3, True),
    (126, 48000, 80, False, 0.0, True),
    (3, 8000, 80, True, 0.0, False),
])
def test_fused_fbank_matches_oracle_and_two_stage(C, sr, P, use_power, gamma, center):
    g = torch.Generator().manual_seed(C + P)
    # ragged frame counts (N % 4 = 1, 2, 3, 0), a batch with leading dimensions, levels 1e-3 .. 1e+3
    for shape in ((3, 2, 1601), (5, 803), (2, 4000 + P), (1, 16000)):
        x = torch.randn(*shape, generator=g) * (10.0 ** torch.empty(*shape[:-1], 1).uniform_(-3, 3, generator=g))
        xd = x.to(DEV)
        stft = dsp.STFT(400, P, 512, center=center, device=DEV)
        fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=C, sample_rate=sr, use_power=use_power, gamma=gamma, device=DEV)
        fused = dsp.fuse(stft, fb)
        with torch.no_grad():
            y = fused(xd)
            assert fused.last_path == "fused" and _lib.last_kernel() == "stft512_fbank_fwd"
            y2 = fb(stft(xd))
        ref = oracle_fbank(x.numpy(), P, C, sr, center=center, use_power=use_power, gamma=gamma)
        assert y.shape == ref.shape
        if gamma == 0.0:
            np.testing.assert_allclose(host(y), ref, rtol=0, atol=2e-5)
            np.testing.assert_allclose(host(y), host(y2), rtol=0, atol=2e-5)
        else:
            np.testing.assert_allclose(host(y), ref, rtol=3e-5, atol=3e-5)
            np.testing.assert_allclose(host(y), host(y2), rtol=3e-5, atol=3e-5)


def test_fused_fbank_datawav_and_mfcc(golden):
    """data.wav (the reference's test signal): filter bank and MFCC through the fused launch against the oracle."""
    x = wav_float(golden("datawav")["pcm"], np.float64)
    xd = torch.from_numpy(x).float().to(DEV)
    stft = dsp.STFT(400, 80, 512, device=DEV)
    fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, device=DEV)
    mf = dsp.MFCC(fft_length=512, mfcc_order=12, n_channel=40, sample_rate=16000, lifter=22, device=DEV)
    with torch.no_grad():
        y = dsp.fuse(stft, fb)(xd)
        c = dsp.fuse(stft, mf)(xd)
    X = O.stft(xd.cpu().numpy().astype(np.float64), 400, 80, 512, eps=1e-9)   # the float32 samples the kernel saw
    H = np.asarray(tables.fbank_matrix(512, 40, 16000, 0.0, None, "htk", None))
    yref, _ = O.fbank(X, H, 1e-5, 0.0, False)
    np.testing.assert_allclose(host(y), yref, rtol=0, atol=2e-5)
    cref = O.mfcc(X, H, 12, 22, 1e-5, 0.0)[0]
    np.testing.assert_allclose(host(c), cref, rtol=1e-5, atol=3e-4)
    with torch.no_grad():
        np.testing.assert_allclose(host(c), host(mf(stft(xd))), rtol=1e-5, atol=3e-4)


def test_fused_fbank_bench_size_properties_and_determinism():
    """BASELINE configs[4] shard size (1024 utterances x 1 s): equals the two-stage path; a gain g on the waveform adds
    2 log g to every power-domain channel above the floor; two launches are bit-identical; every utterance of a batch
    of copies gives the same rows (passes are dealt to different waves)."""
    x = torch.randn(1024, 16000, generator=torch.Generator().manual_seed(5)).to(DEV)
    stft = dsp.STFT(400, 80, 512, device=DEV)
    fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, use_power=True, device=DEV)
    fused = dsp.fuse(stft, fb)
    with torch.no_grad():
        y = fused(x)
        assert fused.last_path == "fused" and y.shape == (1024, 200, 40)
        assert torch.equal(y, fused(x))
        y2 = fb(stft(x))
        np.testing.assert_allclose(host(y), host(y2), rtol=0, atol=2e-5)
        y4 = fused(2 * x)
        np.testing.assert_allclose(host(y4 - y), np.full(y.shape, np.log(4.0)), rtol=0, atol=1e-4)
        xc = x[:1].expand(64, -1).contiguous()
        yc = fused(xc)
        assert torch.equal(yc, yc[:1].expand_as(yc))


def test_fused_falls_back_when_it_must():
    """Everything the fused kernel does not cover runs the two stages -- same numbers, gradients available."""
    x = torch.randn(2, 4000, generator=torch.Generator().manual_seed(6)).to(DEV)
    fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, device=DEV)
    for stft in (dsp.STFT(400, 80, 512, mode="reflect", device=DEV), dsp.STFT(320, 80, 512, device=DEV),
                 dsp.STFT(400, 80, 512, out_format="magnitude", device=DEV), dsp.STFT(400, 80, 512, zmean=True, device=DEV)):
        f = dsp.fuse(stft, fb)
        with torch.no_grad():
            assert torch.equal(f(x), fb(stft(x))) and f.last_path == "two-stage"
    stft = dsp.STFT(400, 80, 512, device=DEV)
    erb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, erb_factor=1.0, device=DEV)
    f = dsp.fuse(stft, erb)   # overlapping ERB filters: no scan plan
    with torch.no_grad():
        assert ops.fbank_scan_plan(erb.H) is None and torch.equal(f(x), erb(stft(x))) and f.last_path == "two-stage"
    yE = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, out_format="yE", device=DEV)
    f = dsp.fuse(stft, yE)
    with torch.no_grad():
        assert torch.equal(f(x), yE(stft(x))) and f.last_path == "two-stage"
    fbl = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, learnable=True, device=DEV)
    f = dsp.fuse(stft, fbl)   # a learnable basis: its gradient needs the differentiable stages
    xg = x.clone().requires_grad_(True)
    f(xg).sum().backward()
    assert f.last_path == "two-stage" and xg.grad is not None and fbl.H.grad is not None
    with pytest.raises(ValueError):
        dsp.fuse(dsp.STFT(400, 80, 1024, device=DEV), fb)


def test_fused_nonfinite_sample_stays_in_its_frames():
    """A NaN sample makes every channel of the frames that contain it non-finite (as the reference's dense product does:
    every bin of such a frame is NaN) and leaves every other frame untouched."""
    x = torch.randn(1, 8000, generator=torch.Generator().manual_seed(7))
    x[0, 4000] = float("nan")
    xd = x.to(DEV)
    stft = dsp.STFT(400, 80, 512, device=DEV)
    fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, use_power=True, device=DEV)
    with torch.no_grad():
        y = dsp.fuse(stft, fb)(xd)
        y2 = fb(stft(xd))
    bad = torch.zeros(y.shape[:-1], dtype=torch.bool, device=DEV)
    bad[0, 48:53] = True   # frames n with 80 n - 200 <= 4000 < 80 n + 200
    for out in (y, y2):
        assert torch.equal((~torch.isfinite(out)).all(-1), bad) and torch.equal((~torch.isfinite(out)).any(-1), bad)
    np.testing.assert_allclose(host(y[~bad]), host(y2[~bad]), rtol=0, atol=2e-5)


def _float64_fbank_gradient(x, cot, C, sr, use_power, gamma, floor=1e-5):
    from oracle import torch_port as TP
    H = torch.from_numpy(np.asarray(tables.fbank_matrix(512, C, sr, 0.0, None, "htk", None))).double()
    xr = x.double().clone().requires_grad_(True)
    Pw = TP.stft_power(xr, 400, 80, 512)
    s = torch.clip((Pw if use_power else torch.sqrt(Pw)) @ H, min=floor)   # fbank.py:306-321
    y = torch.log(s) if gamma == 0 else (torch.pow(s, gamma) - 1) / gamma
    (y * cot.double()).sum().backward()
    return xr.grad


@pytest.mark.parametrize("C,sr,use_power,gamma", [(40, 16000, True, 0.0), (40, 16000, False, 0.0), (80, 22050, True, -0.3),
                                                  (24, 16000, False, 0.4), (126, 48000, True, 0.0)])
def test_fused_gradient_matches_float64_autograd_and_the_two_stage_path(C, sr, use_power, gamma):
    """With a gradient needed the fused launch still runs (forward) and its backward -- channel cotangents spread over the
    bins (dsa_fbank_bins_bwd), then the STFT backward kernel -- never needs the spectrogram.  Against float64 autograd of
    the reference's formulation and against the two differentiable stages: 1e-5 of the utterance's largest gradient entry
    (measured 4e-7 .. 4e-6; levels 1e-2 .. 30)."""
    g = torch.Generator().manual_seed(C)
    x = torch.randn(3, 8000, generator=g) * torch.tensor([1e-2, 1.0, 30.0]).view(3, 1)
    stft = dsp.STFT(400, 80, 512, device=DEV)
    fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=C, sample_rate=sr, use_power=use_power, gamma=gamma, device=DEV)
    fused = dsp.fuse(stft, fb)
    xa = x.to(DEV).requires_grad_(True)
    ya = fused(xa)
    assert fused.last_path == "fused"
    cot = torch.randn(ya.shape, generator=g)
    (ga,) = torch.autograd.grad(ya, xa, cot.to(DEV))
    xb = x.to(DEV).requires_grad_(True)
    (gb,) = torch.autograd.grad(fb(stft(xb)), xb, cot.to(DEV))
    ref = _float64_fbank_gradient(x, cot, C, sr, use_power, gamma)
    scale = ref.abs().amax(-1)
    assert float(((ga.cpu().double() - ref).abs().amax(-1) / scale).max()) < 1e-5
    assert float(((ga - gb).abs().amax(-1).cpu().double() / scale).max()) < 1e-5


@pytest.mark.parametrize("shape,P,center", [((2, 4001), 80, True), ((3, 1283), 160, False), ((1, 5, 401), 160, True)])
def test_fused_gradient_ragged_lengths_and_the_ten_millisecond_period(shape, P, center):
    """Odd lengths (utterances off the 16-byte grid), uncentred framing, leading batch dimensions and frame_period 160:
    the fused forward + its two-launch backward against the two differentiable stages and float64."""
    from oracle import torch_port as TP
    g = torch.Generator().manual_seed(shape[-1])
    x = torch.randn(*shape, generator=g)
    stft = dsp.STFT(400, P, 512, center=center, device=DEV)
    fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, use_power=True, device=DEV)
    fused = dsp.fuse(stft, fb)
    xa = x.to(DEV).requires_grad_(True)
    ya = fused(xa)
    assert fused.last_path == "fused"
    cot = torch.randn(ya.shape, generator=g)
    (ga,) = torch.autograd.grad(ya, xa, cot.to(DEV))
    xb = x.to(DEV).requires_grad_(True)
    (gb,) = torch.autograd.grad(fb(stft(xb)), xb, cot.to(DEV))
    H = torch.from_numpy(np.asarray(tables.fbank_matrix(512, 40, 16000, 0.0, None, "htk", None))).double()
    xr = x.double().clone().requires_grad_(True)
    y64 = torch.log(torch.clip(TP.stft_power(xr, 400, P, 512, center=center) @ H, min=1e-5))
    (y64 * cot.double()).sum().backward()
    scale = xr.grad.abs().amax(-1)
    assert float(((ga.cpu().double() - xr.grad).abs().amax(-1) / scale).max()) < 1e-5
    assert float(((ga - gb).abs().amax(-1).cpu().double() / scale).max()) < 1e-5


def test_fused_mfcc_gradient_and_floor_clamp():
    """MFCC on top of the fused launch (amplitude domain, DCT + lifter as one row product) back-propagates like the two
    stages; a silent utterance sits on the floor in every channel: its gradient is exactly zero, as torch.clip's is."""
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 4000, generator=g)
    x[2] = 0.0
    stft = dsp.STFT(400, 80, 512, device=DEV)
    mf = dsp.MFCC(fft_length=512, mfcc_order=12, n_channel=40, sample_rate=16000, lifter=22, device=DEV)
    fused = dsp.fuse(stft, mf)
    xa = x.to(DEV).requires_grad_(True)
    ya = fused(xa)
    cot = torch.randn(ya.shape, generator=g).to(DEV)
    (ga,) = torch.autograd.grad(ya, xa, cot)
    assert fused.last_path == "fused"
    xb = x.to(DEV).requires_grad_(True)
    (gb,) = torch.autograd.grad(mf(stft(xb)), xb, cot)
    assert torch.equal(ga[2], torch.zeros_like(ga[2])) and torch.equal(gb[2], torch.zeros_like(gb[2]))
    keep = [0, 1, 3]
    scale = gb[keep].abs().amax(-1)
    assert float(((ga[keep] - gb[keep]).abs().amax(-1) / scale).max()) < 1e-5
