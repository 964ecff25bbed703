"""Repeat-launch determinism of every kernel family whose waves share a SIMD with waves in another phase (a packed
float32 instruction with a set op_sel bit has delivered transient wrong values in exactly that situation; the shipped kernels of
these families execute none next to a different instruction stream -- this test is the regression guard for that audit).  Sizes
fill the chip; every output of every launch is compared bit for bit with the first launch's."""
import pytest
import torch

import diffsptk_amd as dsp

pytestmark = pytest.mark.gpu
DEV = "cuda"
LAUNCHES = 12


def _repeat(fn):
    with torch.no_grad():
        ref = fn()
        ref = ref if isinstance(ref, (tuple, list)) else (ref,)
        for _ in range(LAUNCHES):
            out = fn()
            out = out if isinstance(out, (tuple, list)) else (out,)
            for a, b in zip(out, ref):
                assert torch.equal(a, b)


@pytest.fixture(scope="module")
def wave():
    g = torch.Generator().manual_seed(0)
    return torch.randn(512, 16000, generator=g).to(DEV), torch.randn(64, 48000, generator=g).to(DEV)


def test_untuned_48khz_analysis_repeats_bit_for_bit(wave):
    _, x48 = wave
    for fl, fp, nfft, M in ((1200, 240, 2048, 49), (800, 200, 1024, 34)):
        X = dsp.STFT(fl, fp, nfft, device=DEV)(x48)
        m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=DEV)
        _repeat(lambda: m(X))


def test_mgcep_one_launch_step_repeats_bit_for_bit(wave):
    x, _ = wave
    X = dsp.STFT(400, 80, 512, device=DEV)(x[:256])
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=5, device=DEV)
    _repeat(lambda: mg(X))


def test_filter_bank_and_lpc_branches_repeat_bit_for_bit(wave):
    x, _ = wave
    stft = dsp.STFT(400, 80, 512, device=DEV)
    fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, device=DEV)
    _repeat(lambda: dsp.fuse(stft, fb)(x))
    fl = dsp.fuse(dsp.Frame(400, 80), dsp.Window(400, device=DEV), dsp.LPC(400, 24, eps=1e-5, device=DEV))
    _repeat(lambda: fl(x))


def test_backward_kernels_repeat_bit_for_bit(wave):
    x, _ = wave
    stft = dsp.STFT(400, 80, 512, device=DEV)
    mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=DEV)
    fl = dsp.fuse(dsp.Frame(400, 80), dsp.Window(400, device=DEV), dsp.LPC(400, 24, eps=1e-5, device=DEV))

    def grads(f):
        xg = x[:256].clone().requires_grad_(True)
        with torch.enable_grad():
            f(xg).sum().backward()
        return xg.grad

    _repeat(lambda: grads(lambda v: mcep(stft(v))))
    _repeat(lambda: grads(fl))


def test_non_finite_frames_stay_in_their_rows_in_the_round_5_kernels():
    """A frame is a column of every matrix product and a quad of the solves: a non-finite input frame must not reach any other
    frame's result -- two-wave mel-cepstral backward (gradient w.r.t. the spectrogram), the one-launch mgcep step, the binary16
    residual of the 48 kHz set-ups.  The clean frames' results are bit-identical to a launch without the poisoned frame's data."""
    g = torch.Generator().manual_seed(7)
    X = (torch.randn(700, 257, generator=g).square() + 0.05).to(DEV)
    bad = [3, 16, 47, 333, 699]
    Xb = X.clone()
    Xb[bad[0], 5] = float("nan")
    Xb[bad[1], 100] = float("inf")
    Xb[bad[2], :] = float("nan")
    Xb[bad[3], 256] = float("nan")
    Xb[bad[4], 0] = float("inf")
    clean = torch.ones(700, dtype=torch.bool)
    clean[bad] = False
    mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=DEV)
    w = torch.randn(25, generator=g).to(DEV)

    def grad(Xin):
        Xg = Xin.clone().requires_grad_(True)
        (mcep(Xg) * w).sum().backward()
        return Xg.grad

    g0, g1 = grad(X), grad(Xb)
    assert torch.isfinite(g1[clean]).all() and torch.equal(g1[clean], g0[clean])
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=3, device=DEV)
    with torch.no_grad():
        y0, y1 = mg(X), mg(Xb)
    assert torch.isfinite(y1[clean]).all() and torch.equal(y1[clean], y0[clean])
    m48 = dsp.MelCepstralAnalysis(fft_length=1024, cep_order=34, alpha=0.55, n_iter=4, device=DEV)
    X48 = (torch.randn(300, 513, generator=g).square() + 0.05).to(DEV)
    X48b = X48.clone()
    X48b[17, 3] = float("nan")
    X48b[64, :] = float("inf")
    c48 = torch.ones(300, dtype=torch.bool)
    c48[[17, 64]] = False
    with torch.no_grad():
        z0, z1 = m48(X48), m48(X48b)
    assert torch.isfinite(z1[c48]).all() and torch.equal(z1[c48], z0[c48])
