"""Two HIP streams at full load: the packed STFT kernels on stream A against the kernels that keep two waves per SIMD in different
phases (mel-cepstral forward / fused forward / two-wave backward, the one-launch mgcep step, the 48 kHz solver) on stream B --
the situation `dist.analyze_chunked_overlap(alternate_streams=True)` and `bench.py --streams 2` create, where a wave of one kernel
can share a SIMD with a wave of a DIFFERENT instruction stream.  DESIGN.md 4: a packed float32 instruction with a set op_sel bit
has delivered transient wrong values in exactly that situation; since round 6 no kernel of the library contains one
(tests/test_host_cpu.py::test_no_crossed_packed_float32) -- this is the run-time side of that invariant.  1 024 utterances,
>= 50 rounds per pairing, EVERY output of EVERY launch compared bit for bit with the single-stream result."""
import pytest
import torch

import diffsptk_amd as dsp

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROUNDS = 50


@pytest.fixture(scope="module")
def setup():
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1024, 16000, generator=g).to(DEV)
    stft = dsp.STFT(400, 80, 512, device=DEV)
    mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=DEV)
    with torch.no_grad():
        S = stft(x)
    return x, stft, mcep, S


def _tup(v):
    return v if isinstance(v, (tuple, list)) else (v,)


def _cross(fn_a, n_a, fn_b, rounds=ROUNDS):
    """fn_a n_a times per round on stream A, fn_b once on stream B, both queued before either is waited for; the order of the two
    queues alternates so that the kernels meet in every relative phase.  Returns the number of launches compared."""
    torch.cuda.synchronize()
    ref_a, ref_b = _tup(fn_a()), _tup(fn_b())
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    n = 0
    for it in range(rounds):
        outs_a, out_b = [], None
        for who in ((0, 1) if it % 2 == 0 else (1, 0)):
            if who == 0:
                with torch.cuda.stream(sa):
                    for _ in range(n_a):
                        outs_a.append(_tup(fn_a()))
            else:
                with torch.cuda.stream(sb):
                    out_b = _tup(fn_b())
        torch.cuda.synchronize()
        for o in outs_a:
            for a, r in zip(o, ref_a):
                assert torch.equal(a, r), ("stream A", it)
            n += 1
        for a, r in zip(out_b, ref_b):
            assert torch.equal(a, r), ("stream B", it)
        n += 1
        del outs_a, out_b
    return n


def _grad(f, v):
    vg = v.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        f(vg).sum().backward()
    return vg.grad


def test_packed_stft_against_the_mel_cepstral_forward(setup):
    x, stft, mcep, S = setup
    with torch.no_grad():
        assert _cross(lambda: stft(x), 6, lambda: mcep(S)) == ROUNDS * 7
        fused = dsp.fuse(stft, mcep)
        assert _cross(lambda: stft(x), 6, lambda: fused(x)) == ROUNDS * 7
        assert fused.last_path == "fused"


def test_packed_stft_forward_and_backward_against_the_two_wave_mel_cepstral_backward(setup):
    x, stft, mcep, S = setup
    # stream A: packed forward + packed backward (gradient w.r.t. the waveform); stream B: mcep forward with history + bwd2
    _cross(lambda: (stft(x[:512]).detach(), _grad(stft, x[:512])), 4, lambda: _grad(mcep, S))


def test_packed_stft_against_the_one_launch_mgcep_step_and_the_48khz_solver(setup):
    x, stft, mcep, S = setup
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=10, device=DEV)
    Sq = S[:256].contiguous()
    with torch.no_grad():
        _cross(lambda: stft(x), 8, lambda: mg(Sq))
        g = torch.Generator().manual_seed(12)
        x48 = torch.randn(64, 48000, generator=g).to(DEV)
        st48 = dsp.STFT(1200, 240, 2048, device=DEV)
        m48 = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=49, alpha=0.55, n_iter=10, device=DEV)
        X48 = st48(x48)
        _cross(lambda: stft(x), 8, lambda: m48(X48))
        # ... and the reverse pairing: the low-register kernels (filter bank epilogue, LPC) on B next to the packed STFT on A
        fb = dsp.fuse(stft, dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, device=DEV))
        fl = dsp.fuse(dsp.Frame(400, 80), dsp.Window(400, device=DEV), dsp.LPC(400, 24, eps=1e-5, device=DEV))
        _cross(lambda: stft(x), 2, lambda: (fb(x), fl(x)), rounds=25)


def test_overlapped_launches_flag_changes_the_tile_dealing_only(setup):
    """DSA_ALGO_OVERLAPPED_LAUNCHES (ops.overlapped_launches, bench.py --streams 2): the short last round of a launch is packed onto a
    few workgroups so that the next launch -- on the caller's other stream -- takes the freed CUs.  Which wave computes a tile never
    changes the tile's arithmetic: alone, and alternating on two streams, every launch is bit-identical to the plain launch; batches
    with a short round below and above half the wave slots, one that is a whole number of rounds, and one below one round."""
    from diffsptk_amd import ops

    x, stft, mcep, S = setup
    fused = dsp.fuse(stft, mcep)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.no_grad():
        for B in (1024, 900, 512 + 128 + 16, 100):   # 12 800 / 11 250 / 8 200 / 1 250 tiles on 2 048 slots
            xb = x[:B]
            ref, ref2 = fused(xb), mcep(S[:B])
            with ops.overlapped_launches():
                assert torch.equal(fused(xb), ref) and torch.equal(mcep(S[:B]), ref2)
                torch.cuda.synchronize()
                outs = []
                for it in range(30):
                    with torch.cuda.stream(sa if it % 2 == 0 else sb):
                        outs.append(fused(xb) if it % 3 else mcep(S[:B]))
                torch.cuda.synchronize()
                for it, o in enumerate(outs):
                    assert torch.equal(o, ref if it % 3 else ref2), (B, it)
            assert torch.equal(fused(xb), ref)   # the switch is off again outside the context


def test_chunked_overlap_on_alternating_streams_at_full_load(setup, monkeypatch):
    """dist.analyze_chunked_overlap(alternate_streams=True) itself at a size that fills the chip (the older test of that path runs
    600 frames): chunk c + 1's packed STFT on a side stream while chunk c's mel-cepstral kernel is in its Newton phase.  One GPU
    here, so the collective is a stand-in that copies the chunk into both ranks' slots on the stream current at the call."""
    import torch.distributed as tdist

    from diffsptk_amd import dist as ddist

    class _Work:
        def wait(self):
            return True

    def fake_all_gather(out, src, group=None, async_op=False):
        for r in range(out.size(0) // src.size(0)):
            out[r * src.size(0):(r + 1) * src.size(0)].copy_(src)
        return _Work()

    monkeypatch.setattr(tdist, "is_initialized", lambda: True)
    monkeypatch.setattr(tdist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(tdist, "all_gather_into_tensor", fake_all_gather)
    x, stft, mcep, S = setup
    with torch.no_grad():
        ref = mcep(S)
        for it in range(12):
            y = ddist.analyze_chunked_overlap(x, lambda w: mcep(stft(w)), 2 + it % 3)
            torch.cuda.synchronize()
            assert torch.equal(y[:1024], ref) and torch.equal(y[1024:], ref), it
