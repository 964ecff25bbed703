"""GPU parity tests at the sizes BASELINE.json names (run with -m gpu on an MI355X).

One test per BASELINE config that is not already run at full size in tests/test_gpu_parity.py:

  configs[2]  STFT + mcep fwd+bwd, batch 256: the BACKWARD at batch 256 against autograd through the
              float64 ATen port of the reference (oracle/torch_port.py) on sampled utterances -- frames are
              independent, so d mean(mc) / d x_b only involves utterance b and is checkable utterance by
              utterance;
  configs[4]  the per-GPU shard of the 8192-utterance batch (1024 utterances x 1 s): sampled utterances against
              the C oracle, permutation invariance bit-for-bit, Newton fixed point; plus the 8-way split itself
              (what each rank of the 8-GPU run computes) emulated shard by shard on this one GPU.

Tolerances: float32 mel-cepstra |mc - mc64| <= 1e-4 |mc64| + 5e-6 (the reference's own float32 run is 6e-6 from
its float64 run; its test criterion is rtol 1e-4 / atol 1e-6 for a float32 op against SPTK's float32 output,
tests/utils.py:66-72); gradients 3e-6 of the largest entry of the utterance's gradient (3 x the error measured by
tools/measure_tolerances.py).
"""
import os

import numpy as np
import pytest
import torch

import diffsptk_amd as dsp
from diffsptk_amd import _lib, functional as F, ops
from oracle import oracle as O
from oracle import torch_port as TP

pytestmark = pytest.mark.gpu
DEV = "cuda"
MC32 = dict(rtol=1e-4, atol=5e-6)


def host(t):
    return t.detach().cpu().numpy()


def _modules():
    stft = dsp.STFT(400, 80, 512, device=DEV)
    mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=DEV)
    return stft, mcep


def test_config3_backward_batch256_vs_float64_autograd():
    """BASELINE configs[2]: fwd+bwd at batch 256, gradient of mean(mcep(stft(x))) wrt x, on the tuned kernels."""
    B = 256
    x = torch.randn(B, 16000, generator=torch.Generator().manual_seed(3))
    xd = x.to(DEV).requires_grad_(True)
    stft, mcep = _modules()
    mc = mcep(stft(xd))
    assert _lib.last_kernel().startswith("mcep_mfma_fwd")
    mc.mean().backward()
    g = host(xd.grad)
    assert g.shape == (B, 16000) and np.isfinite(g).all()
    sel = [0, 37, 101, 128, 200, 255]
    tab = TP.McepTables(512, 24, 0.42, torch.float64)
    xs = x[sel].double().requires_grad_(True)
    mcs = TP.stft_mcep(xs, tab)
    (mcs.sum() / mc.numel()).backward()           # the same functional: mean over ALL B x N x 25 outputs
    np.testing.assert_allclose(host(mc)[sel], mcs.detach().numpy(), **MC32)
    ref = xs.grad.numpy()
    for i, b in enumerate(sel):
        err = np.abs(g[b] - ref[i]).max()
        # measured 3.2e-7 ... 8.7e-7 of the utterance's largest entry (tools/measure_tolerances.py): bound = 3 x the worst
        assert err < 3e-6 * np.abs(ref[i]).max(), (b, err, np.abs(ref[i]).max())
    # utterances are independent in the backward too: permuting the batch permutes the gradient, bit for bit
    idx = torch.randperm(B, generator=torch.Generator().manual_seed(4)).to(DEV)
    xp = xd.detach()[idx].clone().requires_grad_(True)
    mcep(stft(xp)).mean().backward()
    assert torch.equal(xp.grad, xd.grad[idx])


def test_config5_shard_1024_utterances():
    """BASELINE configs[4], one rank's shard: 1024 utterances x 1 s through STFT -> mcep."""
    B = 1024
    x = torch.randn(B, 16000, generator=torch.Generator().manual_seed(11))
    xd = x.to(DEV)
    stft, mcep = _modules()
    X = stft(xd)
    assert _lib.last_kernel() == "stft512_fwd"
    mc = mcep(X)
    assert _lib.last_kernel().startswith("mcep_mfma_fwd")
    assert mc.shape == (B, 200, 25) and bool(torch.isfinite(mc).all())
    sel = [0, 171, 342, 513, 684, 855, 1023]
    x64 = x[sel].double().numpy()
    X_ref = O.stft(x64, 400, 80, 512)
    err = np.abs(host(X)[sel] - X_ref) / X_ref.max(-1, keepdims=True)
    assert err.max() < 2e-6, err.max()
    np.testing.assert_allclose(host(mc)[sel], O.mcep(X_ref, 24, 0.42, 10), **MC32)
    idx = torch.randperm(B, generator=torch.Generator().manual_seed(12)).to(DEV)
    assert torch.equal(mcep(stft(xd[idx])), mc[idx])          # frames are independent: bitwise
    mc11 = F.mcep(X[:16], 24, 0.42, 11)                       # converged: one more Newton step moves < 1e-4
    assert float((mc11 - mc[:16]).abs().max()) < 1e-4
    # Parseval per frame: sum_k c_k |X_k|^2 = nfft * sum_l (w_l x_l)^2  (the window has unit power)
    fr = dsp.Window(400, device=DEV)(dsp.Frame(400, 80)(xd[:32]))
    lhs = (2 * X[:32].sum(-1) - X[:32, :, 0] - X[:32, :, -1] - 2 * 255 * 1e-9)
    rhs = 512 * fr.square().sum(-1)
    assert float(((lhs - rhs).abs() / rhs).max()) < 1e-4


def test_config5_whole_batch_8192_on_one_gpu():
    """BASELINE configs[4] AS WRITTEN at N = 1 (bench.py --global-batch 8192): all 8192 utterances x 1 s on one GPU through the
    sharding entry point (dist.analyze_chunked_overlap, world of one), two kernels and one launch; sampled utterances against the
    C oracle, and every 1024-utterance shard -- what a rank of the 8-GPU run computes -- equal to its rows bit for bit."""
    from diffsptk_amd.dist import analyze_chunked_overlap, shard_bounds

    B = 8192
    x = torch.randn(B, 16000, generator=torch.Generator().manual_seed(8192))
    xd = x.to(DEV)
    stft, mcep = _modules()
    fused = dsp.fuse(stft, mcep)
    with torch.no_grad():
        mc = analyze_chunked_overlap(xd, lambda xc: mcep(stft(xc)), 1)
        assert mc.shape == (B, 200, 25) and bool(torch.isfinite(mc).all())
        mc1 = analyze_chunked_overlap(xd, fused, 1)
        assert fused.last_path == "fused"
        assert float((mc1 - mc).abs().max()) <= 1e-6 * float(mc.abs().max())
        for r in (0, 3, 7):
            lo, hi = shard_bounds(B, 8, r)
            assert torch.equal(mcep(stft(xd[lo:hi])), mc[lo:hi])
    sel = [0, 1023, 1024, 4095, 6000, 8191]
    X_ref = O.stft(x[sel].double().numpy(), 400, 80, 512)
    ref = O.mcep(X_ref, 24, 0.42, 10)
    np.testing.assert_allclose(host(mc[sel]), ref, **MC32)
    np.testing.assert_allclose(host(mc1[sel]), ref, **MC32)


def test_speech_like_batch_at_bench_size():
    """"also run a speech-like input (data.wav tiled) because conditioning and Newton convergence differ".  The batch
    bench.py times as `speech_like` (bench.speech_like_batch: the reference's test recording repeated end to end, utterance u
    starting 997 u samples in -- 1 024 different alignments, frames of near-silence, onsets and voiced stretches at every tile
    position) through the one-launch step and the two kernels at the BENCH size, against the C oracle in float64 on every eighth
    utterance (25 600 frames), with the float32 tolerance of the goldens; Newton has converged (an eleventh step moves < 1e-4)."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        import bench
    finally:
        sys.path.pop(0)
    B = 1024
    xd = bench.speech_like_batch(B, torch.device(DEV))
    assert xd.shape == (B, 16000) and float(xd.abs().max()) <= 1.0
    stft, mcep = _modules()
    fused = dsp.fuse(stft, mcep)
    with torch.no_grad():
        X = stft(xd)
        mc2 = mcep(X)
        mc1 = fused(xd)
        assert fused.last_path == "fused" and _lib.last_kernel() == "stft512_mcep_fused_fwd"
        assert bool(torch.isfinite(mc1).all()) and bool(torch.isfinite(mc2).all())
        assert float((mc1 - mc2).abs().max()) <= 1e-6 * float(mc2.abs().max())
        mc11 = F.mcep(X[:64], 24, 0.42, 11)
        assert float((mc11 - mc2[:64]).abs().max()) < 1e-4
    sel = list(range(0, B, 8))
    x64 = xd[sel].double().cpu().numpy()
    X_ref = O.stft(x64, 400, 80, 512)
    err = np.abs(host(X)[sel] - X_ref) / X_ref.max(-1, keepdims=True)
    assert err.max() < 2e-6, err.max()
    ref = O.mcep(X_ref, 24, 0.42, 10)
    np.testing.assert_allclose(host(mc1)[sel], ref, **MC32)
    np.testing.assert_allclose(host(mc2)[sel], ref, **MC32)
    # the dynamic range this input brings (what white noise does not): frame energies spread over > 50 dB
    e = X.sum(-1)
    assert float(10 * torch.log10(e.max() / e.min())) > 50


def test_config5_eight_way_split_equals_whole_batch():
    """The 8-GPU run shards the batch contiguously (diffsptk_amd/dist.py:shard_bounds) and all-gathers the
    features; every rank's shard computed on its own must reproduce the rows of the whole-batch result bit for
    bit (what the all-gather then concatenates).  8192 utterances would take 0.5 GB of waveform; the split is
    exercised on 8 x 96 utterances, ragged last shard included."""
    from diffsptk_amd.dist import shard_bounds

    stft, mcep = _modules()
    for B in (768, 761):
        x = torch.randn(B, 16000, generator=torch.Generator().manual_seed(B)).to(DEV)
        whole = mcep(stft(x))
        parts = []
        for r in range(8):
            lo, hi = shard_bounds(B, 8, r)
            parts.append(mcep(stft(x[lo:hi])))
        assert torch.equal(torch.cat(parts), whole)


def test_rccl_path_world1():
    """The RCCL ("nccl" backend) code path of diffsptk_amd.dist on this one GPU: a world-size-1 process group with
    the collectives forced on (tools/nccl_smoke.py) -- in-place all_gather_into_tensor per chunk on alternating
    streams, both layouts, deferred completion.  Two ranks cannot share one device under RCCL, so N > 1 itself is
    covered by the gloo world-size-2 tests (tests/test_dist_cpu.py) and the driver's 8-GPU run."""
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_smoke.py")], env=env, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0 and "nccl smoke OK" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def test_mcep_many_launches_two_tables_four_streams():
    """The library owns no device memory (include/diffsptk_amd.h, Conventions): operand images are per-configuration
    constants prepared by the caller's side once, the tile-queue counters live in a per-call scratch.  More
    launches in flight than any fixed pool could hold -- 4 streams x 40 launches, two different alpha tables
    interleaved, forward and backward -- must reproduce the single-stream results bit for bit."""
    from diffsptk_amd import ops

    stft = dsp.STFT(400, 80, 512, device=DEV)
    mA = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=DEV)
    mB = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.55, n_iter=10, device=DEV)
    X = stft(torch.randn(48, 16000, generator=torch.Generator().manual_seed(77)).to(DEV))
    wts = torch.linspace(-1, 1, 25, device=DEV)

    def fwd_bwd(m, Xin):
        Xg = Xin.clone().requires_grad_(True)
        mc = m(Xg)
        (mc * wts).sum().backward()
        return mc.detach(), Xg.grad

    refA, refB = fwd_bwd(mA, X), fwd_bwd(mB, X)
    assert not torch.equal(refA[0], refB[0])
    imgA = ops.mcep_images(mA.G, mA.D, mA.E, 512, 24)
    assert imgA is not None and imgA is ops.mcep_images(mA.G, mA.D, mA.E, 512, 24)   # prepared once, then cached
    assert imgA.data_ptr() != ops.mcep_images(mB.G, mB.D, mB.E, 512, 24).data_ptr()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = []
    for i in range(160):
        s = streams[i % 4]
        with torch.cuda.stream(s):
            m = mA if (i // 4) % 2 == 0 else mB
            outs.append((i, fwd_bwd(m, X)))
    torch.cuda.synchronize()
    for i, (mc, g) in outs:
        ref = refA if (i // 4) % 2 == 0 else refB
        assert torch.equal(mc, ref[0]) and torch.equal(g, ref[1]), i
    # the explicit error instead of a silent fallback: TUNED without the workspaces
    with pytest.raises(_lib.BackendError):
        mc = torch.empty(4, 25, device=DEV)
        ops._call("dsa_mcep_fwd", ops._p(X[0, :4].contiguous()), 4, 512, 24, 10, ops._p(mA.G), ops._p(mA.D), ops._p(mA.E),
                  ops._p(mA.alpha_vector), _lib.F32, _lib.ALGO_TUNED, None, None, ops._p(mc), None, ops._stream())


def test_learnable_options_match_the_kernels_and_train():
    """``learnable=`` (stft.py:73-76, fftr.py:123-129, fbank.py:112-122, istft.py, unframe.py): the tables become
    Parameters and the affected stage runs on stock device operators (modules/_learnable.py).  At initialisation the
    result must equal the kernels' (same transform), and every Parameter must receive a finite, non-zero gradient."""
    x = torch.randn(3, 1600, generator=torch.Generator().manual_seed(5)).to(DEV)
    fixed = dsp.STFT(400, 80, 512, device=DEV)
    for learn in (True, ["basis"], ["window"]):
        m = dsp.STFT(400, 80, 512, learnable=learn, device=DEV)
        y = m(x)
        ref = fixed(x)
        err = (y.detach() - ref).abs() / ref.amax(-1, keepdim=True)
        assert float(err.max()) < 5e-6, (learn, float(err.max()))
        torch.log(y).sum().backward()
        for n, p in m.named_parameters():
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0, (learn, n)
    mc = dsp.STFT(400, 80, 512, out_format="complex", learnable=["basis"], device=DEV)
    zc = dsp.STFT(400, 80, 512, out_format="complex", device=DEV)(x)
    assert float((mc(x) - zc).abs().max()) < 1e-4 * float(zc.abs().max())
    # inverse path: learnable synthesis basis + window reproduce the waveform of the fused kernel
    inv = dsp.ISTFT(400, 80, 512, learnable=True, device=DEV)
    xr = inv(zc, out_length=1600)
    ref = dsp.ISTFT(400, 80, 512, device=DEV)(zc, out_length=1600)
    assert float((xr - ref).abs().max()) < 2e-5
    xr.square().sum().backward()
    assert all(p.grad is not None and float(p.grad.abs().max()) > 0 for p in inv.parameters())
    # filter bank / MFCC with a learnable H
    X = fixed(x)
    fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, learnable=True, device=DEV)
    fb0 = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, device=DEV)
    np.testing.assert_allclose(host(fb(X)), host(fb0(X)), rtol=2e-5, atol=2e-5)
    fb(X).sum().backward()
    assert float(fb.H.grad.abs().max()) > 0
    mf = dsp.MFCC(fft_length=512, mfcc_order=12, n_channel=40, sample_rate=16000, learnable=True, device=DEV)
    mf0 = dsp.MFCC(fft_length=512, mfcc_order=12, n_channel=40, sample_rate=16000, device=DEV)
    np.testing.assert_allclose(host(mf(X)), host(mf0(X)), rtol=1e-4, atol=1e-4)
    assert [n for n, _ in mf.named_parameters()] == ["H"]


@pytest.mark.parametrize("nfft,L,P", [(32, 32, 8), (64, 50, 16), (256, 200, 64), (1024, 1024, 256), (4096, 2000, 2000), (4096, 4096, 1024)])
def test_generic_rows_power_of_two_fft_matches_oracle_and_autograd(nfft, L, P):
    """The generic row transforms (fftr / spec / STFT of any size, their backward and the inverse path) run a radix-2 FFT
    in LDS for power-of-two lengths instead of the direct sum: forward against the float64 oracle, gradient against
    float64 autograd of the same formula; float64 and float32, zero-padded and cropped rows, zmean + reflect padding."""

    g = torch.Generator().manual_seed(nfft + L)
    x = torch.randn(3, 4 * P + L, generator=g, dtype=torch.float64)
    for dt, rt in ((torch.float64, 1e-9), (torch.float32, 2e-4)):
        xd = x.to(DEV, dt).requires_grad_(True)
        for fmt in ("complex", "power"):
            stft = dsp.STFT(L, P, nfft, out_format=fmt, zmean=(fmt == "power"), mode="reflect" if fmt == "power" else "constant",
                            eps=0.0, dtype=dt, device=DEV)
            y = stft(xd)
            if not (dt == torch.float32 and nfft == 512):
                assert _lib.last_kernel() == "row_fft_generic"
            ref = O.stft(x.numpy(), L, P, nfft, out_format=fmt, zmean=(fmt == "power"), mode="reflect" if fmt == "power" else "constant", eps=0.0)
            yh = y.detach().cpu().numpy()
            scale = np.abs(ref).max()
            assert np.abs(yh - ref).max() <= rt * scale, (nfft, fmt, dt, np.abs(yh - ref).max() / scale)
            # gradient of a fixed random functional against float64 autograd of the same formula
            wgt = torch.randn(y.shape, generator=g, dtype=torch.float64)
            if fmt == "complex":
                wgt = torch.complex(wgt, torch.randn(y.shape, generator=g, dtype=torch.float64))
                loss = (y * wgt.to(DEV).to(y.dtype)).real.sum()
            else:
                loss = (y * wgt.to(DEV, dt)).sum()
            (gx,) = torch.autograd.grad(loss, xd)
            xr = x.clone().requires_grad_(True)
            fr = TPframes(xr, L, P, fmt == "power", "reflect" if fmt == "power" else "constant")
            win = stft.window.detach().cpu().double()
            Y = torch.fft.rfft(fr * win, n=nfft)
            lr = (Y * wgt).real.sum() if fmt == "complex" else ((Y.abs() ** 2) * wgt).sum()
            (gr,) = torch.autograd.grad(lr, xr)
            gs = gr.abs().max().item()
            assert (gx.detach().cpu().double() - gr).abs().max().item() <= (1e-9 if dt == torch.float64 else 3e-4) * gs


def TPframes(x, L, P, zmean, mode):
    """Framing of frame.py:130-138 with stock torch ops (test helper)."""
    left, right = L // 2, (L - 1) // 2
    xp = torch.nn.functional.pad(x.unsqueeze(0), (left, right), mode=mode).squeeze(0)
    fr = xp.unfold(-1, L, P)
    return fr - fr.mean(-1, keepdim=True) if zmean else fr


@pytest.mark.parametrize("K,N,Fr", [(257, 49, 4096), (257, 25, 2001), (300, 100, 1500), (129, 3, 1024), (320, 192, 1100)])
def test_long_row_products_on_matrix_cores(K, N, Fr):
    """c @ A for long float32 rows (the 257-bin spectra of mgcep against its composed matrices): the float32 matrix-core
    row product, 48 output columns per launch into a column slice of the result; against float64, with the backward
    (vector kernels) through autograd, ragged row counts and slice widths."""
    gen = torch.Generator().manual_seed(K + N)
    c = torch.randn(Fr, K, dtype=torch.float64, generator=gen)
    A = torch.randn(K, N, dtype=torch.float64, generator=gen)
    A[K // 3: K // 3 + 70] = 0.0          # empty blocks are skipped by the kernel's program
    g = torch.randn(Fr, N, dtype=torch.float64, generator=gen)
    cd = c.to(DEV, torch.float32).requires_grad_(True)
    out = ops.MatmulRowsFn.apply(cd, A.to(DEV, torch.float32))
    assert _lib.last_kernel() == "freqt_mfma_fwd"
    out.backward(g.to(DEV, torch.float32))
    ref, gref = (c @ A).numpy(), (g @ A.T).numpy()
    assert np.abs(out.detach().cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    assert np.abs(cd.grad.cpu().numpy() - gref).max() <= 2e-5 * np.abs(gref).max()
    ops.MatmulRowsFn.apply(c[:2].to(DEV), A.to(DEV))          # (a tiny float64 call: one-workgroup-per-row kernel)
    assert _lib.last_kernel() == "freqt_fwd"
    out64 = ops.MatmulRowsFn.apply(c.to(DEV), A.to(DEV))      # float64: the LDS-resident vector kernel, or the vendor GEMM when the
    assert _lib.last_kernel() != "freqt_mfma_fwd"             # operands do not fit LDS -- never the float32 matrix-core kernel
    assert np.abs(out64.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-10), (torch.float32, 2e-4)])
def test_mgcep_fused_spectrum_arithmetic_equals_the_differentiable_chain(dt, tol):
    """MelGeneralizedCepstralAnalysis (gamma != 0) without a gradient runs the spectrum arithmetic of a Newton step as one
    kernel (dsa_mgcep_spectra) and the five long row products on matrix cores (float32); with a gradient it composes the
    same step from differentiable operators.  Same result, and against the float64 oracle."""
    x = torch.randn(8, 4000, generator=torch.Generator().manual_seed(11), dtype=torch.float64)
    X = dsp.STFT(400, 80, 512, dtype=dt, device=DEV)(x.to(DEV, dt))
    for gamma, M in ((-0.5, 24), (-0.25, 30), (-1 / 3, 12)):
        mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=M, alpha=0.42, gamma=gamma, n_iter=4, dtype=dt, device=DEV)
        a = mg(X)
        assert _lib.last_kernel() != "" and not a.requires_grad
        Xg = X.clone().requires_grad_(True)
        b = mg(Xg)
        assert b.requires_grad
        scale = float(b.abs().max())
        assert float((a - b.detach()).abs().max()) <= tol * scale
        ref = O.mgcep(X.double().cpu().numpy(), M, 0.42, gamma, 4)
        assert np.abs(a.double().cpu().numpy() - ref).max() <= (1e-8 if dt == torch.float64 else 3e-6) * np.abs(ref).max()   # float32 measured 8.4e-7


@pytest.mark.parametrize("F", [1, 15, 16, 33, 1595, 2049 * 16 + 5])
def test_mgcep_whole_step_in_one_launch_against_the_two_launch_step_and_the_oracle(F, monkeypatch):
    """dsa_mgcep_step_solve (round 5: mgcep.py:199-230 in ONE launch at cep_order 24 / float32 / fft_length 512 -- the matrix chains as
    3-term binary16 splits with per-stage power-of-two scales, the 24 x 24 block elimination and the update behind them) against
    dsa_mgcep_step + dsa_thsolve_update_fwd (float32 matrix instructions) on the same inputs: r within 2e-6 of its maximum, the updated
    coefficients within 1e-5 (measured 6e-7 / 2.4e-6 by tools/check_mgcep_step_solve.py); ragged frame counts around the 16-frame tiles and
    past one round of the 2 048 wave slots; repeated launches bit-identical; in place (b1_out = b1); the analysis through the module
    against the float64 oracle at 3e-6 (measured 4e-7; the two-launch step 8e-7)."""
    gen = torch.Generator().manual_seed(F)
    X = (torch.randn(F, 257, generator=gen).square() * torch.rand(F, 1, generator=gen) + 1e-3).to(DEV)
    b1 = (0.05 * torch.randn(F, 24, generator=gen)).to(DEV)
    for gamma in (-0.5, -1 / 3):
        mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=gamma, n_iter=3, device=DEV)
        pt, qt, r = ops.mgcep_step(X, b1, mg.step_images, gamma)
        ref_b = ops.thsolve_update(pt, qt, r, b1)
        new_b, new_r = ops.mgcep_step_solve(X, b1, mg.step_images_h, gamma)
        assert _lib.last_kernel() == "mgcep_step_solve"
        assert torch.isfinite(new_b).all() and torch.isfinite(new_r).all()
        assert float((new_r - r).abs().max()) <= 2e-6 * float(r.abs().max())
        assert float((new_b - ref_b).abs().max()) <= 1e-5 * max(1.0, float(ref_b.abs().max()))
        again_b, again_r = ops.mgcep_step_solve(X, b1, mg.step_images_h, gamma)
        assert torch.equal(again_b, new_b) and torch.equal(again_r, new_r)
        bi = b1.clone()
        ops.mgcep_step_solve(X, bi, mg.step_images_h, gamma, out=bi)
        assert torch.equal(bi, new_b)
        # three steps in ONE launch (the coefficients stay in LDS between them) = three launches, bit for bit; r and the last step's input too
        b3, r3, p3 = ops.mgcep_step_solve(X, b1, mg.step_images_h, gamma, n_steps=3, want_prev=True)
        s1, _ = ops.mgcep_step_solve(X, new_b, mg.step_images_h, gamma)
        s2, rr2 = ops.mgcep_step_solve(X, s1, mg.step_images_h, gamma)
        assert torch.equal(b3, s2) and torch.equal(r3, rr2) and torch.equal(p3, s1)
        if F >= 33:   # a frame's result does not depend on the launch it is part of
            part_b, part_r = ops.mgcep_step_solve(X[:33], b1[:33], mg.step_images_h, gamma)
            assert torch.equal(part_b[:32], new_b[:32]) and torch.equal(part_r[:32], new_r[:32])
        with torch.no_grad():
            y1 = mg(X)
            monkeypatch.setenv("DSA_MGCEP_STEP_SOLVE", "0")
            y0 = mg(X)
            monkeypatch.delenv("DSA_MGCEP_STEP_SOLVE")
        assert not torch.equal(y0, y1) or F < 16
        if F <= 2048:
            ref = O.mgcep(X.double().cpu().numpy(), 24, 0.42, gamma, 3)
            for y in (y1, y0):
                assert np.abs(y.double().cpu().numpy() - ref).max() <= 3e-6 * np.abs(ref).max()
        else:
            assert float((y1 - y0).abs().max()) <= 5e-6 * float(y0.abs().max())
    # with a graph: the same launch forward (it keeps pt, qt), the adjoint solve + the step's adjoint backward -- against the graph
    # autograd composes from the two-launch step (MgcepStepFn, ThSolveFn and the additions around them)
    if F <= 2048:
        w = torch.randn(24, generator=gen).to(DEV)
        wr = torch.randn(25, generator=gen).to(DEV)
        xa, ba = X.clone().requires_grad_(True), b1.clone().requires_grad_(True)
        o1, r1 = ops.MgcepStepSolveFn.apply(xa, ba, mg.step_images_h, mg.step_images_bwd, gamma)
        ((o1 * w).sum() + (r1 * wr).sum()).backward()
        xb, bb = X.clone().requires_grad_(True), b1.clone().requires_grad_(True)
        pt2, qt2, r2 = ops.MgcepStepFn.apply(xb, bb, mg.step_images, mg.step_images_bwd, gamma)
        o2 = bb + ops.ThSolveFn.apply(pt2, qt2, r2[..., 1:])
        ((o2 * w).sum() + (r2 * wr).sum()).backward()
        assert float((xa.grad - xb.grad).abs().max()) <= 2e-5 * float(xb.grad.abs().max())
        assert float((ba.grad - bb.grad).abs().max()) <= 2e-5 * float(bb.grad.abs().max())
    # unsupported set-ups are refused, not approximated
    with pytest.raises(Exception):
        ops.mgcep_step_solve(X[:, :129].contiguous(), b1, mg.step_images_h, -0.5)
    with pytest.raises(Exception):
        ops.mgcep_step_solve(X, b1[:, :12].contiguous(), mg.step_images_h, -0.5)


@pytest.mark.parametrize("F", [1, 17, 2050])
def test_mgcep_step_adjoint_on_binary16_splits_against_the_float32_kernel(F):
    """dsa_mgcep_step_bwd_h (round 5: the step's adjoint with its 66 products per 32 bins as 3-term binary16 splits; cotangent vectors
    scaled per frame, the cotangent of (re, im) per stage) against dsa_mgcep_step_bwd (float32 matrix instructions) on the same inputs:
    3e-6 of the largest entry of each output (the two differ by both kernels' rounding; against float64 through the module the new one
    measures 3.4e-7, the old 6.5e-7: tools/time_mgcep_grad.py); ragged frame counts; repeated launches bit-identical; cotangents of
    very different levels per frame (the per-frame scales)."""
    gen = torch.Generator().manual_seed(100 + F)
    X = (torch.randn(F, 257, generator=gen).square() + 0.05).to(DEV)
    b1 = (0.05 * torch.randn(F, 24, generator=gen)).to(DEV)
    lvl = 2.0 ** torch.randint(-20, 20, (F, 1), generator=gen).float()
    gpt = (torch.randn(F, 24, generator=gen) * lvl).to(DEV)
    gqt = (torch.randn(F, 47, generator=gen) * lvl * 8).to(DEV)
    gr = (torch.randn(F, 25, generator=gen) * lvl / 8).to(DEV)
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=1, device=DEV)
    assert mg.step_images_bwd.dtype == torch.int16   # binary16 bit patterns: Module.float() must not cast them
    img32 = torch.from_numpy(__import__("diffsptk_amd").utils.tables.mgcep_step_bwd_images(512, 24, 0.42)).to(DEV)

    def run(images):
        gx, gb = torch.empty_like(X), torch.empty_like(b1)
        ops._call(ops._step_bwd_entry(images), X.data_ptr(), b1.data_ptr(), gpt.data_ptr(), gqt.data_ptr(), gr.data_ptr(), F, 512, 24, -0.5,
                  images.data_ptr(), _lib.F32, None, gx.data_ptr(), gb.data_ptr(), ops._stream())
        return gx, gb

    gx_h, gb_h = run(mg.step_images_bwd)
    assert _lib.last_kernel() == "mgcep_step_bwd_h"
    gx_f, gb_f = run(img32)
    assert torch.isfinite(gx_h).all() and torch.isfinite(gb_h).all()
    for a, b in ((gx_h, gx_f), (gb_h, gb_f)):
        per = (a - b).abs().amax(1) / b.abs().amax(1).clamp_min(1e-30)
        assert float(per.max()) <= 3e-6, float(per.max())
    gx2, gb2 = run(mg.step_images_bwd)
    assert torch.equal(gx2, gx_h) and torch.equal(gb2, gb_h)


def test_mgcep_step_backward_kernel_against_float64_autograd():
    """With a gradient wanted the float32 / 512 / order <= 24 analysis runs the fused step forward AND its adjoint as one
    launch each (ops.MgcepStepFn: dsa_mgcep_step / dsa_mgcep_step_bwd), the order-24 solve's backward on the quad-layout
    solve + the diagonal sums.  Gradient with respect to the spectrum against autograd through the float64 module (the
    differentiable operator chain, itself pinned to the reference's exported gradients in tests/test_gpu_synth.py);
    measured 8e-7 .. 1e-6 of the largest entry (tools/time_mgcep_grad.py)."""
    g = torch.Generator().manual_seed(21)
    X = (torch.randn(3, 171, 257, generator=g).square() + 0.1)
    for gamma, M, n_iter in ((-0.5, 24, 3), (-0.25, 17, 2), (-1.0 / 3, 24, 10)):
        w = torch.randn(M + 1, generator=g)
        m32 = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=M, alpha=0.42, gamma=gamma, n_iter=n_iter, device=DEV)
        m64 = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=M, alpha=0.42, gamma=gamma, n_iter=n_iter, device=DEV,
                                                 dtype=torch.float64)
        xs = X.to(DEV).requires_grad_(True)
        y = m32(xs)
        (y * w.to(DEV)).sum().backward()
        xd = X.double().to(DEV).requires_grad_(True)
        yd = m64(xd)
        (yd * w.double().to(DEV)).sum().backward()
        assert float((y.double() - yd).abs().max()) < 3e-6 * float(yd.abs().max())
        err = float((xs.grad.double() - xd.grad).abs().max() / xd.grad.abs().max())
        assert err < 3e-6, (gamma, M, n_iter, err)
    # the solve's backward alone, order 24: quad-layout solve + diagonal sums against the one-wave-per-system kernel's formulas
    # in float64
    n, Fr = 24, 1000 + 3
    ii = torch.arange(n)
    wgt = torch.exp(torch.randn(Fr, 48, generator=g, dtype=torch.float64))
    om = torch.pi * (torch.arange(48, dtype=torch.float64) + 0.5) / 48
    p = (wgt[:, None, :] * torch.cos(om[None, None, :] * ii[None, :, None])).sum(-1)
    q = 0.5 * (wgt[:, None, :] * torch.cos(om[None, None, :] * torch.arange(2 * n - 1)[None, :, None])).sum(-1)
    r = torch.randn(Fr, n, generator=g, dtype=torch.float64)
    gg = torch.randn(Fr, n, generator=g, dtype=torch.float64)
    outs = {}
    for dt in (torch.float32, torch.float64):
        pp, qq, rr = (t.to(DEV, dt).requires_grad_(True) for t in (p, q, r))
        (ops.ThSolveFn.apply(pp, qq, rr) * gg.to(DEV, dt)).sum().backward()
        outs[dt] = (pp.grad.double(), qq.grad.double(), rr.grad.double())
    for a, b in zip(outs[torch.float32], outs[torch.float64]):
        assert float((a - b).abs().max()) < 2e-3 * float(b.abs().max())   # float32 solves of systems with condition ~1e3


def test_mgcep_gradient_survives_module_float_and_to():
    """nn.Module.float() / .to(dtype) cast every floating-point buffer: the binary16 operand images of the step's adjoint are kept
    as int16 bit patterns (like the forward's bytes), so the same kernel reads the same image afterwards (round-5 advisor finding:
    as a float16 buffer the image became float32 and the float32 kernel read it in the wrong layout -- finite, wrong gradients)."""
    g = torch.Generator().manual_seed(33)
    X = (torch.randn(2, 90, 257, generator=g).square() + 0.1).to(DEV)
    w = torch.randn(25, generator=g).to(DEV)
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=3, device=DEV)

    def grad(m):
        xs = X.clone().requires_grad_(True)
        (m(xs) * w).sum().backward()
        return xs.grad

    g0 = grad(mg)
    assert _lib.last_kernel() is not None
    bits = mg.step_images_bwd.clone()
    for cast in (lambda m: m.float(), lambda m: m.to(torch.float32), lambda m: m.to(DEV, torch.float32), lambda m: m.double().float()):
        mg = cast(mg)
        assert mg.step_images_bwd.dtype == torch.int16 and torch.equal(mg.step_images_bwd, bits)
        assert mg.step_images_h.dtype == torch.uint8
        assert torch.equal(grad(mg), g0)
    with pytest.raises(ValueError, match="operand images"):
        ops._step_bwd_entry(bits.double())


def test_bench_two_ranks_on_one_device():
    """bench.py's N > 1 control flow (sharded batch, deferred in-place all-gather, drain inside the timed region, MAX over
    ranks, one JSON line from rank 0) as two processes on ONE device over gloo -- the 8-GPU RCCL run itself is the
    driver's; this catches argument / rendezvous / shape breakage before it."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSA_BENCH_SINGLE_DEVICE="1", DSA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29561", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--batch", "64",
           "--ramp-seconds", "0"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    assert "configs" not in d and "cpu_baseline" not in d   # N = 1 only


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (WORLD_SIZE unset): bench.py starts the ranks itself through
    torch.distributed.run on the loopback address and rank 0's JSON line comes through -- what a driver that runs
    `python bench.py --gpus N` gets.  Two ranks on this box's one device (DSA_BENCH_SINGLE_DEVICE), collectives over gloo."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DSA_BENCH_SINGLE_DEVICE="1", DSA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "32",
           "--ramp-seconds", "0"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["config"]["global_batch"] == 64


@pytest.mark.parametrize("fl,fp,nfft,M,alpha", [(1200, 240, 2048, 49, 0.55), (1024, 256, 1024, 34, 0.55), (512, 128, 512, 30, 0.42)])
def test_untuned_geometries_forward_as_whole_batch_launches(monkeypatch, fl, fp, nfft, M, alpha):
    """The 48 kHz set-ups of the reference's get_alpha table (fft_length 1024 / 2048, orders 34 .. 60) and any other
    (fft_length, order) without a tuned kernel: without a graph the analysis runs as whole-batch launches (two GEMMs + the
    batched Toeplitz-plus-Hankel solve per Newton step) instead of the one-workgroup-per-frame kernel.  Both against the
    float64 oracle on sampled frames, and against each other; with a graph the generic kernel pair runs and its gradient is
    checked by the existing tests."""
    x = torch.randn(8, 16 * nfft, generator=torch.Generator().manual_seed(4)).to(DEV)
    stft = dsp.STFT(fl, fp, nfft, device=DEV)
    mcep = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=alpha, n_iter=10, device=DEV)
    with torch.no_grad():
        X = stft(x)
        assert X.shape[0] * X.shape[1] >= 256
        mc = mcep(X)
        # (round 6: orders 32 .. 54 run all Newton steps in one persistent launch, csrc/mcep_big_f16.h)
        assert _lib.last_kernel() in ("th_solve_fwd", "th_solve_quad_fwd", "th_solve_quadn_fwd", "th_solve_octn_fwd", "mcep_big_newton"), _lib.last_kernel()
        assert (_lib.last_kernel() == "mcep_big_newton") == (32 <= M <= 54)
        monkeypatch.setenv("DSA_MCEP_COMPOSED", "0")
        mc_g = mcep(X)
        assert _lib.last_kernel() == "mcep_generic_fwd"
        monkeypatch.delenv("DSA_MCEP_COMPOSED")
    idx = torch.tensor([0, 1, 17, X.shape[1] - 1, X.shape[1], 3 * X.shape[1] + 5, X.shape[0] * X.shape[1] - 1])
    ref = O.mcep(host(X.reshape(-1, nfft // 2 + 1)[idx]).astype(np.float64), M, alpha=alpha, n_iter=10)
    got = host(mc.reshape(-1, M + 1)[idx]).astype(np.float64)
    np.testing.assert_allclose(got, ref, **MC32)
    np.testing.assert_allclose(host(mc), host(mc_g), rtol=1e-4, atol=1e-5)
    # a graph is wanted: the same whole-batch composition with autograd through it, whatever the batch (the path is chosen from
    # the geometry alone: 8 frames and 300 take the same kernels) -- gradient against the float64 module on the generic pair
    Xg = X[:1, :8].clone().requires_grad_(True)
    y = mcep(Xg)
    assert _lib.last_kernel() != "mcep_generic_fwd"
    with torch.no_grad():                          # ... and a frame's result does not depend on its batch (bit for bit)
        assert torch.equal(mcep(X[:1, :8]), mc[:1, :8]) and torch.equal(mcep(X[:3]), mc[:3])
    y.sum().backward()
    assert bool(torch.isfinite(Xg.grad).all()) and float(Xg.grad.abs().max()) > 0
    w = torch.randn(M + 1, generator=torch.Generator().manual_seed(6)).to(DEV)
    Xs = X.reshape(-1, nfft // 2 + 1)[:300]
    Xc = Xs.clone().requires_grad_(True)
    (mcep(Xc) * w).sum().backward()
    assert _lib.last_kernel() != "mcep_generic_fwd"
    monkeypatch.setenv("DSA_MCEP_COMPOSED", "0")
    m64 = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=alpha, n_iter=10, device=DEV, dtype=torch.float64)
    Xd = Xs.double().clone().requires_grad_(True)
    (m64(Xd) * w.double()).sum().backward()
    monkeypatch.delenv("DSA_MCEP_COMPOSED")
    err = float((Xc.grad.double() - Xd.grad).abs().max() / Xd.grad.abs().max())
    assert err < 2e-4, err


@pytest.mark.parametrize("two_wave", [False, True])
@pytest.mark.parametrize("tail_tiles,tail_frames", [(8, 8 * 16 - 13), (512, 512 * 16), (37, 37 * 16 - 1)])
def test_mcep_backward_split_tail_is_bit_identical_to_whole_tiles(tail_tiles, tail_frames, two_wave, monkeypatch):
    """The one-wave tuned backward (DSA_MCEP_BWD2=0, or a history without rt rows) cuts a short last round of tiles into pieces of
    Newton steps that hand (lbar, mbar) over through memory (csrc/mcep_mfma_bwd_f16.h, DSA_ALGO_SCRATCH_HAS_WORKSPACE).  Frames are
    independent and a piece repeats the whole tile's arithmetic step by step, so the gradient of the tail frames must equal BIT FOR
    BIT what a launch of those frames alone (fewer tiles than wave slots: no split) computes -- for 2 / 9 / 3 pieces, ragged last
    tiles included.  The two-wave kernel (the default) has no split tail; the same statement -- a frame's gradient does not depend on
    the launch it is part of -- is checked for it too."""
    if not two_wave:
        monkeypatch.setenv("DSA_MCEP_BWD2", "0")
    _, mcep = _modules()
    gen = torch.Generator().manual_seed(tail_tiles)
    F = 2 * 1024 * 16 + tail_frames                      # two full rounds of the 1024 wave slots + the tail
    X = (torch.randn(F, 257, generator=gen).square() + 0.05).to(DEV)
    w = torch.randn(25, generator=gen).to(DEV)

    def grad(Xin):
        Xg = Xin.clone().requires_grad_(True)
        (mcep(Xg) * w).sum().backward()
        return Xg.grad

    g_all = grad(X)
    g_tail = grad(X[-tail_frames:])
    assert torch.isfinite(g_all).all()
    assert torch.equal(g_all[-tail_frames:], g_tail)
    # and the whole tiles of the same launch against a launch of their own
    assert torch.equal(g_all[:4096], grad(X[:4096]))


def test_mcep_backward_with_the_saved_rt_rows_equals_the_recomputing_backward(monkeypatch):
    """DSA_ALGO_HIST_HAS_RT (round 5): the tuned forward keeps every step's rt = e E row behind the iterates and the tuned backward
    loads it instead of re-running its second forward chain.  The row IS what the forward's solve used, so the gradient may differ
    from the recomputing backward only by the rounding of that chain: both against each other (1e-6 of the row maximum) and against
    the float64 autograd of the ATen port on sampled frames; both entries (spectrogram in, waveform in), ragged frame count."""
    stft, mcep = _modules()
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(37, 5000, generator=gen).to(DEV)
    w = torch.randn(25, generator=gen).to(DEV)
    fused = dsp.fuse(stft, mcep)

    def grads():
        X = stft(x).detach().requires_grad_(True)
        (mcep(X) * w).sum().backward()
        xg = x.clone().requires_grad_(True)
        (fused(xg) * w).sum().backward()
        return X.grad, xg.grad

    gX1, gx1 = grads()
    monkeypatch.setenv("DSA_MCEP_HIST_RT", "0")
    gX0, gx0 = grads()
    monkeypatch.delenv("DSA_MCEP_HIST_RT")
    assert float((gX1 - gX0).abs().max()) <= 1e-6 * float(gX0.abs().max())
    assert float((gx1 - gx0).abs().max()) <= 1e-6 * float(gx0.abs().max())
    tab = TP.McepTables(512, 24, 0.42, torch.float64)
    xs = x[[0, 36]].double().cpu().requires_grad_(True)
    (TP.stft_mcep(xs, tab) * w.double().cpu()).sum().backward()
    assert float((gx1[[0, 36]].double().cpu() - xs.grad).abs().max()) <= 3e-6 * float(xs.grad.abs().max())


def test_mcep_backward_two_waves_per_simd_against_the_one_wave_kernel_and_float64(monkeypatch):
    """Round 5: with the saved rt rows the backward runs as mcep_mfma_bwd2_kernel_h (two waves per SIMD: only lbar resident, the bins
    64 at a time with the group's own power-of-two scales, the forward's block elimination with the adjoint right-hand side riding
    along).  Same mathematics, different rounding points: against the one-wave kernel on the same saved rows (DSA_MCEP_BWD2=0) within
    the measured error of either (below), both against float64 autograd on sampled frames; repeated launches are bit-identical;
    a launch on a prefix of the frames gives the same bits (whole tiles); ragged frame counts; n_iter in {1, 3, 10}."""
    gen = torch.Generator().manual_seed(33)
    stft, _ = _modules()
    x = torch.randn(41, 4800, generator=gen).to(DEV)
    X = stft(x).reshape(-1, 257)[:2449]   # 153 tiles + 1 frame
    for n_iter in (1, 3, 10):
        mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=n_iter, device=DEV)
        w = torch.randn(2449, 25, generator=gen).to(DEV)

        def grad(Xin, wt):
            Xg = Xin.detach().clone().requires_grad_(True)
            (mcep(Xg) * wt).sum().backward()
            return Xg.grad

        g2 = grad(X, w)
        assert torch.equal(g2, grad(X, w))
        assert torch.equal(g2[:1600], grad(X[:1600], w[:1600]))
        monkeypatch.setenv("DSA_MCEP_BWD2", "0")
        g1 = grad(X, w)
        monkeypatch.delenv("DSA_MCEP_BWD2")
        assert torch.isfinite(g2).all()
        assert not torch.equal(g1, g2)   # (two different kernels ran: the backward's name is recorded on autograd's thread)
        # Both kernels against float64 autograd on ALL frames, per frame relative to the frame's largest gradient entry.  Measured
        # (tools/check_mcep_bwd_kernels.py; two-wave / one-wave): whole-tensor 1.2e-6 / 1.2e-6, median frame 7e-7 / 7e-7, worst frame
        # n_iter 1: 8.3e-5 / 4.1e-5, 3: 1.2e-4 / 4.0e-4, 10: 1.5e-6 / 3.4e-6 (unconverged iterates are ill-conditioned on a few frames,
        # for either kernel).  Bounds: 3 x the larger of the two.
        worst = {1: 2.5e-4, 3: 1.2e-3, 10: 1.1e-5}[n_iter]
        tab = TP.McepTables(512, 24, 0.42, torch.float64)
        Xs = X.double().cpu().requires_grad_(True)
        (TP.mcep(Xs, tab, n_iter) * w.double().cpu()).sum().backward()
        ref = Xs.grad
        for gk in (g2, g1):
            d = (gk.double().cpu() - ref).abs()
            per = d.amax(1) / ref.abs().amax(1)
            assert float(d.max() / ref.abs().max()) <= 3.5e-6
            assert float(per.median()) <= 2.5e-6
            assert float(per.max()) <= worst, (n_iter, float(per.max()))


def test_hot_path_and_f_rows_replay_from_a_hip_graph():
    """The launches of the analysis (STFT -> mcep), of the mel-generalized analysis and of the multi-stage MLSA filter are plain
    asynchronous launches on the current stream with caller-owned workspaces, so a whole call can be captured in a HIP graph
    (torch.cuda.CUDAGraph) and replayed on new data in the static input buffers: bit-identical to the eager call.  (Small
    batches are launch-bound from Python -- 20 .. 45 launches of 20 .. 90 us; a replay is one submission.)"""
    stft, mcep = _modules()
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=4, device=DEV)
    ml = dsp.MLSA(24, 80, alpha=0.42, mode="multi-stage", taylor_order=8, device=DEV)
    gen = torch.Generator().manual_seed(9)
    x_static = torch.randn(16, 8000, generator=gen).to(DEV)

    def step(x):
        X = stft(x)
        mc = mcep(X)
        return mc, mg(X), ml(x, mc[:, :100])

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():      # warm-up on the capture stream: images, LDS attributes, scratch, allocator
        for _ in range(2):
            step(x_static)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph, stream=side):
        outs_static = step(x_static)
    for seed in (1, 2):
        x_new = torch.randn(16, 8000, generator=torch.Generator().manual_seed(seed)).to(DEV)
        with torch.no_grad():
            want = step(x_new)
        x_static.copy_(x_new)
        graph.replay()
        torch.cuda.synchronize()
        for got, ref in zip(outs_static, want):
            assert torch.equal(got, ref)
    # the packaged form: diffsptk_amd.Graphed
    g = dsp.Graphed(step, x_static)
    x_new = torch.randn(16, 8000, generator=torch.Generator().manual_seed(5)).to(DEV)
    with torch.no_grad():
        want = step(x_new)
    got = g(x_new)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    with pytest.raises(ValueError):
        g(x_new[:8])


@pytest.mark.parametrize("F,K,N", [(300, 1025, 99), (1000, 50, 1025), (257, 513, 69), (64, 31, 17), (1, 1025, 121), (513, 1030, 260),
                                   (130, 1027, 130), (70, 6, 5), (33, 3, 40), (40, 35, 2), (65, 4, 4), (100, 64, 40), (70, 96, 130), (257, 33, 4)])
def test_rows_gemm_matrix_core_product_against_float64(F, K, N):
    """The general float32 row product (dsa_rows_gemm, csrc/rows_gemm.hip: what the 1025-bin products of the 48 kHz set-ups run
    on instead of a vendor GEMM): plain, transposed, with the log prologue and with the exp(aux - 2 .) epilogue of the untuned
    mel-cepstral step, ragged sizes in every dimension, against float64; 2e-6 of the largest entry (float32 accumulation of up
    to 1030 products)."""
    g = torch.Generator().manual_seed(F + K + N)
    c = torch.randn(F, K, generator=g)
    A = torch.randn(K, N, generator=g) / K ** 0.5
    cd, Ad = c.to(DEV), A.to(DEV)
    ref = c.double() @ A.double()
    out = ops.rows_gemm(cd, Ad)
    assert _lib.last_kernel() == "rows_gemm_mfma"
    assert float((out.double().cpu() - ref).abs().max()) < 2e-6 * float(ref.abs().max()) + 1e-6
    At = A.t().contiguous().to(DEV)                      # (N, K): B = At^T
    out_t = ops.rows_gemm(cd, At, ops.ROWS_TRANS)
    assert _lib.last_kernel() == "rows_gemm_mfma_t"
    assert float((out_t.double().cpu() - ref).abs().max()) < 2e-6 * float(ref.abs().max()) + 1e-6
    pos = (c.abs() + 0.1).to(DEV)
    out_l = ops.rows_gemm(pos, Ad, ops.ROWS_PRO_LOG)
    ref_l = torch.log(pos.double().cpu()) @ A.double()
    assert float((out_l.double().cpu() - ref_l).abs().max()) < 3e-6 * float(ref_l.abs().max()) + 1e-6
    aux = torch.randn(F, N, generator=g).to(DEV)
    out_e = ops.rows_gemm(0.1 * cd, Ad, ops.ROWS_EPI_EXPSUB, aux=aux)
    ref_e = torch.exp(aux.double().cpu() - 2 * (0.1 * c.double() @ A.double()))
    assert float(((out_e.double().cpu() - ref_e) / ref_e).abs().max()) < 5e-6
    # the row-product operator takes these shapes here (no vendor GEMM), forward and backward
    if F >= 256 and max(K, N) >= 512:
        cg = cd.clone().requires_grad_(True)
        y = ops.MatmulRowsFn.apply(cg, Ad)
        assert _lib.last_kernel() == "rows_gemm_mfma"
        w = torch.randn(F, N, generator=g).to(DEV)
        (y * w).sum().backward()           # (the backward's kernel name lives on autograd's thread: rows_gemm_mfma_t)
        gref = w.double().cpu() @ A.double().t()
        assert float((cg.grad.double().cpu() - gref).abs().max()) < 2e-6 * float(gref.abs().max()) + 1e-6


@pytest.mark.parametrize("F,K,M1", [(300, 1025, 50), (1000, 513, 35), (65, 1025, 55), (1, 257, 3), (130, 1027, 25), (64, 100, 41), (70, 64, 30),
                                    (33, 33, 5)])
def test_newton_residual_in_one_launch_against_float64(F, K, M1):
    """dsa_mcep_newton_resid: rt = exp(log X - 2 mc D) E (mcep.py:210-215) with e formed in registers, against float64 and against
    the two-launch composition it replaces, on the warping matrices of a real configuration and on ragged sizes (K not a multiple of
    the 32-bin chunk, rows not a multiple of 64, orders at the ends of the three instantiations).  3e-6 of the largest entry: float32
    products over up to 1027 bins plus the float32 exp."""
    g = torch.Generator().manual_seed(F + K + M1)
    N = 2 * M1 - 1
    logx = (torch.randn(F, K, generator=g) * 0.5).to(DEV)
    mc = (torch.randn(F, M1, generator=g) * 0.05).to(DEV)
    D = (torch.randn(M1, K, generator=g) / M1 ** 0.5).to(DEV)
    E = (torch.randn(K, N, generator=g) / K).to(DEV)
    rt = ops.mcep_newton_resid(logx, mc, D, E)
    assert _lib.last_kernel() == "mcep_resid_mfma"
    ref = torch.exp(logx.double() - 2 * mc.double() @ D.double()) @ E.double()
    assert float((rt.double() - ref).abs().max()) < 3e-6 * float(ref.abs().max())
    two = ops.rows_gemm(ops.rows_gemm(mc, D, ops.ROWS_EPI_EXPSUB, aux=logx), E)
    assert float((rt - two).abs().max()) < 3e-6 * float(ref.abs().max())


@pytest.mark.parametrize("F,K,M1", [(300, 1025, 50), (1000, 513, 35), (65, 1025, 55), (1, 257, 3), (130, 1027, 25), (64, 100, 41), (70, 64, 30),
                                    (33, 33, 5), (17, 2049, 33), (129, 129, 32)])
def test_newton_residual_on_binary16_splits_against_float64(F, K, M1):
    """dsa_mcep_newton_resid_h (round 5): the same rt = exp(log X - 2 mc D) E with both products as 3-term binary16 splits on the matrix
    pipe (operand images prepared once by dsa_mcep_resid_prepare, a stage's e scaled by its own power of two): against float64 (2e-6
    of the largest entry; measured 4-6e-7, the float32 matrix instruction's 0.8-1.4e-6) and the float32 launch, on ragged sizes (K not
    a multiple of 32, rows not a multiple of 64, orders at both ends of the one- and two-k-step instantiations), with spectra whose
    level moves by 2^40 along a row (the per-stage scale) and rows that differ by 2^60 (the per-frame scale of mc); repeated launches
    bit-identical; a prefix of the rows gives the same bits."""
    g = torch.Generator().manual_seed(F + K + M1)
    N = 2 * M1 - 1
    logx = (torch.randn(F, K, generator=g) * 0.5 + torch.linspace(-14, 14, K)[None, :]).to(DEV)
    mc = (torch.randn(F, M1, generator=g) * 0.05 * (2.0 ** torch.randint(-30, 3, (F, 1), generator=g).float())).to(DEV)
    D = (torch.randn(M1, K, generator=g) / M1 ** 0.5).to(DEV)
    E = (torch.randn(K, N, generator=g) / K).to(DEV)
    img = ops.mcep_resid_images(D, E)
    assert img is not None and img.numel() == ((K + 31) // 32) * (4 * ((M1 + 31) // 32) + 2 * ((N + 15) // 16)) * 512 * 2
    rt = ops.mcep_newton_resid_h(logx, mc, img)
    assert _lib.last_kernel() == "mcep_resid_h"
    ref = torch.exp(logx.double() - 2 * mc.double() @ D.double()) @ E.double()
    assert torch.isfinite(rt).all()
    assert float((rt.double() - ref).abs().max()) < 2e-6 * float(ref.abs().max())
    assert float((rt - ops.mcep_newton_resid(logx, mc, D, E)).abs().max()) < 4e-6 * float(ref.abs().max())
    assert torch.equal(rt, ops.mcep_newton_resid_h(logx, mc, img))
    if F > 16:
        assert torch.equal(rt[:16], ops.mcep_newton_resid_h(logx[:16].contiguous(), mc[:16].contiguous(), img))


def test_untuned_analysis_runs_on_the_library_s_own_kernels_only():
    """No stock operator carries data in the untuned mel-cepstral analysis: the kernel names of a forward and of a forward +
    backward call at a 48 kHz set-up are all the library's (dsa_last_kernel after every launch is not observable from here, so
    the operator list is checked through the autograd graph and torch's profiler)."""
    from torch.profiler import ProfilerActivity, profile

    x = torch.randn(4, 16 * 2048, generator=torch.Generator().manual_seed(4)).to(DEV)
    stft = dsp.STFT(1200, 240, 2048, device=DEV)
    mcep = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=49, alpha=0.55, n_iter=3, device=DEV)
    with torch.no_grad():
        X = stft(x)
        mcep(X)
    Xg = X.clone().requires_grad_(True)
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        with torch.no_grad():
            mcep(X)
        mcep(Xg).sum().backward()
    names = {e.key for e in prof.key_averages()}
    banned = {"aten::matmul", "aten::mm", "aten::bmm", "aten::addmm", "aten::exp", "aten::log", "aten::linalg_solve", "aten::linalg_solve_ex"}
    assert not (names & banned), names & banned


@pytest.mark.parametrize("n,F", [(50, 300), (35, 65), (7, 130)])
def test_newton_update_with_a_gradient_against_float64_autograd(n, F):
    """ops.McepNewtonUpdateFn (dsa_mcep_newton_update with mc_in = NULL + dsa_mcep_newton_update_bwd): mc + solve(T(rt[:n]) + H(rt),
    rt[:n] - av) and the cotangents of rt and mc against float64 autograd through the dense solve, on the Hessians of the analysis
    (positive definite); 5e-5 of the largest entry at condition numbers of a few hundred."""
    import math

    g = torch.Generator().manual_seed(n + F)
    K = 4 * n
    w = torch.arange(K, dtype=torch.float64) * (math.pi / K)
    cw = torch.cos(torch.arange(2 * n - 1, dtype=torch.float64)[:, None] * w[None, :])
    e = torch.rand(F, K, generator=g, dtype=torch.float64) + 0.05
    rt64 = ((e @ cw.t()) / K).float().double()
    av = (torch.randn(n, generator=g, dtype=torch.float64) * 0.1).float().double()
    mc64 = torch.randn(F, n, generator=g, dtype=torch.float64).float().double()
    wgt = torch.randn(F, n, generator=g, dtype=torch.float64).float().double()

    def dense(rt):
        i = torch.arange(n)
        return rt[:, (i[:, None] - i[None, :]).abs()] + rt[:, i[:, None] + i[None, :]]

    rt_r, mc_r = rt64.clone().requires_grad_(True), mc64.clone().requires_grad_(True)
    out_r = mc_r + torch.linalg.solve(dense(rt_r), rt_r[:, :n] - av)
    (out_r * wgt).sum().backward()
    rt_d, mc_d = rt64.float().to(DEV).requires_grad_(True), mc64.float().to(DEV).requires_grad_(True)
    out_d = ops.McepNewtonUpdateFn.apply(rt_d, av.float().to(DEV), mc_d)
    assert _lib.last_kernel() in ("th_solve_quadn_fwd", "th_solve_octn_fwd")
    (out_d * wgt.float().to(DEV)).sum().backward()
    cond = float(torch.linalg.cond(dense(rt64)).max())
    tol = 5e-6 * cond ** 0.5 + 5e-5
    assert float((out_d.detach().double().cpu() - out_r.detach()).abs().max() / out_r.detach().abs().max()) < tol
    assert float((rt_d.grad.double().cpu() - rt_r.grad).abs().max() / rt_r.grad.abs().max()) < tol
    assert torch.equal(mc_d.grad.cpu(), wgt.float())


@pytest.mark.parametrize("M", [32, 34, 35, 39, 42, 43, 49, 50, 51, 54])
def test_48khz_newton_steps_in_one_launch_equal_the_two_launch_step_bit_for_bit(M, monkeypatch):
    """dsa_mcep_newton_steps (round 6, csrc/mcep_big_f16.h: all Newton steps of mcep.py:208-222 at fft_length 2048 / orders 32 .. 54 in
    one persistent launch -- 12 800 frames 0.92 -> 0.68 ms) against the two launches per step it replaces: the same bits at every
    batch size, ragged tails and iteration counts, repeat launches identical, float64 at the tolerance of the goldens, non-finite
    frames contained, and the kernel chosen by the order alone (1 frame or 102 400)."""
    g = torch.Generator().manual_seed(M)
    Xall = (torch.randn(3300, 1025, generator=g).square() + 0.05).to(DEV)
    for F, n_iter in ((1, 3), (15, 1), (64, 10), (65, 2), (700, 10), (3217, 10)):
        X = Xall[:F]
        m = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=M, alpha=0.55, n_iter=n_iter, device=DEV)
        with torch.no_grad():
            monkeypatch.setenv("DSA_MCEP_BIG", "0")
            a = m(X)
            assert _lib.last_kernel() != "mcep_big_newton"
            monkeypatch.setenv("DSA_MCEP_BIG", "1")
            b = m(X)
            assert _lib.last_kernel() == "mcep_big_newton", (F, _lib.last_kernel())
            assert torch.equal(a, b), (F, n_iter, float((a - b).abs().max()))
            assert torch.equal(m(X), b)
    m64 = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=M, alpha=0.55, n_iter=10, device=DEV, dtype=torch.float64)
    with torch.no_grad():
        y64 = m64(Xall[:700].double())
    np.testing.assert_allclose(host(b[:700]), host(y64), **MC32)
    # a non-finite frame stays in its row: every other frame keeps its bits
    Xb = Xall[:700].clone()
    Xb[[3, 64, 699], 7] = float("nan")
    m = dsp.MelCepstralAnalysis(fft_length=2048, cep_order=M, alpha=0.55, n_iter=10, device=DEV)
    with torch.no_grad():
        yb, y = m(Xb), m(Xall[:700])
    keep = torch.ones(700, dtype=torch.bool, device=DEV)
    keep[[3, 64, 699]] = False
    assert torch.equal(yb[keep], y[keep]) and not torch.isfinite(yb[~keep]).all(-1).any()
    # the same kernel, the same bits per frame, at 200 tiles of 64 frames (one round of the chip) and at 1 600 (6.25 rounds)
    with torch.no_grad():
        Xbig = Xall[:3200].repeat(4, 1)                                   # 12 800 frames
        y1 = m(Xbig)
        assert _lib.last_kernel() == "mcep_big_newton"
        assert torch.equal(y1[:3200], y1[3200:6400]) and torch.equal(y1[:700], y)
        y8 = m(Xbig.repeat(8, 1))                                        # 102 400 frames
        assert _lib.last_kernel() == "mcep_big_newton"
        assert torch.equal(y8[:12800], y1) and torch.equal(y8[-12800:], y1)


@pytest.mark.parametrize("M,nfft", [(34, 1024), (32, 2048), (42, 2048), (49, 2048), (54, 2048)])
def test_48khz_newton_steps_wide_tiles_and_plan_give_the_narrow_tiles_bits(M, nfft, monkeypatch):
    """The wide tile shape of dsa_mcep_newton_steps (128 frames per workgroup, a wave per 16-frame group running both stages of a staged
    pair and solving its 16 systems itself) and the plan that mixes rounds of wide tiles with narrow ones (csrc/mcep_mfma.hip:
    mcep_big_newton) against the narrow tiles: the same bits per frame at ragged sizes, in every round of a multi-round launch and
    across the plan's two launches; non-finite frames contained."""
    K = nfft // 2 + 1
    g = torch.Generator().manual_seed(100 + M)
    Xall = (torch.randn(3300, K, generator=g).square() + 0.05).to(DEV)
    m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=DEV)
    for F, n_iter in ((1, 2), (15, 10), (127, 3), (129, 10), (700, 10), (3217, 1)):
        mi = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=n_iter, device=DEV)
        with torch.no_grad():
            monkeypatch.setenv("DSA_MCEP_BIG_WIDE", "0")
            a = mi(Xall[:F])
            monkeypatch.setenv("DSA_MCEP_BIG_WIDE", "1")
            b = mi(Xall[:F])
            assert _lib.last_kernel() == "mcep_big_newton"
            assert torch.equal(a, b), (F, n_iter, float((a - b).abs().max()))
            assert torch.equal(mi(Xall[:F]), b)
    Xb = Xall[:700].clone()
    Xb[[3, 64, 130, 699], 7] = float("nan")
    with torch.no_grad():
        monkeypatch.setenv("DSA_MCEP_BIG_WIDE", "1")
        yb, y = m(Xb), m(Xall[:700])
    keep = torch.ones(700, dtype=torch.bool, device=DEV)
    keep[[3, 64, 130, 699]] = False
    assert torch.equal(yb[keep], y[keep]) and not torch.isfinite(yb[~keep]).all(-1).any()
    with torch.no_grad():
        Xbig = Xall[:3200].repeat(13, 1)[:40000]                          # 40 000 frames: 1.2 rounds of wide tiles, 2.4 of narrow ones
        monkeypatch.setenv("DSA_MCEP_BIG_WIDE", "0")
        yn = m(Xbig)
        monkeypatch.setenv("DSA_MCEP_BIG_WIDE", "1")
        yw = m(Xbig)
        monkeypatch.delenv("DSA_MCEP_BIG_WIDE")
        yp = m(Xbig)                                                     # the plan: one round of wide tiles + narrow tiles for the rest
        assert torch.equal(yn, yw) and torch.equal(yn, yp)
        assert torch.equal(yp[:3200], yp[3200:6400]) and torch.equal(yp[:700], y) and torch.equal(yp[35200:38400], yp[:3200])


@pytest.mark.parametrize("M,nfft", [(34, 1024), (32, 2048), (33, 512)])
def test_48khz_newton_steps_twin_workgroups_give_the_same_bits(M, nfft, monkeypatch):
    """The TWIN-workgroup shape of dsa_mcep_newton_steps at the quad-layout orders (csrc/mcep_big4_f16.h: two four-wave workgroups per CU,
    the second half a step late) against the eight-wave wide tiles, the narrow tiles and the two launches per step: the same bits per
    frame at ragged sizes, over several rounds, whatever the stagger; non-finite frames contained."""
    K = nfft // 2 + 1
    g = torch.Generator().manual_seed(300 + M)
    Xall = (torch.randn(3300, K, generator=g).square() + 0.05).to(DEV)

    def run(mod, X, wide, twin, stagger=None, big="2"):
        monkeypatch.setenv("DSA_MCEP_BIG", big)
        monkeypatch.setenv("DSA_MCEP_BIG_WIDE", wide)
        monkeypatch.setenv("DSA_MCEP_BIG_TWIN", twin)
        if stagger is None:
            monkeypatch.delenv("DSA_MCEP_BIG_STAGGER", raising=False)
        else:
            monkeypatch.setenv("DSA_MCEP_BIG_STAGGER", stagger)
        with torch.no_grad():
            return mod(X)

    for F, n_iter in ((1, 2), (15, 10), (64, 0), (65, 3), (129, 10), (700, 10), (3217, 1)):
        mi = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=n_iter, device=DEV)
        t = run(mi, Xall[:F], "1", "1")
        assert n_iter == 0 or _lib.last_kernel() == "mcep_big_newton"
        assert torch.equal(t, run(mi, Xall[:F], "1", "0")), (F, n_iter)         # eight-wave wide tiles
        assert torch.equal(t, run(mi, Xall[:F], "0", "0")), (F, n_iter)         # narrow tiles
        assert torch.equal(t, run(mi, Xall[:F], "1", "1", "0")), (F, n_iter)    # no stagger
        assert torch.equal(t, run(mi, Xall[:F], "1", "1", "11")), (F, n_iter)
    m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=DEV)
    y = run(m, Xall[:700], "1", "1")
    assert torch.equal(y, run(m, Xall[:700], "0", "0", big="0"))                # two launches per step
    Xb = Xall[:700].clone()
    Xb[[3, 64, 130, 699], 7] = float("nan")
    yb = run(m, Xb, "1", "1")
    keep = torch.ones(700, dtype=torch.bool, device=DEV)
    keep[[3, 64, 130, 699]] = False
    assert torch.equal(yb[keep], y[keep]) and not torch.isfinite(yb[~keep]).all(-1).any()
    Xbig = Xall[:3200].repeat(23, 1)[:70001]                                    # 2.1 rounds of 512 twin workgroups, ragged end
    yt = run(m, Xbig, "1", "1")
    assert torch.equal(yt, run(m, Xbig, "1", "0"))
    monkeypatch.delenv("DSA_MCEP_BIG_WIDE")
    with torch.no_grad():
        yp = m(Xbig)                                                            # the plan: twin rounds + narrow tiles for the rest
    assert torch.equal(yp, yt)
    assert torch.equal(yt[:3200], yt[3200:6400]) and torch.equal(yt[:700], y) and torch.equal(yt[64000:67200], yt[:3200])
