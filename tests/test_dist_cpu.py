"""world_size-2 gloo test of the N>1 path (sharding + the single all-gather) on CPU.
The compute stand-in is the oracle (tests may use it); the product kernels need a GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffsptk_amd.dist import all_gather_features, analyze_chunked_overlap, analyze_sharded, shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_balance():
    for total in (0, 1, 7, 8, 8192, 8191):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    from oracle import oracle as O

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = torch.randn(B, 1200, generator=torch.Generator().manual_seed(0), dtype=torch.float64)

        def compute(w):
            return torch.from_numpy(O.stft_mcep(w.numpy(), cep_order=8, n_iter=2))

        out = analyze_sharded(x, compute)
        lo, hi = shard_bounds(B, world, rank)
        loc = analyze_sharded(x, compute, gather=False)
        ok = out.shape == (B, 15, 9) and loc.shape[0] == hi - lo and torch.equal(out[lo:hi], loc)
        again = all_gather_features(loc, B)
        ok = ok and torch.equal(again, out)
        if B % world == 0:  # the overlapped variant needs equal shards
            for nch in (1, 2, 3):
                ov = analyze_chunked_overlap(x[lo:hi], compute, n_chunks=nch)
                ok = ok and torch.equal(ov, out)
            # deferred completion: two batches in flight, each read only after its own handle was waited for
            ov1, h1 = analyze_chunked_overlap(x[lo:hi], compute, n_chunks=2, defer=True)
            ov2, h2 = analyze_chunked_overlap(x[lo:hi], compute, n_chunks=3, defer=True)
            h1.wait()
            ok = ok and torch.equal(ov1, out)
            h2.wait()
            h2.wait()   # idempotent
            ok = ok and torch.equal(ov2, out)
            # chunk-major layout: the receive buffer itself, (n_chunks, world, B_c, ...)
            cm = analyze_chunked_overlap(x[lo:hi], compute, n_chunks=2, layout="chunk_major")
            nch, Bc = cm.shape[0], cm.shape[2]
            ok = ok and cm.shape[:3] == (nch, world, (hi - lo) // nch)
            for c in range(nch):
                for r in range(world):
                    rlo, _ = shard_bounds(B, world, r)
                    ok = ok and torch.equal(cm[c, r], out[rlo + c * Bc: rlo + (c + 1) * Bc])
            one, h = analyze_chunked_overlap(x[lo:hi], compute, n_chunks=1, defer=True)   # what bench.py does for N > 1
            h.wait()
            ok = ok and torch.equal(one, out)
        if rank == 0:
            q.put((ok, out.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5])  # equal and ragged shards
def test_two_rank_shard_and_gather(B):
    from oracle import oracle as O

    O.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
    x = torch.randn(B, 1200, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    np.testing.assert_allclose(out, O.stft_mcep(x.numpy(), cep_order=8, n_iter=2), rtol=1e-12, atol=1e-12)
