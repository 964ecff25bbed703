"""Data-parallel sharding of utterance batches across GPUs (one process per GPU).

Every utterance -- in fact every frame -- of the analysis path is independent (frame.py:65-70,
mcep.py:83-88 of the reference broadcast over leading dims), so the path shards by whole
utterances with NO data-path collective; the only exchange is one all-gather of the final
(B, N, M+1) feature tensor (RCCL over xGMI: torch.distributed backend "nccl").  The reference
has no distributed code at all.
"""
from __future__ import annotations

import os
from typing import Callable

import torch
import torch.distributed as dist


def reserved_cus() -> int:
    """CUs the persistent analysis launches leave free while gathers are in flight (ops.reserve_cus, DSA_ALGO_RESERVE_CUS;
    DSA_RESERVE_CUS overrides, 0: none).  A persistent workgroup fills its CU, so the collective's kernel -- on RCCL's own stream --
    starts in a launch's tail and keeps its CUs into the next launch.  What that costs depends on WHEN the analysis waits for it,
    measured with a stand-in collective (W workgroups holding their CUs for 350 us on a second stream, ordered after the step's kernel;
    tools/ab_reserve_cus.py, profiles/r06_reserve_cus_ab.txt; the analysis alone: 0.584 ms per step):
      * completed ONE step later (the next step's launch waits for it):  0.86-0.92 ms with no CU reserved -- analysis and exchange in
        series --, 0.69 with 16-27 reserved;
      * completed TWO steps later (GATHER_DEPTH = 2, what bench.py does):  0.627 ms with none reserved, **0.618 with 8** (0.65 / 0.66 for
        32 / 64 workgroups: any channel count fits); three steps: the same.
    Hence 8 (the launcher rounds up to 9 where that costs no round of tiles: +1.2 % on the launch alone)."""
    try:
        return max(0, min(63, int(os.environ.get("DSA_RESERVE_CUS", "8"))))
    except ValueError:
        return 8


#: how many batches later a streaming caller should complete a deferred gather (PendingGather.wait()): see reserved_cus()
GATHER_DEPTH = 2


def shard_bounds(total: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced split of ``total`` utterances: the first ``total % world_size``
    ranks get one extra.  Returns [lo, hi)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank must be in [0, world_size)")
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_features(local: torch.Tensor, total: int | None = None, group=None) -> torch.Tensor:
    """Concatenate per-rank feature shards (B_r, ...) along dim 0 on every rank.

    Equal shards use ONE ``all_gather_into_tensor`` (a single large collective: on the xGMI mesh
    each rank's shard goes out over its 7 links at once); ragged shards are padded to the largest
    shard for the same single collective and trimmed afterwards.
    """
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    if world == 1:
        return local
    local = local.contiguous()
    if total is None:
        n = torch.tensor([local.size(0)], device=local.device, dtype=torch.int64)
        dist.all_reduce(n, group=group)
        total = int(n.item())
    sizes = [shard_bounds(total, world, r) for r in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    biggest = max(counts)
    if local.size(0) != counts[dist.get_rank(group)]:
        raise ValueError("local shard size does not match the contiguous balanced split")
    if min(counts) == biggest:
        out = local.new_empty((total, *local.shape[1:]))
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    padded = local.new_zeros((biggest, *local.shape[1:]))
    padded[: local.size(0)] = local
    buf = local.new_empty((world * biggest, *local.shape[1:]))
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * biggest: r * biggest + c] for r, c in enumerate(counts)], dim=0)


def analyze_sharded(x: torch.Tensor, compute: Callable[[torch.Tensor], torch.Tensor], gather: bool = True,
                    group=None) -> torch.Tensor:
    """Run ``compute`` (e.g. ``lambda w: mcep(stft(w))``) on this rank's contiguous shard of the
    global batch ``x`` (B, T) and optionally all-gather the features."""
    if not (dist.is_available() and dist.is_initialized()):
        return compute(x)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(x.size(0), world, rank)
    local = compute(x[lo:hi])
    return all_gather_features(local, x.size(0), group) if gather else local


class PendingGather:
    """Handle of the collectives a deferred ``analyze_chunked_overlap`` left in flight: ``wait()`` orders the
    current stream (or, on CPU, the caller) after them.  The gathered tensor must not be read before."""

    def __init__(self, works, keep):
        self._works, self._keep = list(works), list(keep)

    def wait(self) -> None:
        for w in self._works:
            w.wait()
        self._works, self._keep = [], []


def analyze_chunked_overlap(x_local: torch.Tensor, compute: Callable[[torch.Tensor], torch.Tensor], n_chunks: int = 2,
                            group=None, alternate_streams: bool = True, defer: bool = False,
                            layout: str = "rank_major", force_collective: bool = False):
    """Compute this rank's shard ``x_local`` (B_r, T) in ``n_chunks`` utterance chunks and all-gather
    each chunk's features as soon as they exist, so the collective of chunk c runs (on RCCL's own
    stream) while chunk c+1 is being computed.  Every rank must hold the same number of utterances.

    Every chunk is ONE ``all_gather_into_tensor`` that writes in place: the receive buffer is laid out
    ``(n_chunks, world, B_c, ...)``, so chunk c's collective fills the contiguous block ``buf[c]`` (rank-major
    inside the chunk) and RCCL neither stages through a flat temporary nor copies out (a list of strided
    destination views would: +164 MB of device copies per rank and step at 8 x 1024 utterances).  Chunks are
    therefore EQUAL: ``n_chunks`` is lowered to the largest divisor of B_r not above the request.

    ``layout``:
      * ``"rank_major"`` (default): returns (world * B_r, ...) in rank-major utterance order -- the tensor
        ``all_gather_features(compute(x_local))`` returns.  With one chunk this is the receive buffer itself
        (no copy); with several chunks it is one strided device copy of the buffer after the last collective.
      * ``"chunk_major"``: returns the receive buffer (n_chunks, world, B_c, ...) as is; utterance
        ``b = c * B_c + i`` of rank ``r`` is ``out[c, r, i]``.  No copy at any chunk count.
    A streaming caller that wants the exchange hidden and rank-major order should use ONE chunk with
    ``defer=True``: the whole gather then runs behind the next batch's kernels (bench.py does this).

    On a GPU the chunks alternate between the current stream and one side stream
    (``alternate_streams``): the persistent mel-cepstral kernel ends in a tail of partly idle CUs,
    and a chunk launched on the other stream fills that tail instead of queueing behind it
    (tools/ab_chunks.py: 2 chunks on 2 streams cost the same as one unchunked launch, 0.96 ms per
    1024 utterances; 4 chunks cost +20 %).

    ``defer=True`` returns ``(features, PendingGather)`` without waiting for the collectives: the LAST chunk's
    all-gather has nothing of this call left to hide behind, but the caller's next batches can -- call
    ``PendingGather.wait()`` before the features are read; a streaming caller does so ``GATHER_DEPTH`` = 2 batches later
    (one batch later the next launch waits for a kernel that could only start in this launch's tail: reserved_cus()).
    ``force_collective`` runs the collectives even in a world of one (functional test of the RCCL path on one GPU).
    """
    if layout not in ("rank_major", "chunk_major"):
        raise ValueError("layout must be 'rank_major' or 'chunk_major'")
    ready = dist.is_available() and dist.is_initialized()
    if not ready or (dist.get_world_size(group) == 1 and not force_collective):
        out1 = compute(x_local)
        if layout == "chunk_major":
            out1 = out1.reshape(1, 1, *out1.shape)
        return (out1, PendingGather([], [])) if defer else out1
    world = dist.get_world_size(group)
    B = x_local.size(0)
    n_chunks = max(1, min(n_chunks, B))
    while B % n_chunks:          # equal chunks: every collective writes one contiguous block in place
        n_chunks -= 1
    Bc = B // n_chunks
    use_side = alternate_streams and x_local.is_cuda and n_chunks > 1
    main = torch.cuda.current_stream(x_local.device) if use_side else None
    side = torch.cuda.Stream(x_local.device) if use_side else None
    if use_side:
        side.wait_stream(main)  # the input was produced on the current stream
    buf = None
    pending = []
    if x_local.is_cuda:
        from . import ops

        reserve = ops.reserve_cus(reserved_cus())   # the gathers run beside the next chunk's / the next batch's launches
    else:
        reserve = _NullContext()
    for c in range(n_chunks):
        on_side = use_side and c % 2 == 1
        if on_side and buf is not None:
            side.wait_stream(main)  # `buf` was allocated on the current stream
        ctx = torch.cuda.stream(side) if on_side else _NullContext()
        with ctx, reserve:
            feat = compute(x_local[c * Bc:(c + 1) * Bc]).contiguous()
            if buf is None:
                buf = feat.new_empty((n_chunks, world, Bc, *feat.shape[1:]))
            # the collective is ordered after this chunk's kernels on the stream that is current here and
            # fills buf[c] = (world, Bc, ...) directly
            work = dist.all_gather_into_tensor(buf[c].view(world * Bc, *feat.shape[1:]), feat, group=group, async_op=True)
        pending.append((work, feat))
    if use_side:
        main.wait_stream(side)
        for _work, keep in pending:
            keep.record_stream(main)
    handle = PendingGather([w for w, _ in pending], [k for _, k in pending] + [buf])
    if layout == "chunk_major":
        res = buf
    elif n_chunks == 1:
        res = buf.reshape(world * B, *buf.shape[3:])      # a view: rank-major already
    else:
        handle.wait()                                      # the reorder reads every chunk
        res = buf.transpose(0, 1).reshape(world * B, *buf.shape[3:])   # one strided device copy
        handle = PendingGather([], [buf])
    if defer:
        return res, handle
    handle.wait()
    return res


class _NullContext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False
