"""Multi-GPU plumbing: clips shard over ranks, the data path has no collective.

Clips (and STFT rows) are independent — no halo, no reduction across clips
when the log-mel / top_db maximum is taken per clip, which is what every
caller in the reference does (models/whisper/inference.py:413-423 passes one
clip per call).  So the partition is a contiguous block of clips per rank,
one process per GPU, tables rebuilt per device.  The only collectives are
outside the data path: a MAX all-reduce of wall time for measurement and an
optional gather of the finished features to one rank - chunked point-to-point
transfers into that rank only, on a side stream so that they overlap the next
batch (RCCL over xGMI when the tensors live on GPUs, gloo on the CPU).
"""

from __future__ import annotations

import torch


def clip_shard(n_clips: int, rank: int, world: int) -> tuple[int, int]:
    """[start, stop) of the contiguous block of clips rank `rank` owns.

    Blocks are ceil(n/world) long; the tail ranks may own fewer (or no) clips.
    """
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    if n_clips < 0:
        raise ValueError(f"n_clips must be non-negative, got {n_clips}")
    per = -(-n_clips // world)
    start = min(rank * per, n_clips)
    return start, min(start + per, n_clips)


def max_over_ranks(value: float, device: torch.device | str = "cpu") -> float:
    """MAX all-reduce of a scalar (wall seconds of the timed region)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _chunks(n: int, chunk: int):
    for a in range(0, n, chunk):
        yield a, min(a + chunk, n)


class FeatureGather:
    """Collects per-rank feature blocks (shards of axis 0) on rank `dst` ONLY.

    Rank `dst` owns one (n_clips, ...) buffer; every other rank sends its block in chunks
    of `chunk_clips` clips, and `dst` receives each chunk straight into its slice of that
    buffer (grouped point-to-point operations: `ncclSend` / `ncclRecv` under one group call
    per round on GPUs, gloo on the CPU).  Nothing is padded, nothing is concatenated, and no
    rank but `dst` holds more than its own block - an all-gather would materialise the whole
    tensor on every rank.  Gather-to-one is bound by `dst`'s inbound xGMI links (7 peers x
    one link each), so the chunks of all peers of a round travel together.

    On GPUs the transfers are issued on a side stream that first waits for the producing
    stream, so a caller can start the next batch's kernels while this batch is in flight:

        g = FeatureGather(n_clips, feat_shape, dtype, device)
        work = g.start(local)        # returns at once; transfers run beside later kernels
        ... next batch ...
        full = g.finish(work)        # (n_clips, ...) on dst, None elsewhere
    """

    def __init__(self, n_clips: int, feat_shape, dtype, device, dst: int = 0,
                 chunk_clips: int = 64):
        import torch.distributed as dist

        self.dist = dist
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.n_clips, self.dst, self.chunk = n_clips, dst, max(int(chunk_clips), 1)
        self.feat_shape, self.dtype, self.device = tuple(feat_shape), dtype, torch.device(device)
        self.world = dist.get_world_size() if self.on else 1
        self.rank = dist.get_rank() if self.on else 0
        self.start_clip, self.stop_clip = clip_shard(n_clips, self.rank, self.world)
        self.full = None
        if self.on and self.rank == dst:
            self.full = torch.empty((n_clips,) + self.feat_shape, dtype=dtype, device=self.device)
        self.stream = (torch.cuda.Stream(device=self.device)
                       if self.on and self.device.type == "cuda" else None)

    def start(self, local: torch.Tensor):
        """Begin collecting `local` (this rank's block); returns a handle for finish()."""
        if not self.on:
            return local
        if local.shape[0] != self.stop_clip - self.start_clip:
            raise ValueError(f"rank {self.rank} holds {local.shape[0]} clips, its shard has "
                             f"{self.stop_clip - self.start_clip}")
        if tuple(local.shape[1:]) != self.feat_shape or local.dtype != self.dtype:
            raise ValueError("feature block does not match the gather's shape / dtype")
        local = local.contiguous()
        dist = self.dist
        ev = None
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        reqs = []

        def issue():
            if self.rank == self.dst:
                # own block: a device copy; peers: one receive per chunk into its slice
                self.full[self.start_clip:self.stop_clip].copy_(local, non_blocking=True)
                rounds = max((clip_shard(self.n_clips, r, self.world)[1]
                              - clip_shard(self.n_clips, r, self.world)[0] + self.chunk - 1)
                             // self.chunk for r in range(self.world))
                for k in range(rounds):
                    ops = []
                    for r in range(self.world):
                        if r == self.dst:
                            continue
                        a, b = clip_shard(self.n_clips, r, self.world)
                        lo, hi = a + k * self.chunk, min(a + (k + 1) * self.chunk, b)
                        if lo < hi:
                            ops.append(dist.P2POp(dist.irecv, self.full[lo:hi], r))
                    if ops:
                        reqs.extend(dist.batch_isend_irecv(ops))
            else:
                for lo, hi in _chunks(local.shape[0], self.chunk):
                    reqs.extend(dist.batch_isend_irecv(
                        [dist.P2POp(dist.isend, local[lo:hi], self.dst)]))

        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                issue()
        else:
            issue()
        return (reqs, local)  # `local` stays alive until the transfers are done

    def finish(self, work):
        """Wait for the transfers of start(); the full tensor on dst, None elsewhere."""
        if not self.on:
            return work
        reqs, _local = work
        for r in reqs:
            r.wait()
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return self.full if self.rank == self.dst else None


def gather_features(local: torch.Tensor, n_clips: int, dst: int = 0, chunk_clips: int = 64):
    """Collect per-rank feature blocks (shard of axis 0) on rank `dst`: the (n_clips, ...)
    tensor on `dst`, None elsewhere (one blocking FeatureGather round)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    g = FeatureGather(n_clips, local.shape[1:], local.dtype, local.device, dst, chunk_clips)
    return g.finish(g.start(local))
