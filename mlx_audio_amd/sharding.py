"""Multi-GPU plumbing: clips shard over ranks, the data path has no collective.

Clips (and STFT rows) are independent — no halo, no reduction across clips
when the log-mel / top_db maximum is taken per clip, which is what every
caller in the reference does (models/whisper/inference.py:413-423 passes one
clip per call).  So the partition is a contiguous block of clips per rank,
one process per GPU, tables rebuilt per device.  The only collectives are
outside the data path: a MAX all-reduce of wall time for measurement and an
optional gather of the finished features to one rank (RCCL over xGMI when the
tensors live on GPUs, gloo on the CPU).
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


def gather_features(local: torch.Tensor, n_clips: int, dst: int = 0):
    """Collect per-rank feature blocks (shard of axis 0) on rank `dst`.

    One all_gather of equal-sized blocks (the ragged tail is zero-padded for
    the exchange and trimmed afterwards): on GPUs this is a single RCCL ring
    over xGMI instead of world-1 separate sends into `dst`.  Returns the
    (n_clips, ...) tensor on `dst`, None elsewhere.
    """
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-n_clips // world)
    start, stop = clip_shard(n_clips, rank, world)
    if local.shape[0] != stop - start:
        raise ValueError(f"rank {rank} holds {local.shape[0]} clips, its shard has {stop - start}")
    block = local
    if local.shape[0] != per:
        block = local.new_zeros((per,) + tuple(local.shape[1:]))
        block[: local.shape[0]] = local
    parts = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(parts, block.contiguous())
    if rank != dst:
        return None
    return torch.cat(parts, dim=0)[:n_clips]
