"""world_size-2 check of the multi-GPU plumbing on the CPU (gloo).

The data path has no collective: each rank processes its own block of clips.
Here the per-rank compute is the CPU oracle (the HIP path needs a GPU); what
is under test is the partition, the MAX reduction used for timing and the
optional feature gather — the same helpers bench.py uses under RCCL.
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mlx_audio_amd.sharding import clip_shard, gather_features, max_over_ranks


def test_clip_shard_partitions_exactly():
    for n in (0, 1, 5, 8, 1023, 1024):
        for world in (1, 2, 3, 8):
            blocks = [clip_shard(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (a0, a1), (b0, b1) in zip(blocks, blocks[1:]):
                assert a1 == b0 and a0 <= a1 and b0 <= b1
            assert max(b - a for a, b in blocks) == -(-n // world)
    with pytest.raises(ValueError):
        clip_shard(4, 2, 2)
    with pytest.raises(ValueError):
        clip_shard(-1, 0, 1)


def test_single_process_helpers_are_identity():
    t = torch.arange(6.0).reshape(3, 2)
    assert gather_features(t, 3) is t
    assert max_over_ranks(1.25) == 1.25


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n_clips: int, out_path: str):
    import oracle.spectral as o

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(99)  # every rank draws the same full batch
        clips = rng.standard_normal((n_clips, 8000)).astype(np.float32)
        a, b = clip_shard(n_clips, rank, world)
        # per-clip maximum: one oracle call per clip, as the reference's callers do
        mine = [o.compute_log_mel_spectrogram(clips[i], n_mels=80) for i in range(a, b)]
        local = torch.from_numpy(np.stack(mine)) if mine else torch.empty((0, 80, 51))
        slowest = max_over_ranks(1.0 + rank)
        assert slowest == float(world)
        full = gather_features(local, n_clips, dst=0)
        if rank == 0:
            want = np.stack([o.compute_log_mel_spectrogram(c, n_mels=80) for c in clips])
            np.testing.assert_array_equal(full.numpy(), want)
            open(out_path, "w").write("ok")
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


def _gather_worker(rank: int, world: int, port: int, n_clips: int, chunk: int, out_path: str):
    from mlx_audio_amd.sharding import FeatureGather

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = clip_shard(n_clips, rank, world)
        g = FeatureGather(n_clips, (3, 7), torch.float32, "cpu", dst=0, chunk_clips=chunk)
        for batch in range(3):  # the same gather object serves batch after batch
            full_ref = (torch.arange(n_clips * 21, dtype=torch.float32).reshape(n_clips, 3, 7)
                        + 1000.0 * batch)
            work = g.start(full_ref[a:b].clone())
            got = g.finish(work)
            if rank == 0:
                assert got.shape == (n_clips, 3, 7)
                assert torch.equal(got, full_ref), batch
            else:
                assert got is None and g.full is None  # only the destination holds the tensor
        with pytest.raises(ValueError):
            g.start(torch.zeros((b - a + 1, 3, 7)))
        if rank == 0:
            open(out_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_clips,chunk", [(2, 9, 2), (3, 10, 64), (3, 2, 1), (2, 1, 4)])
def test_chunked_gather_to_rank0_only(tmp_path, world, n_clips, chunk):
    """Ragged shards (the last rank may own fewer clips, or none), several chunks per peer,
    and repeated use; no rank but 0 ever allocates the full tensor."""
    out = tmp_path / "ok.txt"
    mp.spawn(_gather_worker, args=(world, _free_port(), n_clips, chunk, str(out)), nprocs=world,
             join=True)
    assert out.read_text() == "ok"


@pytest.mark.parametrize("n_clips", [4, 5])
def test_two_ranks_shard_and_gather(tmp_path, n_clips):
    out = tmp_path / "ok.txt"
    mp.spawn(_worker, args=(2, _free_port(), n_clips, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def test_bench_launcher_spawns_the_ranks_it_is_asked_for():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself and
    reports the world size the process group formed (gloo, no device work)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps",
                          "3", "--warmup", "1", "--workload", "none", "--backend", "gloo"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3
