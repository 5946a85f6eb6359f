"""The XCD-aware tile order of the transposing kernels (`xcd_tile` / `xcd_tile_grid` in
mlx_audio_amd/csrc/mia_internal.h) and the block order of the whole-row FFT's super-passes,
restated in Python: every tile is visited exactly once, workgroups past the last tile leave, each
of the eight XCDs (workgroup index mod 8) owns one contiguous run of tiles with the neighbours
along the second tile axis adjacent in it, and no run is longer than ceil(tiles / 8)."""

import pytest


def xcd_tile_grid(gx, gy):
    return 8 * ((gx * gy + 7) // 8)


def xcd_tile(block, gx, gy):
    total = gx * gy
    per = (total + 7) // 8
    t = (block & 7) * per + (block >> 3)
    if t >= total:
        return None
    return t // gy, t % gy


@pytest.mark.parametrize("gx,gy", [(1, 1), (1, 7), (14, 33), (41, 65), (3, 3), (8, 8), (5, 1), (2, 100), (97, 13)])
def test_every_tile_once_and_contiguous_per_xcd(gx, gy):
    blocks = xcd_tile_grid(gx, gy)
    assert blocks % 8 == 0 and 0 <= blocks - gx * gy < 8 + 8   # at most one spare round of eight
    seen = {}
    per_xcd = {k: [] for k in range(8)}
    for b in range(blocks):
        tile = xcd_tile(b, gx, gy)
        if tile is None:
            continue
        assert tile not in seen
        seen[tile] = b
        per_xcd[b & 7].append(tile[0] * gy + tile[1])
    assert len(seen) == gx * gy
    sizes = []
    for k in range(8):
        run = per_xcd[k]
        assert run == list(range(run[0], run[0] + len(run))) if run else True   # contiguous, in order
        sizes.append(len(run))
    # full shares first, then the remainder, then nothing: no XCD carries more than ceil(tiles / 8)
    full = (gx * gy + 7) // 8
    q, r = divmod(gx * gy, full)
    assert sizes == ([full] * q + ([r] if r else []) + [0] * 8)[:8]


def super_block(block, blocks_u, per_xcd, row_slots):
    """stockham_super_kernel's mapping: (block of super-butterflies, first row) or None."""
    xcd, seq = block & 7, block >> 3
    row_slot = seq // per_xcd
    block_u = xcd * per_xcd + (seq - row_slot * per_xcd)
    if block_u >= blocks_u:
        return None
    return block_u, row_slot


@pytest.mark.parametrize("blocks_u,row_slots", [(46, 16), (57, 8), (1, 1), (8, 3), (9, 2), (1000, 1)])
def test_super_pass_block_order(blocks_u, row_slots):
    per_xcd = (blocks_u + 7) // 8
    grid = 8 * per_xcd * row_slots
    seen = set()
    for b in range(grid):
        m = super_block(b, blocks_u, per_xcd, row_slots)
        if m is None:
            continue
        assert m not in seen and m[1] < row_slots
        seen.add(m)
    assert len(seen) == blocks_u * row_slots
