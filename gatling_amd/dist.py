"""Multi-GPU sharding of the render loop: one process per GPU, scene replicated, image rows partitioned, tiles
gathered to rank 0 with one collective (RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests).

The reference has no multi-device path (it picks one Vulkan device, ``CgpuVk.cpp:892-909``); the partition follows
SURVEY.md section 8e: shard by pixels (contiguous row bands) so that each pixel's sample-order sum stays on one GPU
and an N-way split is bit-identical to the single-GPU image; RNG streams use the global pixel index.
"""
from __future__ import annotations

from typing import Optional, Tuple


def partition_rows(height: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous bands; the first ``height % world_size`` ranks get one extra row."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(height, world_size)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def interleaved_rows(height: int, world_size: int, rank: int) -> Tuple[int, int, int]:
    """(rowBegin, rowEnd, rowStride) of rank's share when rows are dealt round-robin: rank, rank + N, rank + 2N, ...

    Contiguous bands of a frame differ in cost (cornell 1080p, 8 bands: 0.61x .. 1.19x of the mean -- the slowest band sets
    the frame time); every rank's interleaved share samples the whole image, so the shares cost the same."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    return rank, height, world_size


def gather_rows(local_tile, height: int, width: int, group=None, dst: int = 0, interleaved: bool = False):
    """Gathers [rows_r, width, 4] float32 tensors (device or CPU) into the full [height, width, 4] image on ``dst``.

    The shares (contiguous bands, or rows rank::N when ``interleaved``) differ by at most one row; they are padded to a common
    size so a single gather moves everything (C5: 16.6 MB per GPU at 4K -- one collective, no all-reduce: SURVEY.md section 8e)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    max_rows = -(-height // world)
    pad = torch.zeros((max_rows, width, 4), dtype=local_tile.dtype, device=local_tile.device)
    pad[: local_tile.shape[0]] = local_tile  # also packs a strided (interleaved) device view
    out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, out, dst=dst, group=group)
    if rank != dst:
        return None
    full = torch.empty((height, width, 4), dtype=local_tile.dtype, device=local_tile.device)
    for r in range(world):
        if interleaved:
            n = len(range(r, height, world))
            full[r::world] = out[r][:n]
        else:
            r0, r1 = partition_rows(height, world, r)
            full[r0:r1] = out[r][: r1 - r0]
    return full
