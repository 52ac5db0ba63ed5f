"""Multi-GPU sharding of the render loop: one process per GPU, scene replicated, image rows partitioned, tiles
gathered to rank 0 with one collective (RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests).

The reference has no multi-device path (it picks one Vulkan device, ``CgpuVk.cpp:892-909``); the partition follows
SURVEY.md section 8e: shard by pixels (whole rows: rank r renders rows r, r+N, ... -- interleaved_rows; contiguous bands are kept
for comparison) so that each pixel's sample-order sum stays on one GPU
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


class RowGather:
    """One frame's gather of the per-rank row shares to ``dst``, with every buffer allocated ONCE (the padded send buffer, the
    receive buffer and the assembled frame on ``dst``): calling it costs a pack copy, one ``dist.gather`` and, on ``dst``, ONE
    strided copy that re-interleaves the shares -- no allocation inside a timed region.

    The shares (contiguous bands, or rows rank::N when ``interleaved``) differ by at most one row; they are padded to a common
    size so a single gather moves everything (C5: 16.6 MB per GPU at 4K -- one collective, no all-reduce: SURVEY.md section 8e)."""

    def __init__(self, height: int, width: int, dtype, device, group=None, dst: int = 0, interleaved: bool = False):
        import torch
        import torch.distributed as dist
        self.group, self.dst, self.interleaved, self.height, self.width = group, dst, interleaved, height, width
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.max_rows = -(-height // self.world)
        self.pad = torch.zeros((self.max_rows, width, 4), dtype=dtype, device=device)
        # ONE receive buffer [rank, row of the share, ...] whose slices are the gather's outputs, and the frame padded to max_rows * world rows: an interleaved
        # frame is then the receive buffer with its first two axes swapped -- one strided copy on `dst` instead of one per rank
        self.recv = torch.empty((self.world, self.max_rows, width, 4), dtype=dtype, device=device) if self.rank == dst else None
        self.out = [self.recv[r] for r in range(self.world)] if self.rank == dst else None
        self.full_padded = torch.empty((self.max_rows * self.world, width, 4), dtype=dtype, device=device) if self.rank == dst else None
        self.full = self.full_padded[:height] if self.rank == dst else None
        # the pack copy is the only reader of the caller's tile (the library's render buffer): `packed` marks its end on the caller's stream
        self.packed = torch.cuda.Event() if torch.device(device).type == "cuda" else None

    def wait_packed(self):
        """Blocks the host until the last call's pack copy has read the tile it was given: the producer of the next tile (the render library, which
        works on a stream of its own that nothing orders against torch's) may overwrite it only after this returns."""
        if self.packed is not None:
            self.packed.synchronize()

    def __call__(self, local_tile):
        import torch.distributed as dist
        self.pad[: local_tile.shape[0]].copy_(local_tile)  # also packs a strided (interleaved) device view
        if self.packed is not None:
            self.packed.record()
        dist.gather(self.pad, self.out, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        if self.interleaved:  # row k * world + r of the frame is row k of rank r's share (rows past `height` are padding on both sides)
            self.full_padded.view(self.max_rows, self.world, self.width, 4).copy_(self.recv.permute(1, 0, 2, 3))
            return self.full
        for r in range(self.world):
            r0, r1 = partition_rows(self.height, self.world, r)
            self.full[r0:r1].copy_(self.out[r][: r1 - r0])
        return self.full


def gather_rows(local_tile, height: int, width: int, group=None, dst: int = 0, interleaved: bool = False):
    """One-off form of RowGather (allocates its buffers per call): gathers [rows_r, width, 4] float32 tensors (device or CPU) into the
    full [height, width, 4] image on ``dst``; other ranks get None."""
    return RowGather(height, width, local_tile.dtype, local_tile.device, group, dst, interleaved)(local_tile)
