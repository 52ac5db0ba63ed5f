"""N>1 path on CPU: world_size-2 gloo processes shard the image rows, render their band (the oracle stands in for
the HIP library, which needs a GPU) and gather to rank 0 with the same collective bench.py uses on RCCL."""
import os
import sys

import numpy as np
import pytest

from gatling_amd.dist import partition_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_rows_covers_image():
    for h in (1, 7, 270, 1080, 2160):
        for w in (1, 2, 3, 4, 8):
            bands = [partition_rows(h, w, r) for r in range(w)]
            assert bands[0][0] == 0 and bands[-1][1] == h
            assert all(bands[i][1] == bands[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in bands]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        partition_rows(10, 2, 2)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gatling_amd.dist import gather_rows, partition_rows
    from gatling_amd.scene import RenderSettings
    from gatling_amd.scenes import cornell_box
    from oracle import orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h = 40, 23  # odd height: bands differ by one row
    desc, rs = cornell_box(), RenderSettings(spp=2, max_bounces=4)
    r0, r1 = partition_rows(h, world, rank)
    tile, _ = orc.render(desc, rs, w, h, rows=(r0, r1))
    dist.barrier()
    full = gather_rows(torch.from_numpy(tile), h, w)
    # the row-interleaved shares bench.py uses (rows rank::world): here cut out of a whole-frame oracle render as a strided view
    whole, _ = orc.render(desc, rs, w, h)
    full_i = gather_rows(torch.from_numpy(whole)[rank::world], h, w, interleaved=True)
    if rank == 0:
        assert torch.equal(full_i, torch.from_numpy(whole))
        np.save(out_path, full.numpy())
    else:
        assert full is None and full_i is None
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_render(tmp_path, orc):
    import torch.multiprocessing as mp
    from gatling_amd.scene import RenderSettings
    from gatling_amd.scenes import cornell_box

    out = str(tmp_path / "full.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    ref, _ = orc.render(cornell_box(), RenderSettings(spp=2, max_bounces=4), 40, 23)
    got = np.load(out)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
