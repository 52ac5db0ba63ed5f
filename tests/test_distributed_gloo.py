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
    # the preallocated form bench.py keeps across steps: two frames through the same buffers
    from gatling_amd.dist import RowGather
    g = RowGather(h, w, torch.float32, torch.device("cpu"), interleaved=True)
    first = g(torch.from_numpy(whole)[rank::world])
    first = first.clone() if rank == 0 else None
    second = g(torch.from_numpy(whole * 2.0)[rank::world])
    if rank == 0:
        assert torch.equal(first, torch.from_numpy(whole)) and torch.equal(second, torch.from_numpy(whole * 2.0))
        assert torch.equal(full_i, torch.from_numpy(whole))
        np.save(out_path, full.numpy())
    else:
        assert full is None and full_i is None
    dist.destroy_process_group()


def _worker_ragged(rank, world, port, h, out_path):
    """Rows dealt round-robin over `world` ranks with a height that is NOT a multiple of it: the last deal is ragged (some ranks hold one row fewer and pad)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gatling_amd.dist import RowGather, interleaved_rows, partition_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = 9
    whole = torch.arange(h * w * 4, dtype=torch.float32).reshape(h, w, 4) * 0.25 + 1.0   # every texel distinct
    r0, r1, stride = interleaved_rows(h, world, rank)
    share = whole[r0:r1:stride]
    assert share.shape[0] == len(range(rank, h, world))
    g = RowGather(h, w, torch.float32, torch.device("cpu"), interleaved=True)
    for k in range(2):                                    # two frames through the same buffers (no stale padding rows leak into the frame)
        full = g(share * float(k + 1))
        if rank == 0:
            assert torch.equal(full, whole * float(k + 1)), f"world {world}, height {h}, frame {k}"
        else:
            assert full is None
    b0, b1 = partition_rows(h, world, rank)               # contiguous bands, same ragged height
    gb = RowGather(h, w, torch.float32, torch.device("cpu"), interleaved=False)
    fullb = gb(whole[b0:b1])
    if rank == 0:
        assert torch.equal(fullb, whole)
        open(out_path, "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h", [(4, 23), (8, 37), (8, 8), (4, 5)])
def test_interleaved_gather_ragged_last_row(tmp_path, world, h):
    """the 8-way interleave with a ragged last deal was only covered in single-process form.  Worlds of 4 and 8 over gloo, heights that are
    not multiples of the world (and one that is; and one barely larger than the world): interleaved and banded gathers reproduce the frame exactly."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.txt")
    port = 29500 + ((os.getpid() * 7 + world * 13 + h) % 2000)
    mp.spawn(_worker_ragged, args=(world, port, h, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


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


# ---- the same path on RCCL (needs a GPU) ---------------------------------------------------------------------------------------
def _run_bench(extra_env, args, nproc=1, timeout=600):
    import json
    import subprocess
    env = dict(os.environ, **extra_env)
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None)


@pytest.mark.gpu
def test_bench_distributed_path_on_rccl_one_rank(gi):
    """bench.py's N>1 code path -- strided device view of the library's render buffer, RowGather over RCCL ("nccl" backend), pinned D2H
    on rank 0 -- driven on ONE GPU with a world of one (GATLING_BENCH_FORCE_DIST), and its image checksum against the plain path."""
    args = ["--workload", "c1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc"]
    out_d, jd = _run_bench({"GATLING_BENCH_FORCE_DIST": "1", "GATLING_BENCH_CHECKSUM": "1", "GATLING_BENCH_ALSO": "c2", "GATLING_BENCH_ALSO_SPP": "4"}, args)
    out_s, js = _run_bench({"GATLING_BENCH_CHECKSUM": "1"}, args)
    assert out_d.returncode == 0 and jd is not None, out_d.stdout[-2000:] + out_d.stderr[-2000:]
    assert out_s.returncode == 0 and js is not None, out_s.stdout[-2000:] + out_s.stderr[-2000:]
    assert jd["n_gpus"] == 1 and jd["value"] > 0 and jd["image_checksum"] == js["image_checksum"]
    assert isinstance(jd["also"], list) and len(jd["also"]) == 1
    assert jd["also"][0].get("value", 0) > 0 and "error" not in jd["also"][0], jd["also"]  # the second-workload leg bench.py adds at N = 8 (C5)


@pytest.mark.gpu
def test_bench_two_ranks_sharing_one_gpu(gi):
    """Two ranks (torch.distributed.run) that share GPU 0, rows dealt 0::2 / 1::2, gathered over RCCL.  RCCL may refuse two ranks on one
    device ("Duplicate GPU detected"): that refusal is the one accepted failure, anything else -- a hang, a wrong image -- is not."""
    args = ["--workload", "c1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-pmc"]
    out_s, js = _run_bench({"GATLING_BENCH_CHECKSUM": "1"}, args)
    out, j = _run_bench({"GATLING_BENCH_SHARE_GPU": "1", "GATLING_BENCH_CHECKSUM": "1"}, args, nproc=2, timeout=300)
    text = out.stdout + out.stderr
    if out.returncode != 0 and ("Duplicate GPU" in text or "invalid usage" in text.lower() or "ncclInvalidUsage" in text):
        pytest.skip("RCCL refuses two ranks on one GPU (duplicate-GPU check); the 2-rank path is covered on gloo and at N=1 on RCCL")
    assert out.returncode == 0 and j is not None, text[-3000:]
    assert j["n_gpus"] == 2 and j["image_checksum"] == js["image_checksum"]
