"""giCRender's memory plan: the per-sample colour buffer and the path pool are sized from what is FREE on the device, and an
allocation that fails anyway is answered with a smaller plan (more batches, then a smaller pool) -- the image does not depend on the plan.

The GPU box has one 288 GB device to itself; the tests take the memory away with a torch allocation (torch and the library share the HIP runtime)."""
import gc

import numpy as np
import pytest

try:  # torch before the HIP library (two HIP runtimes in one process otherwise)
    import torch
except Exception:  # pragma: no cover
    torch = None

from gatling_amd.scene import RenderSettings
from gatling_amd.scenes import cornell_box, sphere_grid

pytestmark = pytest.mark.gpu


def _hog(leave_mb):
    """Allocates everything but `leave_mb` MiB of the device; returns the tensors (drop them to give the memory back)."""
    torch.cuda.synchronize()
    free, _ = torch.cuda.mem_get_info()
    want = free - (leave_mb << 20)
    chunks = []
    while want > (64 << 20):  # several chunks: one 280 GB request may exceed an allocator limit
        n = min(want, 64 << 30)
        chunks.append(torch.empty(n, dtype=torch.uint8, device="cuda"))
        want -= n
    return chunks


@pytest.mark.skipif(torch is None or not torch.cuda.is_available(), reason="needs torch on the GPU")
def test_render_on_a_nearly_full_device_equals_the_roomy_render(gi):
    """Two scenes alive at once, each with its default plan; then all but 1.5 GiB of the device is taken away and a THIRD and FOURTH scene render: the wavefront
    pipeline with a pool a fraction of the default's, the fused kernel with its frame cut into batches -- bit-identical images, and the first scenes still render."""
    cases = [(sphere_grid(8, 2, 8), RenderSettings(spp=48, max_bounces=6, progressive_accumulation=False), 640, 360),      # wavefront pipeline: 11 M work items
             (cornell_box(), RenderSettings(spp=192, max_bounces=8, progressive_accumulation=False), 1280, 720)]           # fused kernel: 2.8 GB of samples
    roomy, scenes = [], []
    for desc, rs, w, h in cases:
        sc = gi.Scene(desc); scenes.append(sc)
        roomy.append((sc.render(rs, w, h).copy(), sc.stats()))
    assert roomy[0][1]["poolSlots"] == 640 * 360 * 48 and roomy[0][1]["batches"] == 1 and roomy[1][1]["batches"] == 1, [r[1] for r in roomy]
    hog = _hog(1536)
    try:
        for (desc, rs, w, h), (ref, rst) in zip(cases, roomy):
            sc = gi.Scene(desc)
            try:
                img = sc.render(rs, w, h).copy()
                st = sc.stats()
            finally:
                sc.close()
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "the image depends on the memory plan"
            assert st["segments"] == rst["segments"]
            if rst["fusedPath"]:
                assert st["batches"] > 1, st          # 2.8 GB of samples do not fit 1.5 GiB: more batches
            else:
                assert 0 < st["poolSlots"] < rst["poolSlots"], (st, rst)   # 11 M slots with queues want ~3 GB: a smaller pool
        for sc, (desc, rs, w, h), (ref, _) in zip(scenes, cases, roomy):  # the first two scenes keep what they hold and still render
            assert np.array_equal(sc.render(rs, w, h).view(np.uint32), ref.view(np.uint32))
    finally:
        del hog
        gc.collect(); torch.cuda.empty_cache()
        for sc in scenes:
            sc.close()


@pytest.mark.skipif(torch is None or not torch.cuda.is_available(), reason="needs torch on the GPU")
def test_failed_allocation_falls_back_to_a_smaller_plan(gi, monkeypatch):
    """The planner is told the whole device is free while all but 2 GiB of it is taken (someone else allocated between the query and the hipMalloc): the first
    plan's allocation fails with hipErrorOutOfMemory and the render goes on with smaller ones instead of failing."""
    desc, rs, w, h = sphere_grid(8, 2, 8), RenderSettings(spp=64, max_bounces=6, progressive_accumulation=False), 800, 450   # 23 M work items: ~6 GB plan
    sc = gi.Scene(desc)
    try:
        ref = sc.render(rs, w, h).copy(); rst = sc.stats()
    finally:
        sc.close()
    hog = _hog(2048)
    try:
        monkeypatch.setenv("GATLING_OPTIONS", f"assume_free_mb={280 * 1024}")
        sc = gi.Scene(desc)
        try:
            img = sc.render(rs, w, h).copy(); st = sc.stats()
        finally:
            sc.close()
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        assert 0 < st["poolSlots"] < rst["poolSlots"], (st, rst)
    finally:
        del hog
        gc.collect(); torch.cuda.empty_cache()
