"""The configurations the bench numbers are quoted on, AT THEIR SPP, against the oracle.

The device renders exactly what `bench.py` times -- the whole frame, the default code path, the default batch / pool sizes -- and the oracle renders a band of
rows of the same frame (a band of 16 rows at full spp costs the oracle seconds; the whole C2 frame ~3 minutes on the GPU box's 256 cores: that one runs on hosts
with >= 128 cores or when $GATLING_SLOW_TESTS is set).  Bar: bit-identical pixels (the per-pixel sample sum is taken in sample order,
rp_main.rgen:215, 498).

  C2  cornell 1920x1080, spp 1024, 8 bounces: ONE 34 GB batch of the fused kernel, work ids up to 2.12e9 in 32 bits
  C3  1 M-triangle soup, NEE, spp 256 / C4  1 024 instanced icospheres, 32 materials, spp 256: the wavefront pipeline over the 64 Mi-slot pool
  C5  4K interior, spp 1024: rank 3's interleaved share of the 8-GPU partition, sample-buffer budget forced so that THREE batches occur
"""
import os

import numpy as np
import pytest

from gatling_amd.scene import RenderSettings
from gatling_amd.scenes import cornell_box, interior_scene, random_triangle_soup, sphere_grid

pytestmark = pytest.mark.gpu
_CORES = os.cpu_count() or 4
# The two long checks (whole C2 frame through the oracle: 2.1 G samples, ~3 min on 256 cores; 67.7 M flattened triangles: ~10 GB of host arrays, ~2 min) run in the
# DEFAULT suite on a host with >= 128 cores -- the driver's GPU box has 256 -- so that they are the driver's evidence, not only the builder's (VERDICT r05 next #3a);
# $GATLING_SLOW_TESTS=1 forces them anywhere, =0 switches them off.
_SLOW = os.environ.get("GATLING_SLOW_TESTS")
_RUN_SLOW = (_SLOW not in (None, "", "0")) or (_SLOW is None and _CORES >= 128)


def _bands_equal(full, desc, rs, w, h, orc, bands):
    for r0, r1 in bands:
        ref, cnt = orc.render(desc, rs, w, h, rows=(r0, r1), threads=_CORES)
        assert cnt["samples"] == (r1 - r0) * w * rs.spp
        bad = int((full[r0:r1].view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
        assert bad == 0, f"rows {r0}-{r1}: {bad} pixels differ bitwise from the oracle at spp {rs.spp}"


def test_c2_headline_spp1024_one_batch_bands_bit_exact(gi, orc):
    """The headline configuration itself: C2 at spp 1024 in one batch (the driver-timed line).  Rows 532-547 (the middle of the frame) and the last 8 rows
    (the highest work ids: sample-major ids reach 1024 * 2 073 600 - 1 = 2.12e9) against the oracle."""
    desc, rs, w, h = cornell_box(), RenderSettings(spp=1024, max_bounces=8, progressive_accumulation=False), 1920, 1080
    sc = gi.Scene(desc)
    try:
        full = sc.render(rs, w, h).copy()
        st = sc.stats()
    finally:
        sc.close()
    assert st["samples"] == w * h * 1024 and st["fusedPath"] == 1 and st["iterations"] == 1, st   # one batch of the fused kernel, as bench.py runs it
    _bands_equal(full, desc, rs, w, h, orc, [(532, 548), (1072, 1080), (0, 4)])
    assert np.isfinite(full).all() and (full[..., 3] == 1.0).all()


@pytest.mark.skipif(not _RUN_SLOW, reason="the whole C2 frame at spp 1024 through the oracle: ~3 min on 256 cores; runs where os.cpu_count() >= 128 or with GATLING_SLOW_TESTS=1")
def test_c2_headline_spp1024_whole_frame_bit_exact(gi, orc):
    desc, rs, w, h = cornell_box(), RenderSettings(spp=1024, max_bounces=8, progressive_accumulation=False), 1920, 1080
    sc = gi.Scene(desc)
    try:
        full = sc.render(rs, w, h).copy()
        st = sc.stats()
    finally:
        sc.close()
    ref, cnt = orc.render(desc, rs, w, h, threads=_CORES)
    bad = int((full.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
    print(f"C2 1920x1080 spp 1024 whole frame: {bad} of {w * h} pixels differ bitwise; segments device {st['segments']} oracle {cnt['segments']}")
    assert bad == 0 and st["segments"] == cnt["segments"]


@pytest.mark.parametrize("name", ["c3", "c4"])
def test_c3_c4_spp256_band_bit_exact(gi, orc, name):
    """C3 / C4 as bench.py's `also` legs run them (whole 1080p frame, spp 256, default pool): a 16-row band in the middle and 4 rows at the top edge."""
    if name == "c3":
        desc, rs = random_triangle_soup(1_000_000), RenderSettings(spp=256, max_bounces=8, next_event_estimation=True, progressive_accumulation=False)
    else:
        desc, rs = sphere_grid(32, 4, 32), RenderSettings(spp=256, max_bounces=8, progressive_accumulation=False)
    w, h = 1920, 1080
    sc = gi.Scene(desc)
    try:
        full = sc.render(rs, w, h).copy()
        st = sc.stats()
    finally:
        sc.close()
    assert st["samples"] == w * h * 256 and st["fusedPath"] == 0
    _bands_equal(full, desc, rs, w, h, orc, [(532, 548), (1076, 1080)])


def test_c5_spp1024_rank_share_three_batches_bit_exact(gi, orc):
    """C5 at its spp: rows 3, 11, 19, ... (rank 3 of the 8-GPU partition) of the 4K interior at spp 1024, with the per-sample buffer capped at 6 GiB so the
    share's 17 GB of samples take THREE batches (the whole 4K frame on one GPU takes three with the default budget); two rows of the share against the oracle."""
    desc = interior_scene()
    rs = RenderSettings(spp=1024, max_bounces=8, next_event_estimation=True, progressive_accumulation=False)
    w, h = 3840, 2160
    sc = gi.Scene(desc)
    try:
        sc.set_option(gi.OPTION_SAMPLE_BUFFER_MB, 6144)
        share = sc.render(rs, w, h, rows=(3, h), row_stride=8).copy()
        st = sc.stats()
    finally:
        sc.close()
    assert share.shape[0] == 270 and st["samples"] == 270 * w * 1024
    batch = (6144 << 20) // (270 * w * 16)
    assert -(-1024 // batch) == 3, batch
    ks = [100, 269]                            # image rows 803 and 2155 (the share's last row)
    ref, cnt = orc.render(desc, rs, w, h, row_list=[3 + 8 * k for k in ks], threads=_CORES)
    assert cnt["samples"] == 2 * w * 1024
    for i, k in enumerate(ks):
        bad = int((share[k].view(np.uint32) != ref[i].view(np.uint32)).any(axis=-1).sum())
        assert bad == 0, f"image row {3 + 8 * k}: {bad} pixels differ bitwise from the oracle at spp 1024 over three batches"


@pytest.mark.skipif(not _RUN_SLOW, reason="67.7 M flattened triangles: ~10 GB of host arrays on both sides, ~2 min; runs where os.cpu_count() >= 128 or with GATLING_SLOW_TESTS=1")
def test_scene_beyond_2_pow_26_flattened_triangles_takes_the_two_level_layout(gi, orc):
    """VERDICT r03 missing #6: the flat traversal's triangle ring packs (lane, triangle) into 32 bits -- 2^26 triangles -- and such scenes used to be refused.  The
    two-level walk queues MESH triangles, so a heavily instanced scene beyond the bound takes it automatically: 115 x 115 instances of the 5 120-triangle icosphere =
    67.7 M flattened triangles, image and segment counts against the oracle's own traversal."""
    desc = sphere_grid(115, 4, 8)
    assert sum(len(m.faces) * len(m.instance_transforms) for m in desc.meshes) >= 1 << 26
    rs = RenderSettings(spp=2, max_bounces=6, progressive_accumulation=False)
    w, h = 192, 108
    sc = gi.Scene(desc)
    try:
        img = sc.render(rs, w, h).copy()
        st = sc.stats()
    finally:
        sc.close()
    ref, cnt = orc.render(desc, rs, w, h, threads=_CORES)
    print(f"67.7 M flattened triangles: triangleCount {st['triangleCount']}, nodes {st['nodeCount']}, segments device {st['segments']} oracle {cnt['segments']}")
    assert st["triangleCount"] == 115 * 115 * 5120 and st["segments"] == cnt["segments"]
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
