"""GPU parity tests (-m gpu): the HIP path through the C ABI vs the CPU oracle on the same seeded inputs.

Bar: bit-exact for integer work (segment / shadow-ray counts, hit ids).  For the float image the stated tolerance
is SURVEY.md section 8d(i) -- |d| <= 1e-4 + 1e-3*|ref| for >= 99.9 % of pixels, the rest <= 0.05, mean error <= 1e-5
(`assert_image_parity(exact=False)`) -- but because kernels and oracle follow the same arithmetic contract
(fp32, no contraction, IEEE + - * / sqrt, polynomial sincos/log) every test here demands the stronger
bit-identical image (`exact=True`)."""
import os
import sys

import numpy as np
import pytest

try:  # torch before the HIP library: loaded the other way round, torch does not see the GPU (two HIP runtimes in one process)
    import torch
    _TORCH_CUDA = torch.cuda.is_available()
except Exception:  # pragma: no cover
    torch, _TORCH_CUDA = None, False

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden import CASES, build_case  # noqa: E402

from gatling_amd.scene import (MAT_DIFFUSE, MAT_OPEN_PBR, MAT_USD_PREVIEW_SURFACE, CameraDesc, DiskLight, DistantLight, MaterialDesc, MeshDesc,
                               RectLight, RenderSettings, SceneDesc, SphereLight)
from gatling_amd.scenes import cornell_box, interior_scene, random_triangle_soup, sphere_grid, textured_scene, volume_scene

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def assert_image_parity(img, ref, exact=False):
    assert img.shape == ref.shape and np.isfinite(img).all()
    if exact:
        bad = int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
        assert bad == 0, f"{bad} pixels differ bitwise"
        return
    d = np.abs(img - ref)
    within = (d <= 1e-4 + 1e-3 * np.abs(ref)).all(axis=-1)
    assert within.mean() >= 0.999, f"only {within.mean():.5f} of pixels within tolerance"
    assert d.max() <= 0.05
    assert abs(float(img.mean()) - float(ref.mean())) <= 1e-5


def render_both(gi, orc, desc, rs, w, h, exact=True, threads=4):
    sc = gi.Scene(desc)
    try:
        img = sc.render(rs, w, h)
        st = sc.stats()
    finally:
        sc.close()
    ref, cnt = orc.render(desc, rs, w, h, threads=threads)
    assert st["segments"] == cnt["segments"] and st["shadowRays"] == cnt["shadow_rays"] and st["samples"] == cnt["samples"]
    assert_image_parity(img, ref, exact)
    return img, ref, st


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_fixtures(gi, name):
    """Committed fixtures (tests/golden/make_golden.py): no oracle call needed on the GPU box for these."""
    desc, rs, w, h = build_case(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    sc = gi.Scene(desc)
    try:
        img = sc.render(rs, w, h)
        st = sc.stats()
    finally:
        sc.close()
    assert st["segments"] == int(g["segments"]) and st["shadowRays"] == int(g["shadow_rays"])
    assert_image_parity(img, g["color"], exact=True)


def test_c1_diffuse_bit_exact(gi, orc):
    """Config C1 (cornell, diffuse-only model, 4 bounces) at a size the oracle finishes in seconds."""
    render_both(gi, orc, cornell_box(MAT_DIFFUSE), RenderSettings(spp=8, max_bounces=4), 160, 90, exact=True)


def test_c2_model_parity(gi, orc):
    """Config C2's model (UsdPreviewSurface, 8 bounces) on a reduced image."""
    render_both(gi, orc, cornell_box(MAT_USD_PREVIEW_SURFACE), RenderSettings(spp=8, max_bounces=8), 160, 90)


SETTING_CASES = {
    "no_jitter": dict(jittered_sampling=False),
    "uniform_jitter": dict(filter_importance_sampling=False),
    "dof": dict(depth_of_field=True),
    "clip": dict(clipping_planes=True),
    "rr_early": dict(rr_bounce_offset=0, rr_inv_min_term_prob=0.5),
    "one_bounce": dict(max_bounces=1),
    "clamp_low": dict(max_sample_value=0.5),
    "black_bg": dict(clear_color=(0.0, 0.0, 0.0, 0.0)),
}


@pytest.mark.parametrize("case", sorted(SETTING_CASES))
def test_render_settings_matrix(gi, orc, case):
    kw = dict(spp=4, max_bounces=6)
    kw.update(SETTING_CASES[case])
    desc = cornell_box(MAT_DIFFUSE)
    if case == "dof":
        desc.camera.f_stop, desc.camera.focus_distance = 0.5, 6.5
    if case == "clip":
        desc.camera.clip_start, desc.camera.clip_end = 6.5, 7.8  # cuts through the box
    render_both(gi, orc, desc, RenderSettings(**kw), 96, 54, exact=True)


LIGHTS = {
    "sphere": dict(sphere_lights=[SphereLight(pos=(0.3, -0.2, 0.4), base_emission=(6, 5, 4), radius=(0.15, 0.1, 0.2))]),
    "distant": dict(distant_lights=[DistantLight(direction=(0.2, 0.5, -0.8), base_emission=(2, 2, 2), angle=0.1)]),
    "distant_sharp": dict(distant_lights=[DistantLight(direction=(0.0, 0.6, -0.8), base_emission=(1.5, 1.5, 1.5), angle=0.0)]),
    "rect": dict(rect_lights=[RectLight(origin=(0, 0, 0.9), t0=(1, 0, 0), t1=(0, -1, 0), base_emission=(10, 10, 10), width=0.7, height=0.5)]),
    "disk": dict(disk_lights=[DiskLight(origin=(0, 0, 0.9), t0=(1, 0, 0), t1=(0, -1, 0), base_emission=(10, 8, 6), radius_x=0.3, radius_y=0.2)]),
    "all": dict(sphere_lights=[SphereLight(pos=(0.3, -0.2, 0.4), base_emission=(6, 5, 4), radius=(0.1, 0.1, 0.1), diffuse=0.5, specular=2.0)],
                distant_lights=[DistantLight(direction=(0.2, 0.5, -0.8), base_emission=(1, 1, 1), angle=0.05)],
                rect_lights=[RectLight(origin=(0, 0, 0.9), t0=(1, 0, 0), t1=(0, -1, 0), base_emission=(10, 10, 10), width=0.7, height=0.5)],
                disk_lights=[DiskLight(origin=(-0.5, 0.5, 0.9), t0=(1, 0, 0), t1=(0, -1, 0), base_emission=(10, 8, 6), radius_x=0.3, radius_y=0.2)]),
    "none": dict(),  # zero lights with NEE on: reference reads slot 0 of a zero-filled store
}


@pytest.mark.parametrize("klass", [MAT_DIFFUSE, MAT_USD_PREVIEW_SURFACE])
@pytest.mark.parametrize("lights", sorted(LIGHTS))
def test_next_event_estimation(gi, orc, lights, klass):
    desc = cornell_box(klass)
    for k, v in LIGHTS[lights].items():
        setattr(desc, k, v)
    rs = RenderSettings(spp=4, max_bounces=5, next_event_estimation=True, light_intensity_multiplier=1.5)
    desc.camera.exposure = 0.5
    render_both(gi, orc, desc, rs, 96, 54)


def test_progressive_accumulation_across_calls(gi, orc):
    """scene->sampleOffset advances by spp per giRender and resets on any change (Gi.cpp:2125-2129, 2515)."""
    desc = cornell_box(MAT_DIFFUSE)
    rs = RenderSettings(spp=3, max_bounces=4)
    sc = gi.Scene(desc)
    try:
        first = sc.render(rs, 64, 36)
        second = sc.render(rs, 64, 36)
        third = sc.render(RenderSettings(spp=3, max_bounces=5), 64, 36)  # settings change -> offset back to 0
    finally:
        sc.close()
    r1, _ = orc.render(desc, rs, 64, 36, sample_offset=0)
    r2, _ = orc.render(desc, rs, 64, 36, sample_offset=3, prev_color=r1)
    r3, _ = orc.render(desc, RenderSettings(spp=3, max_bounces=5), 64, 36, sample_offset=0)
    assert_image_parity(first, r1, exact=True)
    assert_image_parity(second, r2, exact=True)
    assert_image_parity(third, r3, exact=True)


def test_row_sharding_is_bit_identical(gi):
    """The multi-GPU partition (rowBegin/rowEnd) must reproduce the single-call image exactly."""
    desc = cornell_box()
    rs = RenderSettings(spp=4, max_bounces=6, progressive_accumulation=False)
    sc = gi.Scene(desc)
    try:
        full = sc.render(rs, 80, 45)
        parts = [sc.render(rs, 80, 45, rows=r) for r in ((0, 12), (12, 23), (23, 45))]
    finally:
        sc.close()
    assert np.array_equal(np.concatenate(parts).view(np.uint32), full.view(np.uint32))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_interleaved_row_sharding_is_bit_identical(gi, world):
    """What bench.py --gpus N does: rank r renders rows r, r+N, ... (rowStride = N; balanced load) -- every share equals the same
    rows of the single-call image, colour and AOVs alike, also for heights that N does not divide."""
    from gatling_amd.dist import interleaved_rows
    desc = cornell_box()
    desc.rect_lights = [RectLight(origin=(0, 0, 0.9), t0=(1, 0, 0), t1=(0, -1, 0), base_emission=(10, 10, 10), width=0.7, height=0.5)]
    rs = RenderSettings(spp=3, max_bounces=6, next_event_estimation=True, progressive_accumulation=False)
    w, h = 64, 37
    sc = gi.Scene(desc)
    try:
        full = sc.render(rs, w, h)
        aov_full = sc.render_aovs(rs, w, h, ["normal", "objectId", "nee", "bounces"])
        for rank in range(world):
            r0, r1, stride = interleaved_rows(h, world, rank)
            part = sc.render(rs, w, h, rows=(r0, r1), row_stride=stride)
            assert part.shape[0] == len(range(rank, h, world))
            assert np.array_equal(part.view(np.uint32), full[rank::world].view(np.uint32)), rank
        sc.set_option(gi.OPTION_POOL_SLOTS, 512)  # a pool much smaller than the share: slots walk many rows
        part = sc.render(rs, w, h, rows=(1, h), row_stride=world)
        assert np.array_equal(part, full[1::world])
        aov_part = sc.render_aovs(rs, w, h, ["normal", "objectId", "nee", "bounces"], rows=(1, h), row_stride=world)
        for k in ("normal", "objectId", "nee", "bounces", "color"):
            assert np.array_equal(aov_part[k], aov_full[k][1::world]), k
    finally:
        sc.close()


def test_interleaved_share_as_strided_device_tensor(gi):
    """bench.py hands RCCL a zero-copy torch view of the library's device render buffer; for interleaved shares that view is
    strided (rows rank::N).  Checks the view (what gather_rows packs and sends) against the host copy of the same render."""
    if not _TORCH_CUDA:
        pytest.skip("needs torch on a GPU")
    desc = cornell_box()
    rs = RenderSettings(spp=2, max_bounces=4, progressive_accumulation=False)
    w, h, world, rank = 48, 29, 8, 3
    sc = gi.Scene(desc)
    try:
        host = sc.render(rs, w, h, rows=(rank, h), row_stride=world)
        ptr = sc.device_pointer(w, h)
        n = len(range(rank, h, world))

        class _Tile:
            __cuda_array_interface__ = {"shape": (n, w, 4), "typestr": "<f4", "data": (ptr + rank * w * 16, False), "version": 2,
                                        "strides": (world * w * 16, 16, 4)}
        dev = torch.as_tensor(_Tile(), device="cuda:0")
        pad = torch.zeros((-(-h // world), w, 4), dtype=dev.dtype, device=dev.device)
        pad[:n] = dev  # the packing step of gather_rows
        assert np.array_equal(pad[:n].cpu().numpy(), host)
    finally:
        sc.close()


def test_pool_and_batch_invariance(gi, orc):
    """Scheduling knobs must not change a bit: a 64-slot pool (every slot carries hundreds of work items in turn), a
    pool larger than the work, and a 1 MiB sample buffer (several sample batches per frame, each with its own drain +
    in-order accumulate) all reproduce the oracle image exactly."""
    desc = cornell_box()
    rs = RenderSettings(spp=24, max_bounces=6)
    w, h = 160, 90
    ref, cnt = orc.render(desc, rs, w, h, threads=4)
    for pool, mb, fused in ((64, 0, 0), (1000, 1, 0), (0, 1, 0), (1 << 22, 0, 0), (0, 1, -1)):
        sc = gi.Scene(desc)
        try:
            sc.set_option(gi.OPTION_FUSED_PATH, fused)  # 0: the stage kernels (cornell would otherwise run the fused kernel, which has no pool)
            sc.set_option(gi.OPTION_POOL_SLOTS, pool)
            sc.set_option(gi.OPTION_SAMPLE_BUFFER_MB, mb)
            img = sc.render(rs, w, h)
            st = sc.stats()
        finally:
            sc.close()
        assert st["segments"] == cnt["segments"]
        assert_image_parity(img, ref, exact=True)


@pytest.mark.parametrize("w,h,spp,mb", [(96, 54, 5, 0), (37, 21, 3, 0), (100, 7, 9, 0), (8, 8, 2, 0), (155, 90, 7, 1)])
def test_work_order_invariance(gi, orc, monkeypatch, w, h, spp, mb):
    """The wavefront pipeline may hand its work items out sample-major or pixel-major (8x8 pixel blocks, a pixel's samples consecutive; gi_queues.h
    work_item), with the per-sample buffer laid out to match: any bijection work item -> (pixel, sample) gives the oracle's image.  Widths and heights that
    are no multiples of 8 exercise the ragged blocks; the 1 MiB sample buffer splits the frame into several batches."""
    for desc, nee in ((cornell_box(), False), (_soup(3000), True)):
        rs = RenderSettings(spp=spp, max_bounces=4, next_event_estimation=nee)
        ref, cnt = orc.render(desc, rs, w, h, threads=4)
        for order in ("0", "1"):
            monkeypatch.setenv("GATLING_OPTIONS", f"work_order={order}")
            sc = gi.Scene(desc)
            try:
                sc.set_option(gi.OPTION_FUSED_PATH, 0)
                sc.set_option(gi.OPTION_POOL_SLOTS, 777)
                sc.set_option(gi.OPTION_SAMPLE_BUFFER_MB, mb)
                img = sc.render(rs, w, h)
                st = sc.stats()
            finally:
                sc.close()
            assert st["fusedPath"] == 0 and st["segments"] == cnt["segments"], (order, st)
            assert_image_parity(img, ref, exact=True)


@pytest.mark.parametrize("name", sorted(CASES))
def test_deferred_slot_initialisation_is_invisible(gi, monkeypatch, name):
    """FLAG_DEFER_SLOT (r04): k_raygen no longer writes a new path's Slot; its rng / work item travel beside the camera ray and the slot is written where the first
    segment hits -- a camera ray that leaves the scene retires in k_route / k_trace without one.  Every golden fixture (open cornell box, dome image, medium stacks,
    instanced spheres under a constant background, cutout cards, interior with NEE) through the wavefront stage kernels with the deferral on (default), off, with
    a small pool (slots recycled many times: stale slot contents must never be read) and with the block-synchronous k_trace: the committed image, bit for bit."""
    desc, rs, w, h = build_case(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    for defer, pool, dyn in (("1", 0, -1), ("0", 0, -1), ("1", 301, -1), ("1", 0, 0), ("0", 301, 0), ("1", 64, -2)):
        monkeypatch.setenv("GATLING_OPTIONS", f"defer_slot={defer},bounds_retire={0 if dyn == -2 else 1}")  # (r04n: camera rays that miss the scene's bounds retire in k_raygen; -2 = default traversal without it)
        dyn = -1 if dyn == -2 else dyn
        sc = gi.Scene(desc)
        try:
            sc.set_option(gi.OPTION_FUSED_PATH, 0)
            if pool:
                sc.set_option(gi.OPTION_POOL_SLOTS, pool)
            sc.set_option(gi.OPTION_TRACE_DYNAMIC, dyn)
            img = sc.render(rs, w, h)
            st = sc.stats()
            import copy
            rs2 = copy.copy(rs); rs2.progressive_accumulation = False
            again = sc.render(rs2, w, h)  # (a second frame from sample 0 over the same pool: every slot holds a finished path's leftovers)
        finally:
            sc.close()
        assert st["fusedPath"] == 0 and st["segments"] == int(g["segments"]) and st["shadowRays"] == int(g["shadow_rays"]), (defer, pool, dyn)
        assert_image_parity(img, g["color"], exact=True)
        assert_image_parity(again, g["color"], exact=True)


def test_camera_rays_that_cannot_reach_the_scene_retire_in_raygen(gi, orc, monkeypatch):
    """FLAG_BOUNDS_RETIRE (r04n): on the k_trace_dyn path k_raygen itself retires a deferred-slot camera ray whose slab interval against the scene's bounds is empty
    (same sample, same segment count; no ray record, traversal step or routing pass) and hands the slot to the next k_raygen.  Cameras that make the test bite in
    every way: the frame wider than the scene (most rays miss the bounds), looking away from it (ALL rays miss: whole iterations queue no ray at all and the loop
    must still hand out every work item), inside the bounds (none miss), an axis-parallel view (direction components exactly 0), far away (the slab arithmetic at
    1e4 x the scene's size), thin-lens depth of field (origins off the eye point) and clipping planes that end before the scene begins; each with the default pool
    and a 301-slot pool (slots recycled hundreds of times), on and off: device == oracle bit for bit, same segment counts."""
    import copy
    from gatling_amd.scenes import _look_at_camera
    base = random_triangle_soup(400_000, seed=77)
    cams = {
        "wide": _look_at_camera((0, -6, 0.3), (0, 0, 0), (0, 0, 1), 70.0),
        "away": _look_at_camera((0, -4, 0), (0, -9, 0.5), (0, 0, 1), 40.0),
        "inside": _look_at_camera((0.1, 0.2, -0.1), (1, 1, 0.2), (0, 0, 1), 60.0),
        "axis": _look_at_camera((0, -4, 0), (0, 0, 0), (0, 0, 1), 40.0),
        "far": _look_at_camera((3000, -20000, 900), (0, 0, 0), (0, 0, 1), 0.02),
    }
    cams["dof"] = copy.copy(cams["wide"]); cams["dof"].f_stop = 1.4; cams["dof"].focus_distance = 6.0; cams["dof"].focal_length = 0.6  # lens radius 0.21
    cams["clipped"] = copy.copy(cams["axis"]); cams["clipped"].clip_start = 0.1; cams["clipped"].clip_end = 2.5
    w, h = 96, 54
    for name, cam in cams.items():
        rs = RenderSettings(spp=3, max_bounces=5, next_event_estimation=True, progressive_accumulation=False, depth_of_field=name == "dof", clipping_planes=name == "clipped")
        desc = copy.copy(base); desc.camera = cam
        ref, cnt = orc.render(desc, rs, w, h, threads=4)
        for retire, pool in (("1", 0), ("1", 301), ("0", 0)):
            monkeypatch.setenv("GATLING_OPTIONS", f"bounds_retire={retire}")
            sc = gi.Scene(desc)
            try:
                if pool:
                    sc.set_option(gi.OPTION_POOL_SLOTS, pool)
                img = sc.render(rs, w, h)
                st = sc.stats()
                again = sc.render(rs, w, h)
            finally:
                sc.close()
            assert st["fusedPath"] == 0 and st["segments"] == cnt["segments"] and st["shadowRays"] == cnt["shadow_rays"], (name, retire, pool, st["segments"], cnt["segments"])
            assert_image_parity(img, ref, exact=True)
            assert_image_parity(again, ref, exact=True)
        if name in ("away", "clipped"):
            assert cnt["segments"] == w * h * rs.spp  # every path is its camera ray
    # several batches per frame (sample buffer capped at 1 MiB: four of them) over a 4 099-slot pool, on the instanced spheres under a constant background (C4's shape:
    # most camera rays pass beside the grid)
    desc = sphere_grid(grid=6, subdivisions=2, material_count=5)
    rs = RenderSettings(spp=24, max_bounces=5)
    w, h = 128, 72
    ref1, cnt = orc.render(desc, rs, w, h, threads=4)
    sc = gi.Scene(desc)
    try:
        sc.set_option(gi.OPTION_SAMPLE_BUFFER_MB, 1)
        sc.set_option(gi.OPTION_POOL_SLOTS, 4099)
        sc.set_option(gi.OPTION_FUSED_PATH, 0)
        img = sc.render(rs, w, h)
        st = sc.stats()
    finally:
        sc.close()
    assert st["fusedPath"] == 0 and st["batches"] >= 3 and st["segments"] == cnt["segments"], st
    assert_image_parity(img, ref1, exact=True)
    # retired samples go through the same running sum and progressive blend as any other: a second progressive frame, the sample-major work order, and rank 1's
    # interleaved rows of a three-way split
    ref2, _ = orc.render(desc, rs, w, h, sample_offset=rs.spp, prev_color=ref1, threads=4)
    for order in ("1", "0"):
        monkeypatch.setenv("GATLING_OPTIONS", f"work_order={order}")
        sc = gi.Scene(desc)
        try:
            sc.set_option(gi.OPTION_FUSED_PATH, 0)
            a = sc.render(rs, w, h).copy()
            b = sc.render(rs, w, h).copy()
            rs1 = RenderSettings(spp=24, max_bounces=5, progressive_accumulation=False)
            share = sc.render(rs1, w, h, rows=(1, h), row_stride=3).copy()
        finally:
            sc.close()
        assert_image_parity(a, ref1, exact=True)
        assert_image_parity(b, ref2, exact=True)
        assert np.array_equal(share.view(np.uint32), ref1[1::3].view(np.uint32)), order
    monkeypatch.delenv("GATLING_OPTIONS")


def test_texture_coordinate_transforms_on_device(gi, orc):
    """UsdTransform2d folded into the texture bindings (gi_texture.h tex_transform_st): every textured input of the texture test scene and the textured cutout
    opacity of the leaf cards (looked up inside the any-hit test) with a rotation + scale + translation -- device == oracle, bit for bit."""
    from gatling_amd.scene import usd_transform_2d
    desc = textured_scene()
    k = 0
    for m in desc.materials:
        for slot, b in m.textures.items():
            b.transform = usd_transform_2d(17.0 * (k + 1), (1.0 + 0.3 * k, 0.7 + 0.2 * k), (0.1 * k, -0.05 * k)); k += 1
    assert k >= 6
    render_both(gi, orc, desc, RenderSettings(spp=4, max_bounces=6, next_event_estimation=True), 96, 54, exact=True)
    leaves, rs, w, h = build_case("leaf_cards_64x36_spp4_b5")
    n = 0
    for m in leaves.materials:
        for slot, b in getattr(m, "textures", {}).items():
            b.transform = usd_transform_2d(-40.0, (2.0, 1.5), (0.3, 0.6)); n += 1
    assert n >= 1
    img, ref, st = render_both(gi, orc, leaves, rs, w, h, exact=True)
    g = np.load(os.path.join(GOLDEN, "leaf_cards_64x36_spp4_b5.npz"))
    assert not np.array_equal(img, g["color"])     # the transform really moved the cutout pattern


def test_volumetric_subsurface_on_device(gi, orc):
    """OpenPBR's volumetric subsurface_bsdf (open_pbr_surface.mtlx:182-192) through the medium stack: diffuse-transmission entry, the subsurface medium pushed
    (MaterialRec::sss), Henyey-Greenstein walk, exit through the same lobe from inside -- device == oracle bit for bit, with and without NEE, mixed with dielectric
    transmission (two media of one material) and a coat; and with mediumStackSize 0 the lobe is off on both sides."""
    desc = sphere_grid(3, 2, 6)
    M = MaterialDesc
    desc.materials = [M.open_pbr(name="wax", base_color=(0.9, 0.8, 0.6), subsurface_weight=1.0, subsurface_color=(0.9, 0.5, 0.3), subsurface_radius=0.15, specular_roughness=0.4),
                      M.open_pbr(name="milk", base_color=(0.8, 0.8, 0.8), subsurface_weight=0.6, subsurface_color=(0.95, 0.93, 0.88), subsurface_radius=0.4,
                                 subsurface_radius_scale=(1.0, 0.7, 0.4), subsurface_scatter_anisotropy=0.6, coat_weight=0.5, coat_roughness=0.1),
                      M.open_pbr(name="jade", base_color=(0.2, 0.6, 0.3), subsurface_weight=0.8, subsurface_color=(0.3, 0.8, 0.4), subsurface_radius=0.05,
                                 subsurface_scatter_anisotropy=-0.4, transmission_weight=0.3, transmission_color=(0.7, 0.9, 0.7), transmission_depth=0.5),
                      M.open_pbr(name="thin", base_color=(0.5, 0.5, 0.9), geometry_thin_walled=True, subsurface_weight=0.7, subsurface_color=(0.4, 0.4, 0.9)),
                      M.open_pbr(name="plain", base_color=(0.7, 0.7, 0.7)),
                      M.usd_preview_surface(name="ups", diffuseColor=(0.6, 0.3, 0.2))]
    desc.rect_lights = [RectLight(origin=(0.0, -1.0, 3.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(12, 12, 12), width=1.5, height=1.5)]
    for nee in (False, True):
        rs = RenderSettings(spp=4, max_bounces=24, next_event_estimation=nee, medium_stack_size=3)
        img, ref, st = render_both(gi, orc, desc, rs, 80, 45, exact=True)
    rs0 = RenderSettings(spp=4, max_bounces=24, medium_stack_size=0)
    off, _, _ = render_both(gi, orc, desc, rs0, 80, 45, exact=True)
    assert not np.array_equal(off, img)


def test_coat_normal_on_device(gi, orc):
    """OpenPBR geometry_coat_normal as a seventh texture slot (gi_shading.h resolve_material_textures / to_local_coat): the coat lobe in its own frame -- sampling,
    evaluation (NEE), Fresnel split with the base -- on the coated ball of the oracle test and on the texture scene's OpenPBR ball with base AND coat normal maps and a
    texture-coordinate transform on the coat map: device == oracle bit for bit; the Albedo AOV follows the coat frame too."""
    from test_oracle_render import _coated_ball
    from gatling_amd.scene import TEX_COAT_NORMAL, TEX_NORMAL, TextureBinding, usd_transform_2d
    yy, xx = np.mgrid[0:32, 0:64]
    bumpy = np.zeros((32, 64, 4), np.float32)
    bumpy[..., 0] = 0.5 + 0.35 * np.sin(xx * 1.7); bumpy[..., 1] = 0.5 + 0.35 * np.cos(yy * 2.3); bumpy[..., 2] = 0.85; bumpy[..., 3] = 1.0
    rs = RenderSettings(spp=6, max_bounces=4, next_event_estimation=True)
    render_both(gi, orc, _coated_ball(bumpy), rs, 64, 64, exact=True)
    desc = textured_scene()
    desc.textures.append(bumpy)
    ball = desc.materials[1]
    assert ball.klass == MAT_OPEN_PBR
    ball.params = MaterialDesc.open_pbr(base_color=(0.8, 0.8, 0.8), specular_roughness=0.25, coat_weight=0.8, coat_roughness=0.1, coat_color=(0.9, 0.8, 0.7)).params
    ball.textures[TEX_COAT_NORMAL] = TextureBinding(texture=len(desc.textures) - 1, scale=(2.0, 2.0, 2.0, 1.0), bias=(-1.0, -1.0, -1.0, 0.0),
                                                   transform=usd_transform_2d(25.0, (3.0, 2.0), (0.1, 0.2)))
    ball.textures[TEX_NORMAL] = TextureBinding(texture=3, scale=(2.0, 2.0, 2.0, 1.0), bias=(-1.0, -1.0, -1.0, 0.0))
    img, ref, st = render_both(gi, orc, desc, RenderSettings(spp=4, max_bounces=6, next_event_estimation=True), 96, 54, exact=True)
    sc = gi.Scene(desc)
    try:
        got = sc.render_aovs(RenderSettings(spp=2, max_bounces=2, progressive_accumulation=False), 96, 54, ["albedo"])["albedo"]
    finally:
        sc.close()
    want = orc.render_aovs(desc, RenderSettings(spp=2, max_bounces=2, progressive_accumulation=False), 96, 54, ["albedo"])["albedo"]
    assert np.array_equal(got[..., :3].view(np.uint32), want[..., :3].view(np.uint32))


def test_coat_tangent_on_device(gi, orc):
    """OpenPBR geometry_coat_tangent / geometry_tangent as turns of the coat's / the base lobes' tangent (GI_C_P_COAT_ROTATION, GI_C_P_SPECULAR_ROTATION; gi_shading.h coat_turn_local / coat_turn_world, cosine and sine from the
    host in MaterialRec::sss[6..7]): an anisotropic coat with its tangent turned, sampled and evaluated (NEE) -- on the coated ball alone, with a coat normal map (the
    turn applies inside the map's frame), beside an untouched material in one scene, through the fused kernels (no NEE) and with a medium stack: device == oracle
    bit for bit; the turn changes the image; the class the material is binned to is the full OpenPBR one."""
    from test_oracle_render import _coated_ball
    yy, xx = np.mgrid[0:32, 0:64]
    bumpy = np.zeros((32, 64, 4), np.float32)
    bumpy[..., 0] = 0.5 + 0.35 * np.sin(xx * 1.7); bumpy[..., 1] = 0.5 + 0.35 * np.cos(yy * 2.3); bumpy[..., 2] = 0.85; bumpy[..., 3] = 1.0
    def ball(turn, coat_map=None):
        d = _coated_ball(coat_map)
        d.materials[0].params = MaterialDesc.open_pbr(base_color=(0.05, 0.05, 0.05), specular_weight=0.0, coat_weight=1.0, coat_roughness=0.3, coat_ior=1.6,
                                                      coat_roughness_anisotropy=0.85, coat_rotation=turn).params
        return d
    rs = RenderSettings(spp=6, max_bounces=4, next_event_estimation=True)
    plain, _, _ = render_both(gi, orc, ball(0.0), rs, 64, 64, exact=True)
    turned, _, _ = render_both(gi, orc, ball(0.125), rs, 64, 64, exact=True)
    assert not np.array_equal(plain, turned)
    render_both(gi, orc, ball(0.3, bumpy), rs, 64, 64, exact=True)
    render_both(gi, orc, ball(-0.2), RenderSettings(spp=5, max_bounces=5), 48, 48, exact=True)
    render_both(gi, orc, ball(0.7), RenderSettings(spp=4, max_bounces=5, next_event_estimation=True, medium_stack_size=3), 48, 48, exact=True)
    desc = textured_scene()
    b = desc.materials[1]
    assert b.klass == MAT_OPEN_PBR
    b.params = MaterialDesc.open_pbr(base_color=(0.8, 0.8, 0.8), specular_roughness=0.25, coat_weight=0.8, coat_roughness=0.2, coat_color=(0.9, 0.8, 0.7),
                                     coat_roughness_anisotropy=0.6, coat_rotation=0.45).params
    render_both(gi, orc, desc, RenderSettings(spp=4, max_bounces=6, next_event_estimation=True), 96, 54, exact=True)
    # geometry_tangent (GI_C_P_SPECULAR_ROTATION; gi_shading.h spec_turn_frame: the shading frame's tangents turned in place before the BSDF): a brushed metal ball,
    # then the same under an anisotropic coat (the coat's relative turn), with a coat normal map (the coat's absolute turn in the map's frame), without NEE, with a
    # medium stack; and on the texture scene's ball with a base normal map
    def brushed(turn, coat=0.0, coat_turn=0.0, coat_map=None):
        d = _coated_ball(coat_map)
        d.materials[0].params = MaterialDesc.open_pbr(base_color=(0.9, 0.7, 0.4), base_metalness=1.0, specular_roughness=0.35, specular_roughness_anisotropy=0.85, specular_rotation=turn,
                                                      coat_weight=coat, coat_roughness=0.25, coat_ior=1.6, coat_roughness_anisotropy=0.7, coat_rotation=coat_turn).params
        return d
    plain, _, _ = render_both(gi, orc, brushed(0.0), rs, 64, 64, exact=True)
    turned, _, _ = render_both(gi, orc, brushed(0.125), rs, 64, 64, exact=True)
    assert not np.array_equal(plain, turned)
    render_both(gi, orc, brushed(0.3, coat=0.7), rs, 64, 64, exact=True)
    render_both(gi, orc, brushed(-0.2, coat=0.7, coat_turn=0.6, coat_map=bumpy), rs, 64, 64, exact=True)
    render_both(gi, orc, brushed(0.4, coat=0.5, coat_turn=0.1), RenderSettings(spp=5, max_bounces=5), 48, 48, exact=True)
    render_both(gi, orc, brushed(0.7), RenderSettings(spp=4, max_bounces=5, next_event_estimation=True, medium_stack_size=3), 48, 48, exact=True)
    b.params = MaterialDesc.open_pbr(base_color=(0.8, 0.8, 0.8), specular_roughness=0.3, specular_roughness_anisotropy=0.7, specular_rotation=0.15, transmission_weight=0.3,
                                     coat_weight=0.5, coat_roughness=0.2, coat_roughness_anisotropy=0.6).params
    render_both(gi, orc, desc, RenderSettings(spp=4, max_bounces=6, next_event_estimation=True), 96, 54, exact=True)


def test_textured_transmission_inputs_on_device(gi, orc):
    """OpenPBR transmission_weight / transmission_color resolved per hit (slots 7 / 8; gi_shading.h resolve_material_textures, opbr_params): a weight map, a colour map
    with a texture-coordinate transform, both from scene data (per-vertex weight, constant colour), and with a transmission_depth -- where the colour is the medium's
    and stays the material's constant -- without and with a medium stack: device == oracle bit for bit, image and Albedo AOV."""
    from test_oracle_render import _const_tex, _glass_ball
    from gatling_amd.scene import TEX_TRANSMISSION_COLOR, usd_transform_2d
    yy, xx = np.mgrid[0:16, 0:32]
    wmap = np.zeros((16, 32, 4), np.float32); wmap[..., 0] = wmap[..., 1] = wmap[..., 2] = 0.5 + 0.5 * np.sin(xx * 0.9) * np.cos(yy * 0.7); wmap[..., 3] = 1.0
    cmap = np.zeros((8, 8, 4), np.float32); cmap[..., 0] = (xx[:8, :8] % 2) * 0.6 + 0.3; cmap[..., 1] = 0.9; cmap[..., 2] = (yy[:8, :8] % 3) * 0.3 + 0.2; cmap[..., 3] = 1.0
    rs = RenderSettings(spp=6, max_bounces=6, next_event_estimation=True)
    desc = _glass_ball(weight=wmap, colour=cmap)
    desc.materials[0].textures[TEX_TRANSMISSION_COLOR].transform = usd_transform_2d(30.0, (2.0, 3.0), (0.1, 0.3))
    render_both(gi, orc, desc, rs, 72, 72, exact=True)
    render_both(gi, orc, _glass_ball(primvar=True), rs, 72, 72, exact=True)
    render_both(gi, orc, _glass_ball(weight=_const_tex(0.7, 0.7, 0.7, 1.0), tc=(0.3, 0.9, 0.5)), rs, 72, 72, exact=True)
    for stack in (0, 2):
        rs2 = RenderSettings(spp=4, max_bounces=6, next_event_estimation=True, medium_stack_size=stack)
        render_both(gi, orc, _glass_ball(weight=wmap, colour=cmap, tc=(0.3, 0.9, 0.5), depth=0.5), rs2, 72, 72, exact=True)
    sc = gi.Scene(desc)
    try:
        got = sc.render_aovs(RenderSettings(spp=2, max_bounces=2, progressive_accumulation=False), 72, 72, ["albedo"])["albedo"]
    finally:
        sc.close()
    want = orc.render_aovs(desc, RenderSettings(spp=2, max_bounces=2, progressive_accumulation=False), 72, 72, ["albedo"])["albedo"]
    assert np.array_equal(got[..., :3].view(np.uint32), want[..., :3].view(np.uint32))


def _aov_scene():
    desc = sphere_grid(grid=3, subdivisions=1, material_count=4)
    rng = np.random.default_rng(8)
    for i, m in enumerate(desc.meshes):
        m.id = 100 + i
        m.face_ids = rng.integers(0, 300, len(m.faces)).astype(np.int32)
        m.max_face_id = 299 if i % 2 else 40  # 2-byte and 1-byte strides
        m.instance_ids = np.arange(len(m.instance_transforms), dtype=np.int32) * 7 + i
        m.double_sided = bool(i % 2)
    desc.materials[1] = MaterialDesc.open_pbr(base_color=(0.9, 0.5, 0.2), base_metalness=1.0, coat_weight=0.5)
    return desc


AOV_NAMES = ["normal", "barycentrics", "texcoords", "opacity", "tangents", "bitangents", "thinWalled", "objectId", "depth", "faceId",
             "instanceId", "doubleSided", "albedo"]
AOV_CLEAR = {"normal": (0.5, 0.5, 0.5, 0.5), "objectId": -1, "faceId": -1, "instanceId": -1, "depth": 1.0, "albedo": (0.1, 0.2, 0.3, 0.0)}


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b)


def test_non_colour_aovs(gi, orc):
    """The 13 non-colour AOVs the core produces (rp_main.chit:192-290, rp_main.rgen:132-183, 517-520) == oracle, together with
    the colour AOV in the same giRender call, then again as a progressive second call (the accumulating Normal/Albedo AOVs blend)."""
    desc = _aov_scene()
    rs = RenderSettings(spp=3, max_bounces=3, clipping_planes=True)
    desc.camera.clip_start, desc.camera.clip_end = 0.5, 40.0
    w, h = 64, 36
    sc = gi.Scene(desc)
    try:
        first = sc.render_aovs(rs, w, h, AOV_NAMES, AOV_CLEAR)
        second = sc.render_aovs(rs, w, h, AOV_NAMES, AOV_CLEAR)
    finally:
        sc.close()
    ref_color, _ = orc.render(desc, rs, w, h)
    ref1 = orc.render_aovs(desc, rs, w, h, AOV_NAMES, AOV_CLEAR)
    ref2 = orc.render_aovs(desc, rs, w, h, AOV_NAMES, AOV_CLEAR, sample_offset=3, prev=ref1)
    assert_image_parity(first["color"], ref_color, exact=True)
    for name in AOV_NAMES:
        assert _same(first[name], ref1[name]), name
        assert _same(second[name], ref2[name]), name
    hit = first["objectId"] >= 100
    assert 0.1 < hit.mean() < 0.9 and set(np.unique(first["objectId"][hit])) <= {100, 101, 102, 103}
    assert (first["faceId"][hit] >= 0).all() and (first["faceId"][hit] <= 15).all()  # the reference's (stride*8-1) mask, kept as is
    assert np.isfinite(first["depth"][hit]).all() and (np.abs(first["depth"][hit]) <= 1.0).all()


def test_aov_only_render_and_unproduced_aovs(gi):
    """Without a colour binding only the primary-hit AOV pass runs; with NEE off no shadow ray is traced, so the NEE AOV keeps its
    clear value (the reference never clears or writes it then)."""
    desc = cornell_box(MAT_DIFFUSE)
    sc = gi.Scene(desc)
    try:
        out = sc.render_aovs(RenderSettings(spp=2, max_bounces=2), 48, 27, ["objectId", "nee"], {"objectId": -1, "nee": (0.25, 0.5, 0.75, 1.0)},
                             with_color=False)
        st = sc.stats()
    finally:
        sc.close()
    assert "color" not in out and (out["objectId"] >= -1).all() and (out["objectId"] >= 0).mean() > 0.3
    assert np.allclose(out["nee"], (0.25, 0.5, 0.75, 1.0))


def test_edge_cases(gi, orc):
    """Empty scene, 1x1 target (the reference's Render.Empty1x1), invisible / instance-less / material-less meshes."""
    cam = CameraDesc(position=(0, 0, 5))
    empty = SceneDesc(materials=[MaterialDesc.usd_preview_surface()], camera=cam)
    render_both(gi, orc, empty, RenderSettings(spp=2, max_bounces=3, clear_color=(0.5, 0.25, 1.0, 1.0)), 1, 1, exact=True)
    render_both(gi, orc, empty, RenderSettings(spp=1, max_bounces=1), 7, 3, exact=True)
    desc = cornell_box(MAT_DIFFUSE)
    desc.meshes[6].visible = False
    desc.meshes[7].instance_transforms = np.zeros((0, 4, 4), np.float32)
    desc.meshes[5].material = -1
    render_both(gi, orc, desc, RenderSettings(spp=2, max_bounces=4), 48, 27, exact=True)


def test_parallel_scene_sync(gi, orc):
    """Hydra syncs prims from a thread pool (renderDelegate.cpp:376-381): meshes, materials and lights created and edited from eight
    threads at once -- no crash, and once the creation ORDER is made deterministic again (one mesh per thread-safe slot) the image is the
    sequential one."""
    import ctypes as C
    import threading
    from gatling_amd import capi
    desc = sphere_grid(grid=4, subdivisions=1, material_count=4)
    rs = RenderSettings(spp=2, max_bounces=4, progressive_accumulation=False)
    sc = gi.Scene(desc)
    try:
        ref = sc.render(rs, 64, 36)
        L = sc.L
        errors = []

        def hammer(k):
            try:
                for it in range(40):
                    m = desc.meshes[(k + it) % len(desc.meshes)]
                    h = sc.meshes[(k + it) % len(sc.meshes)]
                    L.giCSetMeshTransform(h, capi._fp(np.asarray(m.transform, np.float32).reshape(-1)))
                    L.giCSetMeshVisibility(h, 1)
                    md = capi.GiCMaterialDesc(1, 0, (C.c_float * capi.P_COUNT)(*([0.5] * capi.P_COUNT)))
                    mat = L.giCCreateMaterial(sc.handle, b"tmp", C.byref(md))
                    light = L.giCCreateSphereLight(sc.handle)
                    L.giCSetSphereLightRadius(light, 0.1, 0.1, 0.1)
                    L.giCDestroySphereLight(sc.handle, light)
                    L.giCDestroyMaterial(mat)
            except Exception as e:  # noqa: BLE001
                errors.append(e)
        threads = [threading.Thread(target=hammer, args=(k,)) for k in range(8)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors
        again = sc.render(rs, 64, 36)  # same meshes, same order, same transforms: the scene was only marked dirty and rebuilt
    finally:
        sc.close()
    assert np.array_equal(again, ref)


def test_error_behaviour(gi):
    desc = cornell_box()
    sc = gi.Scene(desc)
    try:
        with pytest.raises(gi.GiError):
            sc.render(RenderSettings(spp=0), 8, 8)
        with pytest.raises(gi.GiError):
            sc.render(RenderSettings(spp=1, medium_stack_size=16), 8, 8)  # the payload's medium index has four bits: stacks deeper than 15 cannot be addressed
        img = sc.render(RenderSettings(spp=1, max_bounces=2), 8, 8)  # still usable afterwards
        assert np.isfinite(img).all()
    finally:
        sc.close()


def test_device_square_root_is_the_correctly_rounded_one(gi):
    """gi_device_math.h gi_sqrt -- the compiler's correctly rounded sqrtf expansion with the denormal scaling and the zero / infinity fix-up taken off the path of
    arguments in [2^-95, +inf) -- equals sqrtf for EVERY float: all 2^32 bit patterns on the device, a second of one GPU (the arithmetic contract of DESIGN.md
    section 2 says IEEE square root; the oracle's is the host's)."""
    L = gi.load_library()
    assert L.giCDebugCheckSqrt(0, 1 << 32) == 0
    assert L.giCDebugCheckSqrt(0x0f000000, 0x02000000) == 0   # around the range test's lower edge


def test_bsdf_known_answers_on_device(gi, orc):
    """Closed-form BSDF sample/evaluate on random frames: device == oracle."""
    rng = np.random.default_rng(5)
    n = 4096
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    t = np.cross(nrm, rng.normal(size=(n, 3))); t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(nrm, t)

    def hemi(z_min):
        v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        v[:, 2] = np.abs(v[:, 2]) * (1 - z_min) + z_min
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        return v[:, :1] * t + v[:, 1:2] * b + v[:, 2:3] * nrm
    items = np.concatenate([nrm, t, b, nrm, hemi(0.02), hemi(0.02), rng.uniform(size=(n, 4))], axis=1).astype(np.float32)
    mats = [MaterialDesc.usd_preview_surface(diffuseColor=(0.7, 0.4, 0.2), klass=MAT_DIFFUSE),
            MaterialDesc.usd_preview_surface(diffuseColor=(0.7, 0.4, 0.2), roughness=0.5),
            MaterialDesc.usd_preview_surface(diffuseColor=(0.9, 0.6, 0.1), roughness=0.2, metallic=1.0),
            MaterialDesc.usd_preview_surface(diffuseColor=(0.2, 0.4, 0.8), roughness=0.7, clearcoat=1.0, clearcoatRoughness=0.1),
            MaterialDesc.usd_preview_surface(diffuseColor=(0.2, 0.4, 0.8), useSpecularWorkflow=1, specularColor=(0.3, 0.2, 0.1), roughness=0.05),
            MaterialDesc.open_pbr(),
            MaterialDesc.open_pbr(base_color=(0.9, 0.5, 0.2), base_metalness=1.0, specular_color=(0.8, 0.9, 1.0), specular_roughness=0.25),
            MaterialDesc.open_pbr(base_color=(0.3, 0.6, 0.2), coat_weight=0.8, coat_color=(0.9, 0.7, 0.6), coat_roughness=0.2, specular_weight=0.7),
            MaterialDesc.open_pbr(transmission_weight=1.0, transmission_color=(0.6, 0.8, 0.9), transmission_depth=0.5, specular_roughness=0.1),
            MaterialDesc.open_pbr(transmission_weight=0.6, base_metalness=0.3, coat_weight=0.4, specular_ior=1.33),
            MaterialDesc.open_pbr(base_color=(0.8, 0.4, 0.2), base_diffuse_roughness=0.7, coat_weight=1.0, coat_roughness=0.5, coat_color=(0.9, 0.8, 0.7), coat_darkening=0.6),
            MaterialDesc.open_pbr(base_color=(0.5, 0.5, 0.9), base_diffuse_roughness=1.0, specular_weight=0.3, base_weight=0.8),
            MaterialDesc.open_pbr(transmission_weight=1.0, specular_roughness=0.3, specular_ior=1.45, geometry_thin_walled=True),
            # thin-walled subsurface (open_pbr_surface.mtlx:140-196): alone, and mixed with the base diffuse under a coat with a rough Oren-Nayar reflection
            MaterialDesc.open_pbr(base_color=(0.2, 0.4, 0.8), geometry_thin_walled=True, subsurface_weight=1.0, subsurface_color=(0.9, 0.6, 0.3), subsurface_scatter_anisotropy=0.25),
            MaterialDesc.open_pbr(base_color=(0.7, 0.7, 0.2), geometry_thin_walled=True, subsurface_weight=0.55, subsurface_color=(0.4, 0.8, 0.5), subsurface_scatter_anisotropy=-0.4,
                                  base_diffuse_roughness=0.6, coat_weight=0.5, coat_roughness=0.15, transmission_weight=0.2),
            # fuzz layer (open_pbr_surface.mtlx:569-581): alone over a dark base, smooth fuzz (albedo table's steep corner) over a coat, rough fuzz over metal
            MaterialDesc.open_pbr(base_color=(0.05, 0.05, 0.05), specular_weight=0.2, fuzz_weight=1.0, fuzz_color=(0.9, 0.5, 0.3), fuzz_roughness=0.5),
            MaterialDesc.open_pbr(base_color=(0.6, 0.2, 0.2), coat_weight=0.7, coat_roughness=0.1, fuzz_weight=0.8, fuzz_color=(0.8, 0.8, 1.0), fuzz_roughness=0.07),
            MaterialDesc.open_pbr(base_color=(0.9, 0.7, 0.3), base_metalness=1.0, specular_roughness=0.35, fuzz_weight=0.45, fuzz_color=(1.0, 0.9, 0.8), fuzz_roughness=1.0),
            MaterialDesc.open_pbr(base_color=(0.3, 0.5, 0.7), transmission_weight=0.5, fuzz_weight=0.3, fuzz_roughness=0.23),
            # specular / coat anisotropy (open_pbr_surface.mtlx:133-136, 552-555): metal, dielectric with transmission, anisotropic coat over an isotropic base
            MaterialDesc.open_pbr(base_color=(0.9, 0.8, 0.6), base_metalness=1.0, specular_roughness=0.45, specular_roughness_anisotropy=0.8),
            MaterialDesc.open_pbr(base_color=(0.4, 0.6, 0.8), transmission_weight=0.7, specular_roughness=0.3, specular_roughness_anisotropy=0.5, specular_ior=1.4),
            MaterialDesc.open_pbr(base_color=(0.6, 0.3, 0.3), specular_roughness=0.4, coat_weight=0.9, coat_roughness=0.35, coat_roughness_anisotropy=0.95, fuzz_weight=0.2),
            MaterialDesc.open_pbr(base_color=(0.5, 0.5, 0.5), specular_roughness=0.2, specular_roughness_anisotropy=1.0, coat_weight=0.3, coat_roughness=0.1, coat_roughness_anisotropy=0.3),
            # geometry_coat_tangent (open_pbr_surface.mtlx:91, 561) as a turn of the coat's tangent: alone, over an anisotropic metal, negative / beyond one turn with fuzz
            MaterialDesc.open_pbr(base_color=(0.5, 0.5, 0.5), specular_weight=0.0, coat_weight=1.0, coat_ior=3.0, coat_roughness=0.3, coat_roughness_anisotropy=0.75, coat_rotation=0.125),
            MaterialDesc.open_pbr(base_color=(0.9, 0.8, 0.6), base_metalness=1.0, specular_roughness=0.4, specular_roughness_anisotropy=0.7, coat_weight=0.6, coat_roughness=0.25,
                                  coat_roughness_anisotropy=0.5, coat_rotation=0.3),
            MaterialDesc.open_pbr(base_color=(0.6, 0.3, 0.3), specular_roughness=0.4, coat_weight=0.9, coat_roughness=0.35, coat_roughness_anisotropy=0.95, fuzz_weight=0.2, coat_rotation=-1.71),
            MaterialDesc.open_pbr(base_color=(0.6, 0.3, 0.3), coat_weight=0.9, coat_roughness=0.35, coat_roughness_anisotropy=0.0, coat_rotation=0.4),   # isotropic coat: not read
            # geometry_tangent (open_pbr_surface.mtlx:89) as a turn of the base lobes' tangent: metal, dielectric with transmission and a rough diffuse base, under an
            # anisotropic coat without / with a turn of its own (the coat's relative turn), with fuzz and thin film, isotropic base (not read)
            MaterialDesc.open_pbr(base_color=(0.9, 0.8, 0.6), base_metalness=1.0, specular_roughness=0.45, specular_roughness_anisotropy=0.8, specular_rotation=0.125),
            MaterialDesc.open_pbr(base_color=(0.4, 0.6, 0.8), transmission_weight=0.6, specular_roughness=0.3, specular_roughness_anisotropy=0.5, specular_ior=1.4, base_diffuse_roughness=0.7,
                                  specular_rotation=-0.31),
            MaterialDesc.open_pbr(base_color=(0.5, 0.5, 0.5), specular_roughness=0.2, specular_roughness_anisotropy=1.0, coat_weight=0.5, coat_roughness=0.2, coat_roughness_anisotropy=0.6,
                                  specular_rotation=0.2),
            MaterialDesc.open_pbr(base_color=(0.5, 0.5, 0.5), base_metalness=0.5, specular_roughness=0.3, specular_roughness_anisotropy=0.7, coat_weight=0.5, coat_roughness=0.2,
                                  coat_roughness_anisotropy=0.6, specular_rotation=0.2, coat_rotation=0.45, fuzz_weight=0.3, thin_film_weight=0.6, thin_film_thickness=0.3),
            MaterialDesc.open_pbr(base_color=(0.6, 0.3, 0.3), specular_roughness=0.4, specular_rotation=0.4, coat_weight=0.4, coat_roughness_anisotropy=0.5),
            # thin film (open_pbr_surface.mtlx:300-304, 404-431, 450-464): on a dielectric with transmission (front and back faces: relative indices), on a metal, with everything else on
            MaterialDesc.open_pbr(base_color=(0.7, 0.7, 0.7), transmission_weight=0.6, specular_roughness=0.2, thin_film_weight=1.0, thin_film_thickness=0.35, thin_film_ior=1.8),
            MaterialDesc.open_pbr(base_color=(0.9, 0.6, 0.4), base_metalness=1.0, specular_roughness=0.3, thin_film_weight=0.8, thin_film_thickness=0.6, thin_film_ior=1.33),
            MaterialDesc.open_pbr(base_color=(0.4, 0.5, 0.6), base_metalness=0.4, coat_weight=0.5, coat_roughness=0.1, fuzz_weight=0.3, specular_roughness_anisotropy=0.4,
                                  geometry_thin_walled=True, subsurface_weight=0.5, thin_film_weight=0.5, thin_film_thickness=0.2, thin_film_ior=2.1)]
    items[: n // 2, 21] = 0.75  # xi.w >= 0.5: the debug hook shades these as back faces (eta inverted)
    for m in mats:
        got, ref = gi.bsdf_debug(m, items), orc.bsdf_debug(m, items)
        assert np.array_equal(got[:, 7], ref[:, 7])  # event types
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_three_pipelines_for_lds_resident_scenes_agree(gi, orc):
    """The same frame through the wavefront stage kernels (option 0), the lane-per-path fused kernel k_path (2, the default) and the wave-local
    wavefront k_path_bw (1): bit-identical images and segment counts, also with cutouts / textures in play and
    for work totals that do not fill the waves' path pools."""
    from gatling_amd.capi import OPTION_FUSED_PATH
    cases = [(cornell_box(), RenderSettings(spp=6, max_bounces=8), 96, 54), (cornell_box(MAT_DIFFUSE), RenderSettings(spp=3, max_bounces=4), 7, 5),
             (cornell_box(), RenderSettings(spp=1, max_bounces=3, rr_bounce_offset=0), 33, 1)]
    for desc, rs, w, h in cases:
        ref, cnt = orc.render(desc, rs, w, h, threads=4)
        for mode in (0, 2, 1):
            sc = gi.Scene(desc)
            try:
                sc.set_option(OPTION_FUSED_PATH, mode)
                img = sc.render(rs, w, h)
                stats = sc.stats()
            finally:
                sc.close()
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (mode, w, h)
            assert stats["segments"] == cnt["segments"] and stats["fusedPath"] == (1 if mode else 0)


def test_c4_parameter_sets_bsdf_properties_on_device(gi, orc):
    """The 32 material parameter sets of config C4 (sphere_grid / _parameter_sets; OpenPBR and UsdPreviewSurface, metal / coat /
    transmission mixes), evaluated ON THE DEVICE through the debug hook: white-furnace bound (directional albedo of the white version
    <= 1), consistency of evaluate() with the sampling routine (hemisphere integral of evaluate == mean sampled reflection weight, pdf
    integrates to the reflection probability), Helmholtz reciprocity of evaluate() for the opaque sets -- and bit-equality with the oracle."""
    import copy
    from gatling_amd.scene import P_BASE_COLOR, P_TRANSMISSION_WEIGHT, P_CLEARCOAT, P_METALLIC, P_ROUGHNESS
    from test_oracle_render import _frames
    mats = sphere_grid(4, 1, 32).materials
    assert len(mats) == 32
    rng = np.random.default_rng(77)
    for i, m in enumerate(mats):
        white = copy.deepcopy(m); white.params[P_BASE_COLOR:P_BASE_COLOR + 3] = 1.0
        for c in (0.9, 0.4):
            items = _frames(60000, rng, c)
            out = gi.bsdf_debug(white, items)
            assert np.isfinite(out).all()
            albedo = out[:, 3:6].mean(axis=0)
            assert np.all(albedo <= 1.03) and np.all(albedo >= 0.0), (i, c, albedo)
        items = _frames(120000, rng, 0.7)
        out = gi.bsdf_debug(m, items)
        if i % 8 == 0:
            assert np.array_equal(out.view(np.uint32), orc.bsdf_debug(m, items).view(np.uint32))
        refl = (out[:, 7].astype(int) & 8) != 0  # EV_REFLECTION
        sampled = (out[:, 3:6] * refl[:, None]).mean(axis=0)
        integ = (out[:, 8:11] + out[:, 11:14]).mean(axis=0) * (2 * np.pi)  # uniform hemisphere pdf 1 / (2 pi); evaluate returns bsdf * cos
        # (uniform-direction quadrature of a GGX peak needs far more samples below roughness ~0.2 or under a 0.05-rough coat: those sets get a loose bound)
        tol = 0.015 if (m.params[P_ROUGHNESS] > 0.2 and m.params[P_CLEARCOAT] == 0.0) else 0.08
        np.testing.assert_allclose(integ, sampled, rtol=0.08, atol=tol, err_msg=f"set {i}")
        np.testing.assert_allclose(out[:, 14].mean() * 2 * np.pi, refl.mean(), rtol=0.08, atol=tol, err_msg=f"set {i}")
        # reciprocity f(k1, k2) = f(k2, k1) holds for the symmetric lobes; the layered forms weight the base by the view-side Fresnel only
        # (1 - F(n.k1)), as MDL's fresnel_layer does, which is not symmetric: check the sets without coat / dielectric layering
        if m.params[P_CLEARCOAT] == 0.0 and m.params[P_TRANSMISSION_WEIGHT] == 0.0 and m.params[P_METALLIC] == 1.0:
            it = _frames(4000, rng, 0.6)
            sw = it.copy(); sw[:, 12:15], sw[:, 15:18] = it[:, 15:18], it[:, 12:15]
            a, b = gi.bsdf_debug(m, it), gi.bsdf_debug(m, sw)
            ok = it[:, 17] > 0.05
            fa = (a[ok, 8:14].reshape(-1, 2, 3).sum(axis=1)) / it[ok, 17:18]
            fb = (b[ok, 8:14].reshape(-1, 2, 3).sum(axis=1)) / sw[ok, 17:18]
            np.testing.assert_allclose(fa, fb, rtol=2e-3, atol=1e-6, err_msg=f"set {i}")


def _soup(n, seed=99):
    d = random_triangle_soup(n, seed=seed)
    return d


def test_bvh_traversal_matches_oracle(gi, orc):
    """Closest hits of random rays through the BVH8 kernel (tree larger than the LDS-staged top) vs the oracle's
    independent traversal: identical triangle ids and bit-identical (t,u,v)."""
    desc = _soup(30000)
    rng = np.random.default_rng(3)
    n = 20000
    o = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[:50, 0] = 0.0  # axis-parallel directions exercise the zero-component guard
    sc = gi.Scene(desc)
    try:
        tuv, ip = sc.trace_rays(o, d)
        assert sc.stats()["nodeCount"] > 384  # beyond LDS_NODES: exercises the global-memory node path
    finally:
        sc.close()
    rtuv, rip = orc.trace_rays(desc, o, d)
    assert np.array_equal(ip, rip)
    hit = rip[:, 0] >= 0
    assert 0.05 < hit.mean() < 1.0
    assert np.array_equal(tuv[hit].view(np.uint32), rtuv[hit].view(np.uint32))


def _glass_scene():
    """Cornell box whose two blocks are OpenPBR glass (absorbing) and coated metal: refraction, the 1-bit medium toggle,
    Beer-Lambert attenuation and the F82 metal lobe in one image."""
    desc = cornell_box(MAT_USD_PREVIEW_SURFACE)
    desc.materials.append(MaterialDesc.open_pbr(name="glass", transmission_weight=1.0, transmission_color=(0.7, 0.9, 0.8), transmission_depth=0.3,
                                                specular_roughness=0.05, specular_ior=1.5))
    desc.materials.append(MaterialDesc.open_pbr(name="metal", base_color=(0.95, 0.64, 0.54), base_metalness=1.0, specular_roughness=0.2,
                                                coat_weight=0.5, coat_roughness=0.05))
    desc.meshes[6].material = len(desc.materials) - 2
    desc.meshes[7].material = len(desc.materials) - 1
    return desc


@pytest.mark.parametrize("nee", [False, True])
def test_open_pbr_scene_parity(gi, orc, nee):
    desc = _glass_scene()
    if nee:
        desc.rect_lights = [RectLight(origin=(0, 0, 0.9), t0=(1, 0, 0), t1=(0, -1, 0), base_emission=(10, 10, 10), width=0.7, height=0.5)]
    img, ref, st = render_both(gi, orc, desc, RenderSettings(spp=8, max_bounces=10, next_event_estimation=nee, meters_per_scene_unit=2.0), 128, 72)
    assert st["segments"] > 128 * 72 * 8 * 2


@pytest.mark.parametrize("nee", [False, True])
def test_cutout_opacity_parity(gi, orc, nee):
    """Stochastic cutouts (rp_main.ahit) on closest-hit and shadow rays, order-independent any-hit randomness."""
    desc = cornell_box(MAT_USD_PREVIEW_SURFACE)
    desc.materials.append(MaterialDesc.usd_preview_surface(name="half", diffuseColor=(0.2, 0.3, 0.9), opacity=0.5))
    desc.materials.append(MaterialDesc.usd_preview_surface(name="masked", diffuseColor=(0.9, 0.9, 0.1), opacity=0.3, opacityThreshold=0.5))
    desc.materials.append(MaterialDesc.open_pbr(name="veil", base_color=(0.9, 0.2, 0.2)))
    desc.materials[-1].params[14] = 0.25  # geometry_opacity
    desc.meshes[6].material, desc.meshes[7].material, desc.meshes[3].material = 4, 5, 6
    if nee:
        desc.rect_lights = [RectLight(origin=(0, 0, 0.9), t0=(1, 0, 0), t1=(0, -1, 0), base_emission=(10, 10, 10), width=0.7, height=0.5)]
    # rr_bounce_offset 0: Russian roulette draws at every bounce, so a shadow ray's any-hit test must use the rng copy taken BEFORE that
    # draw (rp_main.rgen:399) -- the slot's state afterwards is a different number
    render_both(gi, orc, desc, RenderSettings(spp=8, max_bounces=6, next_event_estimation=nee, rr_bounce_offset=0 if nee else 3), 128, 72)


@pytest.mark.parametrize("nee", [False, True])
def test_textured_cutout_parity(gi, orc, nee):
    """Opacity textures (binary leaf mask through opacityThreshold, a soft mask used stochastically, OpenPBR geometry_opacity from the
    alpha channel) evaluated per candidate in the any-hit test of closest-hit and shadow rays -- bit-identical to the oracle."""
    from gatling_amd.scenes import leaf_card_scene
    desc = leaf_card_scene()
    rs = RenderSettings(spp=6, max_bounces=5, next_event_estimation=nee)
    img, ref, st = render_both(gi, orc, desc, rs, 128, 72)
    if nee:
        assert st["shadowRays"] > 0
        return
    # Opacity AOV: (1,0,0) on opaque materials, viridis(opacity of the accepted primary hit) on cutout materials (rp_main.ahit:45-49)
    sc = gi.Scene(desc)
    try:
        got = sc.render_aovs(rs, 96, 54, ["opacity"], with_color=False)["opacity"]
    finally:
        sc.close()
    want = orc.render_aovs(desc, rs, 96, 54, ["opacity"])["opacity"]
    assert np.array_equal(got[..., :3], want[..., :3])
    colours = {tuple(c) for c in np.round(want[..., :3].reshape(-1, 3), 4)}
    assert (1.0, 0.0, 0.0) in colours and len(colours) > 3  # ground + several opacity levels on the cards


def test_instanced_scene_parity(gi, orc):
    """Instancing + several materials (C4's structure at small scale): flattened BVH vs the oracle."""
    desc = sphere_grid(grid=4, subdivisions=1, material_count=5)
    render_both(gi, orc, desc, RenderSettings(spp=4, max_bounces=6), 96, 54)


def test_soup_scene_with_nee_parity(gi, orc):
    """C3's structure at small scale: triangle soup, rect light, NEE on."""
    desc = _soup(20000)
    render_both(gi, orc, desc, RenderSettings(spp=2, max_bounces=5, next_event_estimation=True), 96, 54)


@pytest.mark.parametrize("scene_kind", ["soup", "interior", "instances+cutouts"])
def test_shadow_walk_order_and_helper_lanes_are_invisible(gi, orc, monkeypatch, scene_kind):
    """k_trace_dyn's scheduling freedoms (round 5) change no bit: shadow walks visit children near-to-far or in slot order (pinned either way, and chosen by the library
    from its own node-visit counts across frames), and in thin launches -- few rays per launch: here every launch -- idle lanes take over parts of the longest walks.
    Images, segment and shadow-ray counts equal the oracle's in every mode and on every one of several progressive frames."""
    if scene_kind == "soup":
        desc, rs = _soup(30000, seed=3), RenderSettings(spp=2, max_bounces=6, next_event_estimation=True)
    elif scene_kind == "interior":
        desc, rs = interior_scene(clutter_instances=60, subdivisions=2, prototypes=4, material_count=8), RenderSettings(spp=2, max_bounces=6, next_event_estimation=True)
    else:
        desc = sphere_grid(grid=4, subdivisions=2, material_count=4)
        desc.materials[1].params[14] = 0.5   # stochastic cutout: the any-hit draw inside shared walks
        desc.rect_lights = [RectLight(origin=(0, 0, 6.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(15, 15, 15), width=3.0, height=3.0)]
        rs = RenderSettings(spp=2, max_bounces=5, next_event_estimation=True, rr_bounce_offset=0)
    w, h = 160, 90
    frames, prev = [], None
    for k in range(5):  # progressive frames: the library makes its own choice of the order between them (after 2^16 rays walked in either order)
        img, cnt = orc.render(desc, rs, w, h, sample_offset=k * rs.spp, prev_color=prev, threads=8)
        frames.append((img, cnt)); prev = img
    for opts in ("shadow_order=0", "shadow_order=1", ""):
        monkeypatch.setenv("GATLING_OPTIONS", opts)
        sc = gi.Scene(desc)
        try:
            for k, (ref, cnt) in enumerate(frames):
                img = sc.render(rs, w, h)
                st = sc.stats()
                assert st["segments"] == cnt["segments"] and st["shadowRays"] == cnt["shadow_rays"], (opts, k, st)
                assert_image_parity(img, ref, exact=True)
        finally:
            sc.close()


def _telescope(n, ratio):
    """n triangles whose sizes and distances from the origin shrink geometrically: SAH peels them off one cluster at a time, so the tree is
    as deep as trees get (host-side depth 39 ... 54 for the cases below against 8 ... 10 for BASELINE's scenes)."""
    from gatling_amd.meshprep import bake_vertices
    i = np.arange(n, dtype=np.float64)
    s = ratio ** (-i)
    c = np.stack([s * 3.0, 0.3 * s * ((i % 5) - 2), 0.05 * s * (i % 3)], 1)
    tri = np.array([[0, -0.5, 0], [1, 0, 0.1], [0, 0.5, 0]])
    p = (c[:, None, :] + tri[None] * s[:, None, None]).astype(np.float32).reshape(-1, 3)
    nrm = np.cross(p[1::3] - p[0::3], p[2::3] - p[0::3]); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    desc = SceneDesc(meshes=[MeshDesc("telescope", bake_vertices(p, np.repeat(nrm.astype(np.float32), 3, axis=0)), np.arange(3 * n, dtype=np.uint32).reshape(-1, 3), 0, double_sided=True)],
                     materials=[MaterialDesc.usd_preview_surface(diffuseColor=(0.8, 0.7, 0.6), roughness=0.5)],
                     camera=CameraDesc(position=(1.6, 0.0, 3.2), forward=(-0.05, 0.0, -1.0), up=(0, 1, 0), vfov=0.9))
    desc.rect_lights.append(RectLight(origin=(1.0, 0.0, 3.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(9, 9, 9), width=2.0, height=2.0))
    return desc


@pytest.mark.parametrize("n,ratio,spill8", [(100, 1.1, False), (400, 1.03, False), (2000, 1.006, False), (2000, 1.006, True), (400, 1.2, False)])
def test_deep_trees_parity(gi, orc, monkeypatch, n, ratio, spill8):
    """Trees deeper than BASELINE's scenes give (host-side depth 7 / 11 / 12 / 39): 100 triangles run the fused kernels with the 8-entry
    stack nearly full, 400 and 2000 the 16-entry stack, `spill8` the 8-entry stack with the scratch overflow, the last case the deepest
    tree the builder produces here (16 entries + overflow) -- images with shadow rays, and random rays, against the oracle.
    The last case is compared by image only: its triangles shrink to 1e-32 at the origin, and a ray that passes within float resolution of
    the origin makes the exact test's `dot(o - v0, d x e2)` cancel to exactly 0, so it "hits" a 5e-14-sized triangle it geometrically misses
    by 1e-7 -- a hit no box test can promise to keep (3 of 4000 such rays differ between this tree and the oracle's)."""
    if spill8:
        monkeypatch.setenv("GATLING_OPTIONS", "trace_dyn_spill8=1")
    desc = _telescope(n, ratio)
    render_both(gi, orc, desc, RenderSettings(spp=3, max_bounces=4, next_event_estimation=True), 80, 45)
    if ratio ** (-n) < 1e-7:
        return
    rng = np.random.default_rng(n)
    m = 4000
    o = np.stack([rng.uniform(-0.2, 3.5, m), rng.uniform(-0.6, 0.6, m), rng.uniform(0.3, 2.0, m)], 1).astype(np.float32)
    j = rng.integers(0, n, m); sj = ratio ** (-j.astype(np.float64))
    t = np.stack([3.3 * sj, 0.3 * sj * ((j % 5) - 2), 0.05 * sj * (j % 3)], 1)  # aim at triangles of every size
    d = (t - o); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    sc = gi.Scene(desc)
    try:
        tuv, ip = sc.trace_rays(o, d)
    finally:
        sc.close()
    rtuv, rip = orc.trace_rays(desc, o, d)
    assert np.array_equal(ip, rip)
    hit = rip[:, 0] >= 0
    assert hit.mean() > 0.5 and np.array_equal(tuv[hit].view(np.uint32), rtuv[hit].view(np.uint32))


@pytest.mark.parametrize("variant", ["openpbr+dome", "ups", "dome-hidden", "diffuse"])
def test_textured_scene_parity(gi, orc, variant):
    """Texture runtime (mdl_interface.glsl:8-38, 127-145, 238-256) + dome light (rp_main.miss:38-86) through the C ABI:
    base-colour / roughness / metallic / emission / normal maps with all four wrap modes, scale and bias, equirectangular
    dome with rotation and emission multiplier -- bit-identical to the oracle, colour and Albedo / Normal AOVs.  "diffuse": both materials of the diffuse-only class,
    whose base colour follows its texture too (the defect the differential campaign of round 6 found: tests/fuzz_parity.py, profiles/r06t_fuzz_parity_first_campaign.log)."""
    from gatling_amd.scene import MAT_DIFFUSE, MAT_OPEN_PBR, MAT_USD_PREVIEW_SURFACE
    desc = textured_scene(dome=variant != "ups", klass_sphere=MAT_USD_PREVIEW_SURFACE if variant == "ups" else MAT_OPEN_PBR)
    if variant == "diffuse":
        for m in desc.materials: m.klass = MAT_DIFFUSE
    rs = RenderSettings(spp=4, max_bounces=6, next_event_estimation=True, dome_light_camera_visible=variant != "dome-hidden")
    render_both(gi, orc, desc, rs, 96, 54)
    sc = gi.Scene(desc)
    try:
        got = sc.render_aovs(rs, 64, 36, ["albedo", "normal", "texcoords"], with_color=False)
    finally:
        sc.close()
    ref = orc.render_aovs(desc, rs, 64, 36, ["albedo", "normal", "texcoords"])
    for k in ("albedo", "normal", "texcoords"):
        assert np.array_equal(got[k][..., :3], ref[k][..., :3]), k


def test_dome_light_image_file(gi, orc, tmp_path):
    """giCCreateDomeLight(filePath) decodes Radiance .hdr in-library (the reference goes through imgio, Gi.cpp:2215-2230); the
    result equals handing the same decoded pixels over as a texture, and an unreadable path falls back to the clear colour."""
    import ctypes as C
    from gatling_amd import capi
    from gatling_amd.scene import DomeLight, SceneDesc, CameraDesc
    rng = np.random.default_rng(11)
    h, w = 8, 16
    rgbe = rng.integers(1, 255, (h, w, 4)).astype(np.uint8)
    rgbe[..., 3] = rng.integers(126, 131, (h, w))
    path = tmp_path / "env.hdr"
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        f.write(rgbe.tobytes())
    pixels = np.ones((h, w, 4), np.float32)
    pixels[..., :3] = rgbe[..., :3].astype(np.float32) * np.ldexp(np.float32(1.0), rgbe[..., 3:4].astype(np.int32) - 136).astype(np.float32)
    desc = SceneDesc()
    desc.camera = CameraDesc(position=(0, 0, 0), forward=(0, 0, -1), up=(0, 1, 0), vfov=1.2)
    desc.textures = [np.ascontiguousarray(pixels[::-1])]  # imgio orientation: row 0 of a loaded image = the file's bottom scanline
    desc.dome_light = DomeLight(texture=0)
    rs = RenderSettings(spp=2, max_bounces=2, clear_color=(0.3, 0.6, 0.9, 1.0))
    ref, _ = orc.render(desc, rs, 48, 24)
    sc = gi.Scene(SceneDesc(camera=desc.camera))
    try:
        L = sc.L
        sc.dome = L.giCCreateDomeLight(sc.handle, str(path).encode())
        from_file = sc.render(rs, 48, 24)
        L.giCDestroyDomeLight(sc.dome)
        sc.dome = L.giCCreateDomeLight(sc.handle, str(tmp_path / "missing.exr").encode())
        fallback = sc.render(rs, 48, 24)
    finally:
        sc.close()
    assert np.array_equal(from_file, ref)
    q = np.floor(np.float32([0.3, 0.6, 0.9]) * np.float32(255.0)) / np.float32(255.0)
    assert np.array_equal(fallback[..., :3], np.broadcast_to(q.astype(np.float32), fallback[..., :3].shape))


def test_images_through_the_asset_reader_and_the_loader_hook_render_the_file_image(gi, tmp_path):
    """The reference reads every image through the registered GiAssetReader and decodes it with imgio (TextureManager.cpp:39-52; rendererPlugin.cpp:95-143, 189).
    A dome light whose path only the asset reader can serve renders the image of the same bytes on disk, bit for bit; so does one whose format ("EXR") only the
    registered loader can decode -- asked with keepHdr set, as Gi.cpp:2215-2230 loads dome images; file textures under asset paths share the path cache."""
    import ctypes as C
    from gatling_amd import capi
    from gatling_amd.scene import CameraDesc, SceneDesc
    from test_capi_host import _MemoryAssets, _write_png
    rng = np.random.default_rng(5)
    _write_png(tmp_path / "sky.png", rng.integers(0, 256, (8, 16, 3), dtype=np.uint8), 2, 8, [0, 1, 2, 3, 4])
    blob = open(tmp_path / "sky.png", "rb").read()
    cam = CameraDesc(position=(0, 0, 0), forward=(0, 0, -1), up=(0, 1, 0), vfov=1.2)
    rs = RenderSettings(spp=2, max_bounces=2)
    L = capi.load_library()

    def dome_render(path):
        sc = gi.Scene(SceneDesc(camera=cam))
        try:
            sc.dome = L.giCCreateDomeLight(sc.handle, path.encode())
            return sc.render(rs, 48, 24).copy()
        finally:
            sc.close()

    from_file = dome_render(str(tmp_path / "sky.png"))
    assert from_file[..., :3].std() > 0.05  # the image, not the fallback colour
    w, h = C.c_uint32(), C.c_uint32()
    px = np.empty(16 * 8 * 4, np.float32)
    assert L.giCDebugDecodeImage(str(tmp_path / "sky.png").encode(), 0, C.byref(w), C.byref(h), px.ctypes.data_as(C.POINTER(C.c_float)), px.size) == 1
    seen = []

    def _load(user, path, data, size, keep_hdr, out):
        if C.string_at(data, 4) != b"EXR!":
            return 0
        seen.append((path.decode(), keep_hdr))
        out[0].format, out[0].width, out[0].height, out[0].pixels = capi.IMAGE_RGBA32_FLOAT, w.value, h.value, px.ctypes.data
        return 1

    loader = capi.GiCImageLoader(None, capi.IMAGE_LOAD(_load), capi.IMAGE_RELEASE(lambda user, img: None))
    mem = _MemoryAssets({"pkg.usdz[env/sky.png]": blob, "pkg.usdz[env/sky.exr]": b"EXR!" + b"\0" * 60})
    L.giCRegisterAssetReader(C.byref(mem.struct)); L.giCSetImageLoader(C.byref(loader))
    try:
        via_reader = dome_render("pkg.usdz[env/sky.png]")
        via_loader = dome_render("pkg.usdz[env/sky.exr]")
        sc = gi.Scene(SceneDesc(camera=cam))
        try:
            t1 = L.giCCreateTextureFromFile(sc.handle, b"pkg.usdz[env/sky.png]", 1)
            t2 = L.giCCreateTextureFromFile(sc.handle, b"pkg.usdz[env/sky.png]", 1)
            assert t1 and t1 == t2 and not L.giCCreateTextureFromFile(sc.handle, b"pkg.usdz[env/none.png]", 1)
            L.giCDestroyTexture(t1); L.giCDestroyTexture(t2)
        finally:
            sc.close()
    finally:
        L.giCSetImageLoader(None); L.giCRegisterAssetReader(None)
    assert np.array_equal(via_reader.view(np.uint32), from_file.view(np.uint32))
    assert np.array_equal(via_loader.view(np.uint32), from_file.view(np.uint32))
    assert seen == [("pkg.usdz[env/sky.exr]", 1)] and not mem.open_assets


def test_file_textures_are_shared_by_path(gi, tmp_path):
    """giCCreateTextureFromFile hands out ONE texture per (path, colour space) while it is alive -- the reference's weak-pointer
    file cache (TextureManager.cpp:100-150) -- and giCDestroyTexture releases one reference at a time."""
    from gatling_amd.scene import SceneDesc
    from test_capi_host import _write_png
    img = np.full((2, 2, 3), 128, np.uint8)
    _write_png(tmp_path / "a.png", img, 2, 8, [0, 1])
    sc = gi.Scene(SceneDesc())
    try:
        L, path = sc.L, str(tmp_path / "a.png").encode()
        t1 = L.giCCreateTextureFromFile(sc.handle, path, 1)
        t2 = L.giCCreateTextureFromFile(sc.handle, path, 1)
        t3 = L.giCCreateTextureFromFile(sc.handle, path, 0)  # other colour space: other pixels
        assert t1 and t1 == t2 and t3 and t3 != t1
        L.giCDestroyTexture(t1)                      # one of two references
        assert L.giCCreateTextureFromFile(sc.handle, path, 1) == t2
        L.giCDestroyTexture(t2); L.giCDestroyTexture(t2); L.giCDestroyTexture(t3)  # last references: freed
    finally:
        sc.close()


@pytest.mark.parametrize("stack,nee", [(1, True), (2, True), (2, False), (4, True), (12, True), (15, False)])
def test_volume_medium_stack_parity(gi, orc, stack, nee):
    """Participating media with a medium stack (rp_main.rgen:48-97, 317-346, 462-477; rp_main.miss:16-34; rp_main.chit:160-186,
    447-480): distance sampling, Henyey-Greenstein random walk, nested dielectrics with relative IOR -- bit-identical to the
    oracle, segment counts included (scattering events are segments without a surface hit)."""
    desc = volume_scene()
    rs = RenderSettings(spp=4, max_bounces=12, next_event_estimation=nee, medium_stack_size=stack)
    render_both(gi, orc, desc, rs, 96, 54)


@pytest.mark.parametrize("scene_kind", ["soup", "instances"])
def test_medium_stack_in_scenes_beyond_lds(gi, orc, scene_kind):
    """Medium stacks in scenes the persistent-wave traversal walks (k_trace_dyn + k_route: a segment that ends inside a medium is a scattering event routed to
    k_shade<OpenPBR, VOLUME>; the miss record carries (tMax, origin)): a scattering dielectric in a 20 000-triangle soup / on instanced spheres, NEE on and off."""
    desc = _soup(20000, seed=7) if scene_kind == "soup" else sphere_grid(grid=4, subdivisions=2, material_count=4)
    p = np.array(desc.materials[0].params, np.float32, copy=True)
    p[23] = 1.0; p[24:27] = (0.8, 0.9, 0.7); p[28] = 0.5; p[29:32] = (0.3, 0.3, 0.3)  # transmission weight, colour, depth, scatter: an absorbing, scattering medium
    desc.materials[0].params = p
    for stack, nee in ((2, True), (4, False)):
        rs = RenderSettings(spp=3, max_bounces=8, next_event_estimation=nee, medium_stack_size=stack)
        img, ref, st = render_both(gi, orc, desc, rs, 96, 54)
        assert st["fusedPath"] == 0


def test_volume_stack_of_one_equals_toggle(gi):
    """One absorbing, non-scattering, un-nested medium: mediumStackSize 1 reproduces the inside/outside toggle exactly."""
    desc = volume_scene(scatter=(0, 0, 0), nested=False)
    sc = gi.Scene(desc)
    try:
        a = sc.render(RenderSettings(spp=4, max_bounces=10, progressive_accumulation=False), 96, 54)
        b = sc.render(RenderSettings(spp=4, max_bounces=10, progressive_accumulation=False, medium_stack_size=1), 96, 54)
    finally:
        sc.close()
    assert np.array_equal(a, b)


@pytest.mark.parametrize("with_color", [True, False])
def test_nee_and_bounces_aovs(gi, orc, with_color):
    """The two debug AOVs that follow whole paths.  NEE (rp_main.rgen:431-435): written at bounce 0 ONLY, for every sample (the last
    one wins), red if that bounce's shadow ray was traced and occluded, green otherwise -- an untraced shadow ray is dispatched with
    an empty interval, misses, and counts as "not shadowed", so with NEE on no pixel keeps the clear value.  Bounces (:483-486):
    inferno colour of the last sample's bounce count.  Equal to the oracle although the wavefront loop retires samples out of order."""
    desc = cornell_box()
    desc.rect_lights = [RectLight(origin=(0, 0, 0.9), t0=(1, 0, 0), t1=(0, -1, 0), base_emission=(10, 10, 10), width=0.7, height=0.5)]
    rs = RenderSettings(spp=5, max_bounces=7, next_event_estimation=True)
    clear = {"nee": (0.25, 0.5, 0.75, 0.0), "bounces": (0.0, 0.0, 0.0, 0.0)}
    names = ["nee", "bounces", "clockCycles"]
    ref = orc.render_aovs(desc, rs, 80, 45, names, clear_values=clear)
    sc = gi.Scene(desc)
    try:
        got = sc.render_aovs(rs, 80, 45, names, clear_values=clear, with_color=with_color)
    finally:
        sc.close()
    for k in ("nee", "bounces"):
        assert np.array_equal(got[k][..., :3], ref[k][..., :3]), k
    # ClockCycles: the per-pixel cost (here: ray segments of all samples, a deterministic proxy for the reference's shader clock) as a
    # Turbo heat map normalised to the frame maximum, alpha 255 (_EncodeRenderBufferAsHeatmap, Gi.cpp:327-343)
    assert np.array_equal(got["clockCycles"], ref["clockCycles"])
    assert (got["clockCycles"][..., 3] == 255.0).all() and len(np.unique(got["clockCycles"][..., :3].reshape(-1, 3), axis=0)) > 8
    kinds = {tuple(v) for v in np.unique(ref["nee"][..., :3].reshape(-1, 3), axis=0).tolist()}
    assert kinds == {(1.0, 0.0, 0.0), (0.0, 1.0, 0.0)}  # both outcomes occur, and nothing else: every pixel's bounce 0 writes the AOV


def test_scene_data_primvar_inputs(gi, orc):
    """scene_data_lookup_float3 / _float (mdl_interface.glsl:281-301, 337-424) through the C ABI: material inputs driven by named
    primvars with vertex, uniform, instance and constant interpolation, instancer primvars overridden by mesh primvars, a mesh
    without the primvar keeping the constant, a short array reading zeros -- bit-identical to the oracle."""
    from gatling_amd.scene import (INTERP_CONSTANT, INTERP_INSTANCE, INTERP_UNIFORM, INTERP_VERTEX, PRIMVAR_FLOAT, PRIMVAR_VEC3, PRIMVAR_VEC4, Primvar,
                                   TEX_BASE_COLOR, TEX_EMISSION, TEX_METALLIC, TEX_ROUGHNESS)
    rng = np.random.default_rng(21)
    desc = sphere_grid(grid=3, subdivisions=1, material_count=3)
    for mat in desc.materials:
        mat.primvar_inputs = {TEX_BASE_COLOR: "displayColor", TEX_ROUGHNESS: "rough", TEX_METALLIC: "metal", TEX_EMISSION: "glow"}
    for k, m in enumerate(desc.meshes):
        nv, nf, ni = len(m.vertices), len(m.faces), len(m.instance_transforms)
        if k == 0:
            m.primvars = [Primvar("displayColor", PRIMVAR_VEC3, INTERP_VERTEX, rng.uniform(0, 1, (nv, 3))),
                          Primvar("rough", PRIMVAR_FLOAT, INTERP_UNIFORM, rng.uniform(0.1, 0.9, nf)),
                          Primvar("glow", PRIMVAR_VEC4, INTERP_CONSTANT, np.float32([[0.2, 0.1, 0.0, 1.0]]))]
            m.instancer_primvars = [Primvar("metal", PRIMVAR_FLOAT, INTERP_INSTANCE, rng.uniform(0, 1, ni))]
        elif k == 1:
            m.instancer_primvars = [Primvar("displayColor", PRIMVAR_VEC3, INTERP_INSTANCE, rng.uniform(0, 1, (ni, 3))),
                                    Primvar("rough", PRIMVAR_FLOAT, INTERP_INSTANCE, np.full(ni, 0.9, np.float32))]
            m.primvars = [Primvar("rough", PRIMVAR_FLOAT, INTERP_VERTEX, rng.uniform(0.05, 0.5, nv - 7))]  # overrides the instancer's; 7 entries short
        # k == 2: no primvars at all -> every input keeps its constant
    rs = RenderSettings(spp=4, max_bounces=6)
    render_both(gi, orc, desc, rs, 96, 54)
    plain = sphere_grid(grid=3, subdivisions=1, material_count=3)
    sc = gi.Scene(plain)
    try:
        ref_plain = sc.render(rs, 96, 54)
    finally:
        sc.close()
    got, _, _ = render_both(gi, orc, desc, rs, 96, 54)
    assert not np.array_equal(got, ref_plain)


def test_scene_data_int_primvars_and_named_ubo_values(gi, orc):
    """The rest of the scene-data runtime: integer primvars (scene_data_lookup_int / _int3, mdl_interface.glsl:426-476: the NEAREST vertex's
    value, no blending) and the two names answered from the UBO instead of a buffer -- CAMERA_POSITION for float3 lookups (:329-334) and
    FRAME for float lookups (:390-395; GiRenderSettings.frame) -- bit-identical to the oracle, and actually reaching the image."""
    from gatling_amd.scene import (INTERP_UNIFORM, INTERP_VERTEX, PRIMVAR_INT, PRIMVAR_INT3, Primvar, TEX_BASE_COLOR, TEX_EMISSION, TEX_METALLIC, TEX_ROUGHNESS)
    rng = np.random.default_rng(33)
    desc = sphere_grid(grid=3, subdivisions=1, material_count=3)
    desc.materials[0].primvar_inputs = {TEX_BASE_COLOR: "paletteIndex", TEX_METALLIC: "isMetal"}
    desc.materials[1].primvar_inputs = {TEX_EMISSION: "CAMERA_POSITION", TEX_ROUGHNESS: "FRAME"}
    desc.materials[2].primvar_inputs = {TEX_BASE_COLOR: "CAMERA_POSITION", TEX_ROUGHNESS: "isMetal"}  # int primvar read by a float input
    for m in desc.meshes:
        nv, nf = len(m.vertices), len(m.faces)
        m.primvars = [Primvar("paletteIndex", PRIMVAR_INT3, INTERP_VERTEX, rng.integers(0, 2, (nv, 3))),
                      Primvar("isMetal", PRIMVAR_INT, INTERP_UNIFORM, rng.integers(0, 2, nf))]
    imgs = []
    for frame in (0.25, 0.75):
        rs = RenderSettings(spp=4, max_bounces=5, frame=frame)
        got, _, _ = render_both(gi, orc, desc, rs, 96, 54)
        imgs.append(got)
    assert not np.array_equal(imgs[0], imgs[1])  # FRAME drives a roughness
    # nearest, not blended: a vertex-interpolated 0/1 integer colour only ever yields the corner colours at the primary hit
    alb = orc.render_aovs(desc, RenderSettings(spp=1, max_bounces=1, jittered_sampling=False), 96, 54, ["albedo"])["albedo"]
    assert np.isfinite(alb).all()


def test_colour_inputs_reading_narrow_primvars(gi, orc):
    """A three-component input bound to a one- or two-component primvar: the reference's float3 lookup applies the primvar's own stride and reads the next two
    floats of the packed buffer (mdl_interface.glsl:343-349) -- for the last entry whatever follows the array.  Here, on both sides, what lies past an array is zero
    (gi_build.cpp pads it): a constant float makes the colour (x, 0, 0), a vertex-interpolated float2 blends (x, y, next x).  Found by the differential campaign of
    round 6 (tests/fuzz_parity.py, profiles/r06w_fuzz_parity_extended_first_run.log): the packed neighbour showed through."""
    from gatling_amd.scene import (INTERP_CONSTANT, INTERP_INSTANCE, INTERP_VERTEX, PRIMVAR_FLOAT, PRIMVAR_VEC2, PRIMVAR_VEC3, Primvar, TEX_BASE_COLOR, TEX_EMISSION,
                                   TEX_ROUGHNESS)
    rng = np.random.default_rng(71)
    desc = sphere_grid(grid=3, subdivisions=1, material_count=3)
    desc.materials[0].primvar_inputs = {TEX_BASE_COLOR: "narrow", TEX_ROUGHNESS: "rough"}
    desc.materials[1].primvar_inputs = {TEX_EMISSION: "narrow", TEX_BASE_COLOR: "pair"}
    desc.materials[2].primvar_inputs = {TEX_BASE_COLOR: "perInstance"}
    for m in desc.meshes:
        nv, ni = len(m.vertices), len(m.instance_transforms)
        m.primvars = [Primvar("narrow", PRIMVAR_FLOAT, INTERP_CONSTANT, np.float32([0.6])),              # packed first: the next array is its neighbour
                      Primvar("pair", PRIMVAR_VEC2, INTERP_VERTEX, rng.uniform(0.1, 0.9, (nv, 2))),
                      Primvar("rough", PRIMVAR_VEC3, INTERP_VERTEX, rng.uniform(0.2, 0.8, (nv, 3)))]
        m.instancer_primvars = [Primvar("perInstance", PRIMVAR_FLOAT, INTERP_INSTANCE, rng.uniform(0.2, 0.9, ni))]
    for nee in (False, True):
        render_both(gi, orc, desc, RenderSettings(spp=4, max_bounces=4, next_event_estimation=nee), 96, 54)
    alb = orc.render_aovs(desc, RenderSettings(spp=1, max_bounces=1, jittered_sampling=False), 96, 54, ["albedo"])["albedo"]
    sc = gi.Scene(desc)
    try:
        got = sc.render_aovs(RenderSettings(spp=1, max_bounces=1, jittered_sampling=False), 96, 54, ["albedo"], with_color=False)["albedo"]
    finally:
        sc.close()
    assert np.array_equal(got[..., :3], alb.reshape(got.shape)[..., :3])


def test_everything_at_once_parity(gi, orc):
    """Feature interactions: textured + primvar-driven inputs, normal map, dome light, medium stack with scattering, cutouts, all four
    light types with NEE, depth of field and clipping planes in ONE scene (the TEXTURED x VOLUME x DOME kernel specialisations),
    across two render calls (progressive accumulation) -- bit-identical to the oracle."""
    from gatling_amd.scene import INTERP_UNIFORM, PRIMVAR_VEC3, Primvar, TEX_BASE_COLOR, TEX_EMISSION, TextureBinding
    desc = textured_scene(dome=True)
    vol = volume_scene()
    base = len(desc.materials)
    desc.materials += vol.materials[1:]                    # murky + clear glass
    for m in vol.meshes[1:]:
        m.material += base - 1
        m.transform = m.transform.copy(); m.transform[3, 0] += 1.6
        desc.meshes.append(m)
    cut = MaterialDesc.usd_preview_surface(name="leaf", diffuseColor=(0.2, 0.7, 0.2), opacity=0.5)
    cut.primvar_inputs = {TEX_BASE_COLOR: "displayColor"}
    desc.materials.append(cut)
    from gatling_amd.meshprep import bake_vertices
    qp = np.float32([[-2.2, -1.0, 0.2], [-1.2, -1.0, 0.2], [-1.2, -1.0, 1.6], [-2.2, -1.0, 0.2], [-1.2, -1.0, 1.6], [-2.2, -1.0, 1.6]])
    quad = MeshDesc(name="/Leaf", vertices=bake_vertices(qp, np.tile([0, -1, 0], (6, 1))), faces=np.arange(6, dtype=np.uint32).reshape(-1, 3),
                    material=len(desc.materials) - 1, id=7, double_sided=True)
    quad.primvars = [Primvar("displayColor", PRIMVAR_VEC3, INTERP_UNIFORM, np.float32([[0.9, 0.3, 0.1], [0.1, 0.3, 0.9]]))]
    desc.meshes.append(quad)
    desc.sphere_lights = [SphereLight(pos=(-1.5, -1.5, 2.5), base_emission=(6, 5, 4), radius=(0.2, 0.2, 0.2))]
    desc.distant_lights = [DistantLight(direction=(0.3, 0.4, -0.85), base_emission=(1.5, 1.5, 1.2), angle=0.05)]
    desc.disk_lights = [DiskLight(origin=(1.5, 1.0, 3.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(8, 8, 8), radius_x=0.4, radius_y=0.3)]
    desc.camera.f_stop = 2.0; desc.camera.focus_distance = 5.0; desc.camera.focal_length = 0.05
    desc.camera.clip_start = 0.5; desc.camera.clip_end = 40.0
    rs = RenderSettings(spp=3, max_bounces=10, next_event_estimation=True, medium_stack_size=2, depth_of_field=True, clipping_planes=True)
    img1, ref1, _ = render_both(gi, orc, desc, rs, 96, 54)
    sc = gi.Scene(desc)
    try:
        sc.render(rs, 96, 54)
        img2 = sc.render(rs, 96, 54)  # second call accumulates on the first
    finally:
        sc.close()
    ref2, _ = orc.render(desc, rs, 96, 54, sample_offset=3, prev_color=ref1, threads=4)
    assert np.array_equal(img2, ref2)


def test_interior_scene_parity(gi, orc):
    """C5's structure at small scale: room + instanced clutter (affine instance transforms), three material classes incl.
    transmission, four rect lights, NEE on."""
    desc = interior_scene(clutter_instances=60, subdivisions=1, prototypes=4, material_count=12)
    render_both(gi, orc, desc, RenderSettings(spp=3, max_bounces=6, next_event_estimation=True), 96, 54)


@pytest.mark.parametrize("scene_kind", ["soup", "instances"])
def test_traversal_kernel_variants_agree(gi, orc, scene_kind):
    """Scenes beyond LDS: the block-synchronous k_trace, k_trace_dyn (any refill threshold) and the oracle give the
    same image bit for bit -- the kernels differ in scheduling (who tests which triangle when), never in arithmetic."""
    from gatling_amd import capi
    if scene_kind == "soup":
        desc, rs, w, h = _soup(6000, seed=5), RenderSettings(spp=3, max_bounces=6, next_event_estimation=True), 80, 45
    else:
        desc, rs, w, h = sphere_grid(grid=3, subdivisions=2, material_count=4), RenderSettings(spp=3, max_bounces=6), 80, 45
    rs.progressive_accumulation = False
    ref, _ = orc.render(desc, rs, w, h, threads=4)
    scene = gi.Scene(desc)
    try:
        for refill in (0, 1, 8, 33, 64):
            scene.set_option(capi.OPTION_TRACE_DYNAMIC, refill)
            img = scene.render(rs, w, h)
            assert np.array_equal(img, ref), f"refill={refill}"
    finally:
        scene.close()


@pytest.mark.parametrize("scene_kind", ["instances", "interior", "instances+cutouts+nee"])
def test_two_level_layout_parity(gi, orc, scene_kind):
    """Instanced scenes beyond LDS: the two-level layout (TLAS over instances + one object-space BLAS per mesh, candidates rebuilt
    in world space with the host's arithmetic) gives the flat layout's image and the oracle's, bit for bit -- closest-hit and shadow rays,
    affine (sheared, non-uniformly scaled) instance transforms, cutouts."""
    from gatling_amd import capi
    if scene_kind == "instances":
        desc, rs = sphere_grid(grid=5, subdivisions=2, material_count=6), RenderSettings(spp=3, max_bounces=6)
    elif scene_kind == "interior":
        desc, rs = interior_scene(clutter_instances=80, subdivisions=1, prototypes=5, material_count=10), RenderSettings(spp=3, max_bounces=6, next_event_estimation=True)
    else:
        desc = sphere_grid(grid=4, subdivisions=1, material_count=4)
        desc.materials[1].params[14] = 0.4   # stochastic cutout
        desc.rect_lights = [RectLight(origin=(0, 0, 6.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(15, 15, 15), width=3.0, height=3.0)]
        rs = RenderSettings(spp=3, max_bounces=5, next_event_estimation=True, rr_bounce_offset=0)
    rs.progressive_accumulation = False
    w, h = 96, 54
    ref, _ = orc.render(desc, rs, w, h, threads=4)
    scene = gi.Scene(desc)
    try:
        scene.set_option(capi.OPTION_COUNT_TRAVERSAL, 1)
        scene.set_option(capi.OPTION_TWO_LEVEL, 0)
        flat = scene.render(rs, w, h); st_flat = scene.stats()
        scene.set_option(capi.OPTION_TWO_LEVEL, 1)
        two = scene.render(rs, w, h); st_two = scene.stats()
    finally:
        scene.close()
    assert np.array_equal(flat, ref)
    assert np.array_equal(two, ref)
    assert st_two["segments"] == st_flat["segments"] and st_two["shadowRays"] == st_flat["shadowRays"]
    assert st_two["nodesVisited"] != st_flat["nodesVisited"]  # it really was the other traversal


# ---------------------------------------------------------------------------------------------------------------
# full BASELINE.json sizes, against the checker: the oracle renders these on all host cores (C1 in full, as BASELINE.md 3.1 says;
# C2 - C5 at the full resolution and the real scene at a reduced spp -- RNG streams, traversal and shading do not depend on spp)
# ---------------------------------------------------------------------------------------------------------------
_CORES = os.cpu_count() or 4


def test_c1_full_baseline_bit_exact(gi, orc):
    """BASELINE config C1 in full: cornell.usda 512x512, spp 64, max-bounces 4, diffuse-only -- 16.8 M samples, bit for bit."""
    img, ref, st = render_both(gi, orc, cornell_box(MAT_DIFFUSE), RenderSettings(spp=64, max_bounces=4), 512, 512, exact=True, threads=_CORES)
    assert st["samples"] == 512 * 512 * 64


def test_c2_baseline_1080p_bit_exact(gi, orc):
    """BASELINE config C2 at its resolution (1920x1080, 8 bounces, UsdPreviewSurface) and spp 8 against the oracle, plus the
    size-independent properties: run-to-run determinism, the 8-GPU row partition (contiguous bands and interleaved rows) stitched ==
    whole frame, a checksum of checksums, the per-sample clamp bound."""
    desc = cornell_box()
    rs = RenderSettings(spp=8, max_bounces=8, progressive_accumulation=False)
    w, h = 1920, 1080
    sc = gi.Scene(desc)
    try:
        full = sc.render(rs, w, h).copy()
        st = sc.stats()
        again = sc.render(rs, w, h).copy()
        bands = [sc.render(rs, w, h, rows=(r * 135, (r + 1) * 135)).copy() for r in range(8)]
        inter = [sc.render(rs, w, h, rows=(r, h), row_stride=8).copy() for r in (0, 5)]
    finally:
        sc.close()
    ref, cnt = orc.render(desc, rs, w, h, threads=_CORES)
    assert st["segments"] == cnt["segments"] and st["samples"] == cnt["samples"] == w * h * 8
    assert_image_parity(full, ref, exact=True)
    assert np.array_equal(full.view(np.uint32), again.view(np.uint32))  # determinism despite atomics in the queues
    assert np.array_equal(np.concatenate(bands).view(np.uint32), full.view(np.uint32))
    for r, part in zip((0, 5), inter):
        assert np.array_equal(part.view(np.uint32), full[r::8].view(np.uint32))
    assert sum(int(b.view(np.uint32).astype(np.uint64).sum()) for b in bands) == int(full.view(np.uint32).astype(np.uint64).sum())
    assert (full[..., 3] == 1.0).all() and full[..., :3].min() >= 0.0 and full[..., :3].max() <= rs.max_sample_value * (1 + 1e-5)
    assert 0.55 < float((full[..., :3] == 1.0).all(axis=-1).mean()) < 0.70  # the open front shows the quantised clear colour


def test_c3_baseline_1080p_bit_exact(gi, orc):
    """BASELINE config C3 with the real scene (1 000 000-triangle soup, seed 1234, one OpenPBR material, rect light, NEE) at
    1920x1080, spp 2: image, segment and shadow-ray counts equal the oracle's (independent BVH2 traversal)."""
    desc = random_triangle_soup(1_000_000)
    img, ref, st = render_both(gi, orc, desc, RenderSettings(spp=2, max_bounces=8, next_event_estimation=True), 1920, 1080, exact=True, threads=_CORES)
    assert st["triangleCount"] == 1_000_000 and st["shadowRays"] > 0


def test_c4_baseline_1080p_bit_exact(gi, orc):
    """BASELINE config C4 with the real scene (32x32 instanced icospheres = 5 242 880 triangles, 32 OpenPBR / UsdPreviewSurface
    parameter sets, seed 4321) at 1920x1080, spp 2."""
    desc = sphere_grid(32, 4, 32)
    img, ref, st = render_both(gi, orc, desc, RenderSettings(spp=2, max_bounces=8), 1920, 1080, exact=True, threads=_CORES)
    assert st["triangleCount"] == 1024 * 5120


def test_c5_baseline_band_bit_exact(gi, orc):
    """BASELINE config C5 at full geometric size (10.24 M instanced triangles, 51 materials, 4 rect lights, NEE, 3840x2160): the
    270-row band rank 3 of the 8-GPU partition at spp 1 equals the oracle's render of those rows, equals the same rows of a
    whole-frame render, and so do bands 0 and 7 (the per-pixel RNG streams use the global pixel index)."""
    from gatling_amd.dist import partition_rows
    desc = interior_scene()
    rs = RenderSettings(spp=1, max_bounces=8, next_event_estimation=True, progressive_accumulation=False)
    w, h = 3840, 2160
    sc = gi.Scene(desc)
    try:
        full = sc.render(rs, w, h).copy()
        st = sc.stats()
        assert st["triangleCount"] == 2000 * 5120 + 12
        for rank in (0, 3, 7):
            r0, r1 = partition_rows(h, 8, rank)
            assert r1 - r0 == 270
            band = sc.render(rs, w, h, rows=(r0, r1)).copy()
            bst = sc.stats()
            assert np.array_equal(band.view(np.uint32), full[r0:r1].view(np.uint32)), f"band {rank}"
            if rank == 3:
                ref, cnt = orc.render(desc, rs, w, h, rows=(r0, r1), threads=_CORES)
                assert bst["segments"] == cnt["segments"] and bst["shadowRays"] == cnt["shadow_rays"]
                assert_image_parity(band, ref, exact=True)
    finally:
        sc.close()
    assert np.isfinite(full).all() and full[..., :3].mean() > 0.01


def test_small_pool_three_material_classes(gi, orc):
    """Queue sizing (shardCapacity): a pool of ~1-2 k slots with all three material classes present feeds TRACE / REGEN from up to
    five launches that all start dealing blocks at shard 0; the image must still equal the oracle's and no shard may overflow
    (giCRender would fail).  The pool is forced small with the scene option."""
    desc = sphere_grid(grid=3, subdivisions=1, material_count=6)
    desc.materials[0] = MaterialDesc.usd_preview_surface(name="lambert", diffuseColor=(0.7, 0.3, 0.2), klass=MAT_DIFFUSE)
    assert {m.klass for m in desc.materials} == {0, 1, 2}
    rs = RenderSettings(spp=2, max_bounces=6)
    ref, cnt = orc.render(desc, rs, 32, 32, threads=4)
    for pool in (1024, 1536, 2048):
        sc = gi.Scene(desc)
        try:
            sc.set_option(gi.OPTION_POOL_SLOTS, pool)
            img = sc.render(rs, 32, 32)
            st = sc.stats()
        finally:
            sc.close()
        assert st["segments"] == cnt["segments"]
        assert_image_parity(img, ref, exact=True)


def test_max_bounces_zero_is_black(gi, orc):
    """rp_main.rgen:298-304: the bounce loop tests `bounce >= maxBounces` before tracing, so max-bounces 0 traces nothing: black
    samples (no emission, no background term), alpha 1, zero segments."""
    img, ref, st = render_both(gi, orc, cornell_box(), RenderSettings(spp=3, max_bounces=0), 48, 27, exact=True)
    assert st["segments"] == 0 and (img[..., :3] == 0.0).all() and (img[..., 3] == 1.0).all()


def test_full_size_furnace(gi):
    """Energy conservation at full HD: closed emitting Lambertian box, radiance = sum albedo^k."""
    from test_oracle_render import _furnace
    albedo, bounces = 0.5, 8
    rs = RenderSettings(spp=4, max_bounces=bounces, rr_bounce_offset=100, max_sample_value=1e9)
    sc = gi.Scene(_furnace(albedo))
    try:
        img = sc.render(rs, 1920, 1080)
        st = sc.stats()
    finally:
        sc.close()
    assert st["segments"] == 1920 * 1080 * 4 * bounces
    np.testing.assert_allclose(img[..., :3], sum(albedo ** k for k in range(bounces)), rtol=1e-5)


def test_thin_walled_subsurface_scene_parity(gi, orc):
    """Leaf cards with a thin-walled subsurface material (half of the subsurface share is diffusely transmitted: light reaches the floor behind the cards) lit by
    a rect light with NEE: image and counters bit-identical to the oracle; the cards must let light through."""
    from gatling_amd.scenes import leaf_card_scene
    imgs = {}
    for w in (0.0, 0.9):
        desc = leaf_card_scene()
        for m in desc.materials:
            if m.klass == 2:  # the cards' OpenPBR material
                m.params[54] = 1.0; m.params[55] = w; m.params[56:59] = (0.35, 0.85, 0.25); m.params[59] = 0.3  # thin-walled, subsurface weight / colour / anisotropy
        rs = RenderSettings(spp=4, max_bounces=6, next_event_estimation=True)
        ref, cnt = orc.render(desc, rs, 80, 45, threads=8)
        sc = gi.Scene(desc)
        try:
            img = sc.render(rs, 80, 45)
            st = sc.stats()
        finally:
            sc.close()
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), w
        assert st["segments"] == cnt["segments"] and st["shadowRays"] == cnt["shadow_rays"]
        imgs[w] = img
    assert not np.array_equal(imgs[0.0], imgs[0.9])


def test_fuzz_layer_scene_parity(gi, orc):
    """Spheres with OpenPBR's optional lobes -- fuzz over dark and over coated bases, brushed (anisotropic) highlights, a thin film -- next to UsdPreviewSurface
    ones, lit by a rect light and the uniform dome, NEE on and off: the wavefront pipeline's image == the oracle's, bit for bit (the per-material feature word
    the host derives decides which inputs k_shade loads), and the lobes visibly change the image."""
    desc = sphere_grid(grid=3, subdivisions=2, material_count=6)
    fz = [(1.0, (0.9, 0.4, 0.3), 0.5), (0.8, (0.7, 0.8, 1.0), 0.1), (0.5, (1.0, 1.0, 1.0), 1.0)]
    k = 0
    for m in desc.materials:
        if m.klass == MAT_OPEN_PBR and k < len(fz):
            w, c, r = fz[k]
            m.params[49] = w; m.params[50:53] = c; m.params[53] = r
            if k == 0: m.params[60] = 0.8; m.params[61] = 0.5                              # specular / coat roughness anisotropy
            if k == 1: m.params[62] = 1.0; m.params[63] = 0.35; m.params[6] = 1.7         # thin film: weight, thickness (um), ior
            if k == 2: m.params[62] = 0.6; m.params[63] = 0.5; m.params[6] = 1.33; m.params[60] = 0.4
            k += 1
    assert k >= 2
    desc.rect_lights.append(RectLight(origin=(0.0, -3.0, 4.0), t0=(1, 0, 0), t1=(0, 0.8, 0.6), base_emission=(14, 13, 12), width=2.0, height=2.0))
    for nee in (False, True):
        img, _, _ = render_both(gi, orc, desc, RenderSettings(spp=3, max_bounces=5, next_event_estimation=nee), 96, 54)
    plain = sphere_grid(grid=3, subdivisions=2, material_count=6)
    plain.rect_lights = list(desc.rect_lights)
    sc = gi.Scene(plain)
    try:
        ref = sc.render(RenderSettings(spp=3, max_bounces=5, next_event_estimation=True), 96, 54)
    finally:
        sc.close()
    assert not np.array_equal(img, ref)


def test_remaining_texture_entry_points_on_device(gi):
    """tex_texel_float4_2d, tex_resolution_2d, tex_lookup_float4_3d, tex_texel_float4_3d (mdl_interface.glsl:45-65, 86-105, 167-221) on the device == the oracle,
    which tests/test_oracle_ref.py holds to the reference's text: in range, outside, the invalid texture, every wrap mode."""
    import ctypes as C
    from oracle import orc as O
    from test_oracle_ref import _tex_runtime_queries
    rng = np.random.default_rng(13)
    w, h, d, n = 6, 5, 4, 20000
    vol = np.ascontiguousarray(rng.uniform(0, 1, (d, h, w, 4)).astype(np.float32))
    q = _tex_runtime_queries(rng, w, h, d, n)
    got, ref = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
    FP = C.POINTER(C.c_float)
    L = gi.load_library()
    assert L.giCDebugTexRuntime(vol.ctypes.data_as(FP), w, h, d, n, q.ctypes.data_as(FP), got.ctypes.data_as(FP)) == gi.GI_C_OK
    lib = O.lib()
    lib.orc_tex_runtime.argtypes = [FP, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, FP, FP]
    lib.orc_tex_runtime(vol.ctypes.data_as(FP), w, h, d, n, q.ctypes.data_as(FP), ref.ctypes.data_as(FP))
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert np.abs(got).sum() > 0


# ---------------------------------------------------------------------------------------------------------------
# k_shade variants (round 6): OpenPBR materials without optional lobes are binned apart and shaded by the BASE variant (gi_shading.h) -- the reference's
# per-material feature #defines (GlslShaderGen.cpp:204-274, two hit groups per material Gi.cpp:1545-1562) done the wavefront way.  Same bits either way.
# ---------------------------------------------------------------------------------------------------------------
def test_base_variant_closed_forms_equal_the_full_ones(gi, orc, monkeypatch):
    """giCDebugEvalBsdf runs the variant a material is binned for: random BASE parameter sets (metalness 0 .. 1, rough / smooth, Oren-Nayar, weights, ior, front and
    back faces) through the BASE variant, through the full closed form (shade_variants=0) and through the oracle -- bit for bit, sample and evaluate."""
    from gatling_amd import capi
    from test_oracle_render import _frames
    rng = np.random.default_rng(606)
    for k in range(24):
        m = MaterialDesc.open_pbr(base_color=tuple(rng.uniform(0, 1, 3)), base_weight=float(rng.uniform(0, 1)), base_metalness=float(rng.choice([0.0, 1.0, rng.uniform(0, 1)])),
                                  specular_weight=float(rng.choice([0.0, 1.0, rng.uniform(0, 1)])), specular_color=tuple(rng.uniform(0, 1, 3)),
                                  specular_roughness=float(rng.choice([0.0, 0.02, rng.uniform(0, 1)])), specular_ior=float(rng.uniform(1.0, 2.5)),
                                  base_diffuse_roughness=float(rng.choice([0.0, rng.uniform(0, 1)])), emission_luminance=float(rng.choice([0.0, 2.0])),
                                  coat_color=tuple(rng.uniform(0, 1, 3)), coat_roughness=float(rng.uniform(0, 1)), coat_darkening=float(rng.uniform(0, 1)))
        monkeypatch.setenv("GATLING_OPTIONS", "shade_variants=1")
        assert capi.shade_class(m) == 3
        items = _frames(20000, rng, float(rng.uniform(0.05, 1.0)))
        items[::7, 21] = 1.0  # back faces (relative_eta)
        base = gi.bsdf_debug(m, items)
        monkeypatch.setenv("GATLING_OPTIONS", "shade_variants=0")
        full = gi.bsdf_debug(m, items)
        assert np.array_equal(base.view(np.uint32), full.view(np.uint32)), k
        assert np.array_equal(base.view(np.uint32), orc.bsdf_debug(m, items).view(np.uint32)), k


@pytest.mark.parametrize("scene_kind", ["soup", "soup+nee", "grid+nee", "interior+nee", "grid+medium", "textured", "lds"])
def test_shade_variants_are_bit_identical(gi, orc, monkeypatch, scene_kind):
    """Whole renders with the variants on and off (and against the oracle): a one-material BASE soup (C3's shape), C4's mix of UsdPreviewSurface / full / BASE
    OpenPBR sets, C5's interior, a render with a medium stack (BASE hits go through the full VOLUME kernel, scattering events are routed to class 2), a scene whose
    OpenPBR material is textured (never BASE), and an LDS-resident scene through the stage kernels."""
    from gatling_amd import capi
    rs = RenderSettings(spp=3, max_bounces=6, next_event_estimation="nee" in scene_kind, progressive_accumulation=False)
    opts = []
    if scene_kind.startswith("soup"):
        desc = _soup(20000, seed=11)
    elif scene_kind == "grid+nee":
        desc = sphere_grid(grid=6, subdivisions=2, material_count=32)
        desc.rect_lights = [RectLight(origin=(0, 0, 8.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(15, 15, 15), width=4.0, height=4.0)]
    elif scene_kind == "interior+nee":
        desc = interior_scene(clutter_instances=120, subdivisions=1, prototypes=6, material_count=50)
    elif scene_kind == "grid+medium":
        desc = sphere_grid(grid=4, subdivisions=2, material_count=32)
        p = np.array(desc.materials[2].params, np.float32, copy=True)
        p[23] = 1.0; p[24:27] = (0.8, 0.9, 0.7); p[28] = 0.5; p[29:32] = (0.3, 0.3, 0.3)
        desc.materials[2].params = p
        rs.medium_stack_size = 3
    elif scene_kind == "textured":
        desc = textured_scene()
    else:
        desc = cornell_box()
        desc.materials = [MaterialDesc.open_pbr(name="light", emission_luminance=1.0, emission_color=(8.5, 6, 4)), MaterialDesc.open_pbr(name="white"),
                          MaterialDesc.open_pbr(name="red", base_color=(1, 0, 0), base_metalness=0.5), MaterialDesc.open_pbr(name="green", base_color=(0, 1, 0), coat_weight=0.5)]
        opts = [(capi.OPTION_FUSED_PATH, 0)]
    classes = {capi.shade_class(m) for m in desc.materials}
    if scene_kind in ("grid+nee", "interior+nee", "grid+medium"):
        assert {2, 3} <= classes
    ref, cnt = orc.render(desc, rs, 96, 54, threads=4)
    imgs = []
    for mode in ("shade_variants=1,merge_shade_variants=0", "shade_variants=1,merge_shade_variants=1", "shade_variants=1", "shade_variants=0"):  # (the third: thin batch, merged by itself)
        monkeypatch.setenv("GATLING_OPTIONS", mode)
        sc = gi.Scene(desc)
        try:
            for k, v in opts:
                sc.set_option(k, v)
            imgs.append(sc.render(rs, 96, 54)); st = sc.stats()
        finally:
            sc.close()
        assert st["segments"] == cnt["segments"] and st["shadowRays"] == cnt["shadow_rays"] and st["fusedPath"] == 0
        assert_image_parity(imgs[-1], ref, exact=True)
    assert all(np.array_equal(imgs[0].view(np.uint32), im.view(np.uint32)) for im in imgs[1:])


# ---------------------------------------------------------------------------------------------------------------
# two streams (round 6): in batches whose work fits the pool the shadow launch of bounce i runs on a second stream beside the closest-hit launch of bounce i + 1
# (gi_render.cpp "two streams"; the reference's default frame is one sample per pixel and giRender call, renderDelegate.cpp:93-110).  Scheduling only: same images.
# ---------------------------------------------------------------------------------------------------------------
_TWO_STREAM_MODES = ["two_stream=0", "two_stream=1", "two_stream=1,two_stream_delay=1", "two_stream=1,two_stream_delay=2"]


@pytest.mark.parametrize("scene_kind", ["soup", "interior", "grid+cutouts", "lds", "one-bounce", "two-level"])
def test_two_stream_iterations_are_invisible(gi, orc, monkeypatch, scene_kind):
    """Every mode -- off, on, on with the main stream held back 0.3 ms per iteration (the shadow launch runs far ahead), on with the second stream held back (the next
    closest-hit launch runs far ahead and k_raygen / k_shade really have to wait) -- gives the oracle's image, segment and shadow-ray counts, on several progressive frames;
    a pool smaller than the batch (several raygen rounds) falls back to the single stream by itself."""
    from gatling_amd import capi
    rs = RenderSettings(spp=2, max_bounces=7, next_event_estimation=True, rr_bounce_offset=1)
    opts = []
    if scene_kind == "soup":
        desc = _soup(20000, seed=21)
    elif scene_kind == "interior":
        desc = interior_scene(clutter_instances=100, subdivisions=1, prototypes=5, material_count=20)
    elif scene_kind == "grid+cutouts":
        desc = sphere_grid(grid=5, subdivisions=2, material_count=8)
        desc.materials[1].params[14] = 0.4
        desc.rect_lights = [RectLight(origin=(0, 0, 7.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(15, 15, 15), width=3.0, height=3.0)]
    elif scene_kind == "lds":
        desc = cornell_box()
        desc.rect_lights = [RectLight(origin=(0, 0, 0.9), t0=(1, 0, 0), t1=(0, -1, 0), base_emission=(10, 10, 10), width=0.7, height=0.5)]
        opts = [(capi.OPTION_FUSED_PATH, 0)]
    elif scene_kind == "one-bounce":
        desc = _soup(5000, seed=22); rs.max_bounces = 1
    else:
        desc = sphere_grid(grid=5, subdivisions=2, material_count=6)
        desc.rect_lights = [RectLight(origin=(0, 0, 7.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(15, 15, 15), width=3.0, height=3.0)]
        opts = [(capi.OPTION_TWO_LEVEL, 1)]
    w, h = 96, 54
    refs, prev = [], None
    for k in range(3):
        r, cnt = orc.render(desc, rs, w, h, sample_offset=k * rs.spp, prev_color=prev, threads=4)
        refs.append((r, cnt)); prev = r
    for mode in _TWO_STREAM_MODES + ["two_stream=1,pool_slots=2048"]:
        monkeypatch.setenv("GATLING_OPTIONS", mode)
        sc = gi.Scene(desc)
        try:
            for k_, v_ in opts:
                sc.set_option(k_, v_)
            for k in range(3):
                img = sc.render(rs, w, h); st = sc.stats()
                assert st["fusedPath"] == 0
                assert st["segments"] == refs[k][1]["segments"] and st["shadowRays"] == refs[k][1]["shadow_rays"], (mode, k)
                assert np.array_equal(img.view(np.uint32), refs[k][0].view(np.uint32)), (mode, k)
        finally:
            sc.close()


def test_two_hundred_one_sample_calls_on_two_streams(gi, orc, monkeypatch):
    """hdGatling's own loop: 200 giRender calls of ONE sample per pixel, 13 bounces, NEE, progressive accumulation -- every one of the 200 frames equals the oracle's
    progressive frame (a scheduling race between the two streams would show as a stray pixel in some frame); the stream that is held back alternates between the calls."""
    desc = _soup(8000, seed=23)
    rs = RenderSettings(spp=1, next_event_estimation=True)   # the delegate's defaults: 13 bounces, Russian roulette from bounce 3
    w, h = 64, 36
    sc = gi.Scene(desc)
    prev = None
    try:
        for k in range(200):
            monkeypatch.setenv("GATLING_OPTIONS", ("two_stream=1", "two_stream=1,two_stream_delay=1", "two_stream=1,two_stream_delay=2")[k % 3] if k % 10 else "two_stream=0")
            img = sc.render(rs, w, h)
            ref, _ = orc.render(desc, rs, w, h, sample_offset=k, prev_color=prev, threads=4)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), k
            prev = ref
    finally:
        sc.close()


def test_two_stream_nee_and_bounces_aovs(gi, orc, monkeypatch):
    """The path-following AOVs (NEE: the shadow test of bounce 0, written by the shadow launch; Bounces; ClockCycles) with the shadow launches on the second stream."""
    desc = _soup(6000, seed=24)
    rs = RenderSettings(spp=3, max_bounces=6, next_event_estimation=True)
    clear = {"nee": (0.25, 0.5, 0.75, 0.0), "bounces": (0.0, 0.0, 0.0, 0.0)}
    names = ["nee", "bounces", "clockCycles"]
    ref = orc.render_aovs(desc, rs, 80, 45, names, clear_values=clear)
    for mode in _TWO_STREAM_MODES:
        monkeypatch.setenv("GATLING_OPTIONS", mode)
        sc = gi.Scene(desc)
        try:
            got = sc.render_aovs(rs, 80, 45, names, clear_values=clear)
        finally:
            sc.close()
        for k in ("nee", "bounces"):
            assert np.array_equal(got[k][..., :3], ref[k][..., :3]), (mode, k)
        assert np.array_equal(got["clockCycles"], ref["clockCycles"]), mode
