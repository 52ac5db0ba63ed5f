"""The flat binary scene file (.gscn) and the plain-C harness that loads it: the writer/reader pair
round-trips every SceneDesc field, tools/gi_render.c compiles as strict C99 against include/gi_c.h and parses what the Python
writer wrote (CPU); on the GPU its image equals the ctypes binding's bit for bit."""
import dataclasses
import os
import subprocess

import numpy as np
import pytest

from gatling_amd.scene import RenderSettings
from gatling_amd.scenefile import load_scene, save_scene
from gatling_amd.scenes import cornell_box, interior_scene, leaf_card_scene, sphere_grid, textured_scene, volume_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "gi_render")


def _build_harness():
    lib_dir = os.path.join(ROOT, "gatling_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "gi_render.c"), "-o", EXE, "-L", lib_dir, "-lgatling_gi", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    return EXE


def _same(a, b, path="scene"):
    if dataclasses.is_dataclass(a):
        for f in dataclasses.fields(a):
            _same(getattr(a, f.name), getattr(b, f.name), path + "." + f.name)
    elif isinstance(a, dict):
        assert a.keys() == b.keys(), path
        for k in a:
            _same(a[k], b[k], f"{path}[{k}]")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    elif isinstance(a, np.ndarray):
        assert np.array_equal(a, np.asarray(b).reshape(a.shape)), path
    elif isinstance(a, float):
        assert np.float32(a) == np.float32(b), path
    else:
        assert a == b, path


SCENES = {
    "cornell": lambda: cornell_box(),
    "textured+dome": lambda: textured_scene(dome=True),
    "volume": lambda: volume_scene(),
    "leaf cards": lambda: leaf_card_scene(),
    "instances": lambda: sphere_grid(grid=3, subdivisions=1, material_count=4),
    "interior": lambda: interior_scene(clutter_instances=12, subdivisions=1, prototypes=3, material_count=6),
}


@pytest.mark.parametrize("name", sorted(SCENES))
def test_gscn_round_trip(tmp_path, name):
    desc = SCENES[name]()
    # (every field away from its default: `frame` -- the FRAME scene-data value -- was written as 0 until the differential campaign's scenes went through the file)
    rs = RenderSettings(spp=3, max_bounces=5, rr_bounce_offset=2, rr_inv_min_term_prob=0.75, max_sample_value=4.0, filter_importance_sampling=False, depth_of_field=True,
                        light_intensity_multiplier=1.5, next_event_estimation=True, clipping_planes=True, medium_stack_size=2, frame=17.0, max_volume_walk_length=5,
                        jittered_sampling=False, meters_per_scene_unit=0.01, progressive_accumulation=False, dome_light_camera_visible=False,
                        clear_color=(0.25, 0.5, 0.75, 1.0))
    defaults = RenderSettings()
    assert all(getattr(rs, f.name) != getattr(defaults, f.name) for f in dataclasses.fields(rs)), "a new settings field: give it a non-default value here"
    save_scene(tmp_path / "s.gscn", desc, rs, 96, 54)
    got, got_rs, w, h = load_scene(tmp_path / "s.gscn")
    assert (w, h) == (96, 54)
    _same(desc, got)
    _same(rs, got_rs)
    save_scene(tmp_path / "bare.gscn", desc)  # settings are optional
    got2, none_rs, _, _ = load_scene(tmp_path / "bare.gscn")
    assert none_rs is None
    _same(desc, got2)


def test_c_harness_compiles_and_parses(tmp_path):
    """Strict C99 (-pedantic -Werror) against the C ABI header; --info parses the whole file without a device."""
    exe = _build_harness()
    desc = textured_scene(dome=True)
    save_scene(tmp_path / "s.gscn", desc, RenderSettings(spp=4, max_bounces=6), 96, 54)
    out = subprocess.run([exe, str(tmp_path / "s.gscn"), "--info"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    tris = sum(len(m.faces) for m in desc.meshes)
    assert f"{len(desc.textures)} textures, {len(desc.materials)} materials, {len(desc.meshes)} meshes, {tris} triangles" in out.stdout
    assert "dome 1, settings 1 (96x54 spp 4 bounces 6)" in out.stdout
    blob = open(tmp_path / "s.gscn", "rb").read()
    rng = np.random.default_rng(1)
    for it in range(60):  # corrupt counts / fields: parsed or refused (exit 0 or 1), never a crash
        b = bytearray(blob)
        i = int(rng.integers(0, min(len(b) - 4, 3000))) & ~3
        b[i:i + 4] = bytes([255, 255, 255, 127]) if it % 2 else bytes(rng.integers(0, 256, 4).astype(np.uint8))
        open(tmp_path / "fuzz.gscn", "wb").write(bytes(b))
        res = subprocess.run([exe, str(tmp_path / "fuzz.gscn"), "--info"], capture_output=True, text=True)
        assert res.returncode in (0, 1), res.returncode
    for cut in (3, 40, len(blob) // 2, len(blob) - 1):  # truncated files are refused, never read out of bounds
        open(tmp_path / "cut.gscn", "wb").write(blob[:cut])
        bad = subprocess.run([exe, str(tmp_path / "cut.gscn"), "--info"], capture_output=True, text=True)
        assert bad.returncode == 1 and "gi_render:" in bad.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "textured+dome", "volume", "interior", "leaf cards"])
def test_c_harness_renders_like_the_binding(gi, tmp_path, name):
    """The same scene through tools/gi_render (plain C over the C ABI) and through gatling_amd.capi (ctypes): identical images,
    whole frame and an interleaved row share; command-line overrides act like the settings they name."""
    exe = _build_harness()
    desc = SCENES[name]()
    rs = RenderSettings(spp=3, max_bounces=5, next_event_estimation=bool(desc.rect_lights), medium_stack_size=2 if name == "volume" else 0,
                        clear_color=(0.25, 0.5, 0.75, 1.0), progressive_accumulation=False)
    w, h = 96, 54
    save_scene(tmp_path / "s.gscn", desc, rs, w, h)
    sc = gi.Scene(desc)
    try:
        ref = sc.render(rs, w, h)
        rs2 = dataclasses.replace(rs, spp=2, max_bounces=3)
        ref2 = sc.render(rs2, w, h, rows=(1, h), row_stride=4)
    finally:
        sc.close()
    out = subprocess.run([exe, str(tmp_path / "s.gscn"), str(tmp_path / "o.raw"), "--stats"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Msamples/s" in out.stdout
    img = np.fromfile(tmp_path / "o.raw", np.float32).reshape(h, w, 4)
    assert np.array_equal(img, ref)
    out = subprocess.run([exe, str(tmp_path / "s.gscn"), str(tmp_path / "p.pfm"), "--spp", "2", "--max-bounces", "3", "--rows", f"1:{h}:4"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    with open(tmp_path / "p.pfm", "rb") as f:
        assert f.readline() == b"PF\n" and f.readline() == b"%d %d\n" % (w, h) and f.readline() == b"-1.0\n"
        pfm = np.frombuffer(f.read(), np.float32).reshape(h, w, 3)
    assert np.array_equal(pfm[1:h:4], ref2[..., :3])
