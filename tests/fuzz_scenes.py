"""Random scenes for the differential test of the HIP path against the oracle (test infrastructure).

`random_case(seed)` draws a whole render -- meshes (planes, boxes, spheres, triangle soups with degenerate members; transformed, mirrored, instanced, some
beyond what the kernels keep in LDS), materials of the three BSDF models with every optional lobe switched on, off or in between, textured inputs with every
wrap mode, all four analytic light types, a dome image, camera (depth of field, clipping planes) and render settings (next event estimation, Russian roulette
offsets, medium stacks, clamping, filter importance sampling) -- from one integer, so a mismatch is reproduced by its seed.

  python tests/fuzz_parity.py 0:500          the campaign (GPU box): every case through the C ABI and through the oracle, images compared bit for bit
  tests/test_gpu_fuzz.py                     a fixed sample of it in the GPU suite
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from gatling_amd.meshprep import bake_vertices, smooth_normals  # noqa: E402
from gatling_amd.scene import (INTERP_CONSTANT, INTERP_INSTANCE, INTERP_UNIFORM, INTERP_VERTEX, PRIMVAR_FLOAT, PRIMVAR_INT, PRIMVAR_INT3, PRIMVAR_VEC2, PRIMVAR_VEC3,  # noqa: E402
                               PRIMVAR_VEC4, Primvar)
from gatling_amd.scene import (MAT_DIFFUSE, MAT_OPEN_PBR, MAT_USD_PREVIEW_SURFACE, TEX_BASE_COLOR, TEX_COAT_NORMAL, TEX_EMISSION, TEX_METALLIC,  # noqa: E402
                               TEX_NORMAL, TEX_OPACITY, TEX_ROUGHNESS, TEX_TRANSMISSION_COLOR, TEX_TRANSMISSION_WEIGHT, CameraDesc, DiskLight, DistantLight,
                               DomeLight, MaterialDesc, MeshDesc, RectLight, RenderSettings, SceneDesc, SphereLight, TextureBinding, usd_transform_2d)
from gatling_amd.scenes import icosphere  # noqa: E402


def _weight(rng):
    """A lobe weight: off, fully on, or in between."""
    c = rng.uniform()
    return 0.0 if c < 0.45 else (1.0 if c < 0.6 else float(rng.uniform(0.02, 0.98)))


def _color(rng, lo=0.0, hi=1.0):
    return tuple(float(x) for x in rng.uniform(lo, hi, 3))


def _rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _transform(rng, spread=2.0, scale=(0.3, 1.5)):
    """USD row-vector 4x4: rotation x (sometimes non-uniform, sometimes mirrored) scale, then a translation."""
    s = np.full(3, rng.uniform(*scale))
    if rng.uniform() < 0.3: s = rng.uniform(*scale, 3)
    if rng.uniform() < 0.12: s[int(rng.integers(3))] *= -1.0
    m = np.eye(4)
    m[:3, :3] = np.diag(s) @ _rotation(rng)
    m[3, :3] = rng.uniform(-spread, spread, 3)
    return m.astype(np.float32)


def _texture(rng):
    h, w = int(rng.integers(1, 17)), int(rng.integers(1, 17))
    a = rng.uniform(0.0, 1.0, (h, w, 4)).astype(np.float32)
    if rng.uniform() < 0.3: a[..., :3] *= np.float32(rng.uniform(1.0, 6.0))     # HDR texels
    if rng.uniform() < 0.4: a[..., 3] = (a[..., 3] > 0.5).astype(np.float32)     # a binary mask in alpha
    return a


def _binding(rng, ntex, vector=False):
    b = TextureBinding(texture=int(rng.integers(ntex)), wrap_s=int(rng.integers(4)), wrap_t=int(rng.integers(4)), channel=int(rng.integers(4)))
    if vector: b.scale, b.bias = (2.0, 2.0, 2.0, 1.0), (-1.0, -1.0, -1.0, 0.0)
    elif rng.uniform() < 0.5:
        b.scale, b.bias = tuple(float(x) for x in rng.uniform(0.2, 1.2, 4)), tuple(float(x) for x in rng.uniform(-0.1, 0.2, 4))
    if rng.uniform() < 0.3:
        b.transform = usd_transform_2d(float(rng.uniform(-180, 180)), tuple(rng.uniform(0.3, 3.0, 2)), tuple(rng.uniform(-1, 1, 2)))
    return b


def _material(rng, i, ntex):
    klass = int(rng.choice([MAT_DIFFUSE, MAT_USD_PREVIEW_SURFACE, MAT_OPEN_PBR], p=[0.12, 0.33, 0.55]))
    emission = _color(rng, 0.0, 4.0) if rng.uniform() < 0.15 else (0.0, 0.0, 0.0)
    opacity = float(rng.uniform(0.05, 0.95)) if rng.uniform() < 0.15 else 1.0
    if klass != MAT_OPEN_PBR:
        m = MaterialDesc.usd_preview_surface(
            name=f"m{i}", diffuseColor=_color(rng), emissiveColor=emission, useSpecularWorkflow=int(rng.uniform() < 0.3), specularColor=_color(rng),
            metallic=_weight(rng), roughness=float(rng.choice([0.0, 1.0, rng.uniform(0.02, 1.0)], p=[0.05, 0.05, 0.9])), clearcoat=_weight(rng),
            clearcoatRoughness=float(rng.uniform(0.0, 1.0)), opacity=opacity, opacityThreshold=float(rng.uniform(0.1, 0.9)) if rng.uniform() < 0.5 else 0.0,
            ior=float(rng.choice([1.0, 1.5, rng.uniform(1.05, 2.5)])), klass=klass)
    else:
        lum = float(rng.uniform(0.5, 5.0)) if emission != (0.0, 0.0, 0.0) else 0.0
        m = MaterialDesc.open_pbr(
            name=f"m{i}", base_weight=float(rng.choice([1.0, rng.uniform(0.0, 1.0)])), base_color=_color(rng), base_metalness=_weight(rng),
            specular_weight=float(rng.choice([1.0, 0.0, rng.uniform(0.0, 1.0)], p=[0.6, 0.1, 0.3])), specular_color=_color(rng, 0.3, 1.0),
            specular_roughness=float(rng.choice([0.0, 1.0, rng.uniform(0.02, 1.0)], p=[0.05, 0.05, 0.9])), specular_ior=float(rng.choice([1.5, rng.uniform(1.02, 2.5)])),
            transmission_weight=_weight(rng) if rng.uniform() < 0.4 else 0.0, transmission_color=_color(rng, 0.2, 1.0),
            transmission_depth=float(rng.uniform(0.1, 2.0)) if rng.uniform() < 0.5 else 0.0, coat_weight=_weight(rng), coat_color=_color(rng, 0.3, 1.0),
            coat_roughness=float(rng.uniform(0.0, 1.0)), coat_ior=float(rng.uniform(1.1, 2.2)), emission_luminance=lum, emission_color=_color(rng),
            base_diffuse_roughness=float(rng.uniform(0.0, 1.0)) if rng.uniform() < 0.4 else 0.0,
            transmission_scatter=_color(rng) if rng.uniform() < 0.3 else (0.0, 0.0, 0.0), transmission_scatter_anisotropy=float(rng.uniform(-0.8, 0.8)),
            coat_darkening=float(rng.choice([1.0, 0.0, rng.uniform(0.0, 1.0)])), fuzz_weight=_weight(rng) if rng.uniform() < 0.5 else 0.0, fuzz_color=_color(rng),
            fuzz_roughness=float(rng.uniform(0.0, 1.0)), geometry_thin_walled=bool(rng.uniform() < 0.2),
            subsurface_weight=_weight(rng) if rng.uniform() < 0.3 else 0.0, subsurface_color=_color(rng),
            subsurface_scatter_anisotropy=float(rng.uniform(-0.8, 0.8)), specular_roughness_anisotropy=float(rng.uniform(0.0, 1.0)) if rng.uniform() < 0.3 else 0.0,
            coat_roughness_anisotropy=float(rng.uniform(0.0, 1.0)) if rng.uniform() < 0.3 else 0.0, thin_film_weight=_weight(rng) if rng.uniform() < 0.3 else 0.0,
            thin_film_thickness=float(rng.uniform(0.05, 1.5)), thin_film_ior=float(rng.uniform(1.1, 2.0)), subsurface_radius=float(rng.uniform(0.05, 2.0)),
            subsurface_radius_scale=_color(rng, 0.1, 1.0), geometry_opacity=opacity,
            coat_rotation=float(rng.choice([0.125, 0.25, rng.uniform(-2.0, 2.0)])) if rng.uniform() < 0.5 else 0.0,   # geometry_coat_tangent as a turn
            specular_rotation=float(rng.choice([0.125, 0.25, rng.uniform(-2.0, 2.0)])) if rng.uniform() < 0.5 else 0.0)   # geometry_tangent likewise
    if ntex and rng.uniform() < 0.4:
        slots = [TEX_BASE_COLOR, TEX_EMISSION, TEX_ROUGHNESS, TEX_METALLIC, TEX_NORMAL, TEX_OPACITY]
        if klass == MAT_OPEN_PBR: slots += [TEX_COAT_NORMAL, TEX_TRANSMISSION_WEIGHT, TEX_TRANSMISSION_COLOR]
        if klass == MAT_DIFFUSE: slots = [TEX_BASE_COLOR, TEX_EMISSION, TEX_OPACITY]
        for slot in slots:
            if rng.uniform() < 0.35:
                m.textures[slot] = _binding(rng, ntex, vector=slot in (TEX_NORMAL, TEX_COAT_NORMAL))
    if rng.uniform() < 0.2:   # UsdPrimvarReader inputs (scene data of the mesh; the two names the reference resolves from the frame's uniforms)
        for slot in [TEX_BASE_COLOR, TEX_EMISSION, TEX_ROUGHNESS, TEX_METALLIC] + ([TEX_TRANSMISSION_WEIGHT, TEX_TRANSMISSION_COLOR] if klass == MAT_OPEN_PBR else []):
            if slot not in m.textures and rng.uniform() < 0.4:
                m.primvar_inputs[slot] = str(rng.choice(["pvA", "pvB", "pvC", "pvI", "CAMERA_POSITION", "FRAME", "missing"]))
    return m


def _primvars(rng, nv, nf, ni):
    """Mesh primvars and instancer primvars under the names the materials may read (some absent, one short of its count)."""
    def data(kind, n):
        if kind == PRIMVAR_FLOAT: return rng.uniform(0.0, 1.0, n)
        if kind == PRIMVAR_VEC2: return rng.uniform(0.0, 1.0, (n, 2))
        if kind == PRIMVAR_VEC3: return rng.uniform(0.0, 1.0, (n, 3))
        if kind == PRIMVAR_VEC4: return rng.uniform(0.0, 1.0, (n, 4))
        if kind == PRIMVAR_INT: return rng.integers(0, 2, n)
        return rng.integers(0, 2, (n, 3))
    mesh, inst = [], []
    for name in ("pvA", "pvB", "pvC", "pvI"):
        if rng.uniform() < 0.3: continue
        kind = int(rng.choice([PRIMVAR_INT, PRIMVAR_INT3])) if name == "pvI" else int(rng.choice([PRIMVAR_FLOAT, PRIMVAR_VEC2, PRIMVAR_VEC3, PRIMVAR_VEC4]))
        interp = int(rng.choice([INTERP_CONSTANT, INTERP_UNIFORM, INTERP_VERTEX, INTERP_INSTANCE]))
        if interp == INTERP_INSTANCE:
            inst.append(Primvar(name, kind, interp, data(kind, ni)))
        else:
            n = {INTERP_CONSTANT: 1, INTERP_UNIFORM: nf, INTERP_VERTEX: nv}[interp]
            if n > 4 and rng.uniform() < 0.1: n -= 3     # fewer values than elements
            mesh.append(Primvar(name, kind, interp, data(kind, n)))
    return mesh, inst


def _sphere_uv(p):
    return np.stack([np.arctan2(p[:, 1], p[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(p[:, 2], -1, 1)) / np.pi], axis=1).astype(np.float32)


def _prototype(rng, big):
    """(vertices, faces) of one mesh in object space."""
    kind = rng.choice(["plane", "box", "sphere", "soup"], p=[0.2, 0.15, 0.35, 0.3])
    if kind == "plane":
        e = float(rng.uniform(1.0, 5.0))
        p = np.array([[-e, -e, 0], [e, -e, 0], [e, e, 0], [-e, e, 0]], np.float32)
        uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32) * np.float32(rng.uniform(0.5, 3.0)) - np.float32(rng.uniform(0.0, 1.0))
        return bake_vertices(p, np.tile([0, 0, 1], (4, 1)), uv), np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    if kind == "box":
        c = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32) * rng.uniform(0.2, 1.0, 3).astype(np.float32)
        quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
        pts, nrm, faces = [], [], []
        for q in quads:
            a, b, cc, d = (c[k] for k in q)
            n = np.cross(b - a, cc - a); n /= max(float(np.linalg.norm(n)), 1e-20)
            k = len(pts); pts += [a, b, cc, d]; nrm += [n] * 4; faces += [(k, k + 1, k + 2), (k, k + 2, k + 3)]
        uv = np.tile(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32), (6, 1))
        return bake_vertices(np.array(pts), np.array(nrm), uv), np.array(faces, np.uint32)
    if kind == "sphere":
        pts, faces = icosphere(int(rng.integers(4, 6)) if big else int(rng.integers(0, 4)))
        return bake_vertices(pts * np.float32(rng.uniform(0.3, 1.2)), pts, _sphere_uv(pts)), faces
    n = int(rng.integers(3000, 20000)) if big else int(rng.integers(1, 200))
    size = float(rng.uniform(0.02, 0.2)) if big else float(rng.uniform(0.1, 1.0))
    c = rng.uniform(-1.5, 1.5, (n, 1, 3))
    p = (c + rng.normal(scale=size, size=(n, 3, 3))).astype(np.float32).reshape(-1, 3)
    if n >= 4 and rng.uniform() < 0.3:   # degenerate members: a point, a needle, a repeated vertex
        p[0:3] = p[0]; p[5] = p[4]; p[8] = (p[6] + p[7]) * np.float32(0.5)
    faces = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    if rng.uniform() < 0.5:
        fn = np.cross(p[1::3] - p[0::3], p[2::3] - p[0::3]); ln = np.linalg.norm(fn, axis=1, keepdims=True); ln[ln == 0] = 1.0
        nrm = np.repeat(fn / ln, 3, axis=0); nrm[np.abs(nrm).sum(axis=1) == 0] = (0, 0, 1)
    else:
        nrm = smooth_normals(p, faces) if rng.uniform() < 0.5 else rng.normal(size=(3 * n, 3))
        ln = np.linalg.norm(nrm, axis=1, keepdims=True); ln[ln == 0] = 1.0; nrm = nrm / ln
        nrm[np.abs(nrm).sum(axis=1) == 0] = (0, 0, 1)
    return bake_vertices(p, nrm, rng.uniform(-1.0, 2.0, (3 * n, 2)).astype(np.float32)), faces


def _frame(rng):
    r = _rotation(rng)
    return tuple(float(x) for x in r[0]), tuple(float(x) for x in r[1])


def random_case(seed: int):
    """(SceneDesc, RenderSettings, width, height, extras) of case `seed`; extras: {"aovs": bool, "second_call": bool}."""
    rng = np.random.default_rng([0x6a71, seed])
    s = SceneDesc()
    s.textures = [_texture(rng) for _ in range(int(rng.integers(0, 4)))]
    s.materials = [_material(rng, i, len(s.textures)) for i in range(int(rng.integers(1, 7)))]
    big = rng.uniform() < 0.3
    n_mesh = int(rng.integers(1, 7))
    for k in range(n_mesh):
        v, f = _prototype(rng, big and k == 0)
        md = MeshDesc(name=f"/fuzz/mesh{k}", vertices=v, faces=f, material=int(rng.integers(len(s.materials))), id=int(rng.integers(0, 1000)),
                      double_sided=bool(rng.uniform() < 0.5), left_handed=bool(rng.uniform() < 0.12), visible=bool(rng.uniform() < 0.94), transform=_transform(rng))
        if rng.uniform() < 0.3:
            ni = int(rng.integers(2, 5)) if rng.uniform() < 0.85 else int(rng.integers(5, 60))
            md.instance_transforms = np.stack([_transform(rng, spread=3.0, scale=(0.5, 1.2)) for _ in range(ni)])
            if rng.uniform() < 0.5: md.instance_ids = rng.integers(0, 100, ni).astype(np.int32)
        if rng.uniform() < 0.3:
            md.max_face_id = int(rng.choice([15, 255, 300, 70000]))   # (the face-id stride is 1, 2 or 4 bytes: Gi.cpp:878-885)
            md.face_ids = rng.integers(0, md.max_face_id + 1, len(f)).astype(np.int32)
        if s.materials[md.material].primvar_inputs or rng.uniform() < 0.1:
            md.primvars, md.instancer_primvars = _primvars(rng, len(v), len(f), len(md.instance_transforms))
        s.meshes.append(md)
    many = rng.uniform() < 0.05   # a light list long enough that the pick of one light matters
    for _ in range(int(rng.integers(0, 3)) if not many else int(rng.integers(10, 40))):
        s.sphere_lights.append(SphereLight(pos=tuple(rng.uniform(-3, 3, 3)), base_emission=_color(rng, 0, 30),
                                           radius=(0.0, 0.0, 0.0) if rng.uniform() < 0.1 else tuple(rng.uniform(0.05, 0.6, 3)),
                                           diffuse=float(rng.choice([1.0, rng.uniform(0, 2)])), specular=float(rng.choice([1.0, rng.uniform(0, 2)]))))
    if rng.uniform() < 0.3:
        s.distant_lights.append(DistantLight(direction=tuple(rng.normal(size=3)), base_emission=_color(rng, 0, 4), angle=float(rng.choice([0.0, rng.uniform(0.01, 0.3)])),
                                             diffuse=float(rng.uniform(0.5, 1.5)), specular=float(rng.uniform(0.5, 1.5))))
    for _ in range(int(rng.integers(0, 3))):
        t0, t1 = _frame(rng)
        s.rect_lights.append(RectLight(origin=tuple(rng.uniform(-3, 3, 3)), t0=t0, t1=t1, base_emission=_color(rng, 0, 20), width=float(rng.uniform(0.1, 2.5)),
                                       height=float(rng.uniform(0.1, 2.5)), diffuse=float(rng.uniform(0.5, 1.5)), specular=float(rng.uniform(0.5, 1.5))))
    for _ in range(int(rng.integers(0, 2))):
        t0, t1 = _frame(rng)
        s.disk_lights.append(DiskLight(origin=tuple(rng.uniform(-3, 3, 3)), t0=t0, t1=t1, base_emission=_color(rng, 0, 20), radius_x=float(rng.uniform(0.1, 1.5)),
                                       radius_y=float(rng.uniform(0.1, 1.5)), diffuse=float(rng.uniform(0.5, 1.5)), specular=float(rng.uniform(0.5, 1.5))))
    if s.textures and rng.uniform() < 0.35:
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        s.dome_light = DomeLight(texture=int(rng.integers(len(s.textures))), rotation=tuple(np.float32(q)), base_emission=_color(rng, 0.2, 1.5),
                                 diffuse=float(rng.uniform(0.5, 1.5)), specular=float(rng.uniform(0.5, 1.5)))
    # camera: somewhere on a shell around the origin, looking near it
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    pos = d * rng.uniform(2.5, 9.0)
    fwd = rng.uniform(-0.7, 0.7, 3) - pos; fwd /= np.linalg.norm(fwd)
    up = np.cross(np.cross(fwd, rng.normal(size=3)), fwd); up /= np.linalg.norm(up)
    dist = float(np.linalg.norm(pos))
    s.camera = CameraDesc(position=tuple(np.float32(pos)), forward=tuple(np.float32(fwd)), up=tuple(np.float32(up)), vfov=float(rng.uniform(0.3, 1.6)),
                          f_stop=float(rng.uniform(0.5, 8.0)), focus_distance=float(rng.uniform(0.5, 1.5)) * dist, focal_length=float(rng.uniform(5.0, 80.0)),
                          clip_start=float(rng.uniform(0.01, 0.6)) * dist, clip_end=float(rng.uniform(0.9, 3.0)) * dist)
    rs = RenderSettings(
        spp=int(rng.integers(1, 7)) if rng.uniform() < 0.9 else int(rng.integers(7, 20)), max_bounces=int(rng.integers(0, 10)) if rng.uniform() < 0.8 else int(rng.integers(10, 16)), rr_bounce_offset=int(rng.integers(0, 6)), rr_inv_min_term_prob=float(rng.uniform(0.5, 1.0)),
        max_sample_value=float(rng.choice([10.0, 1.0e6, 0.5])), filter_importance_sampling=bool(rng.uniform() < 0.7), depth_of_field=bool(rng.uniform() < 0.3),
        light_intensity_multiplier=float(rng.choice([1.0, 0.5, 3.0])), next_event_estimation=bool(rng.uniform() < 0.5), clipping_planes=bool(rng.uniform() < 0.25),
        medium_stack_size=int(rng.choice([0, 0, 0, 1, 2, 4])), frame=0.0, max_volume_walk_length=int(rng.integers(1, 9)), jittered_sampling=bool(rng.uniform() < 0.8),
        meters_per_scene_unit=float(rng.choice([1.0, 0.01, 2.5])), progressive_accumulation=True, dome_light_camera_visible=bool(rng.uniform() < 0.8),
        clear_color=tuple(float(x) for x in rng.uniform(0.0, 1.0, 4)))
    w, h = int(rng.integers(1, 72)), int(rng.integers(1, 44))
    if big: w, h = min(w, 48), min(h, 27)
    rs.frame = float(rng.integers(0, 100))
    # rare corners: no geometry at all, a mesh without faces or without a material, a dome without an image, degenerate settings and cameras
    corner = rng.uniform()
    if corner < 0.01: s.meshes = []
    elif corner < 0.02: s.meshes[0].faces = np.zeros((0, 3), np.uint32); s.meshes[0].face_ids = None
    elif corner < 0.04: s.meshes[int(rng.integers(len(s.meshes)))].material = -1
    elif corner < 0.06 and s.dome_light is None: s.dome_light = DomeLight(texture=-1, base_emission=_color(rng, 0.2, 1.5))
    elif corner < 0.08: rs.rr_inv_min_term_prob = float(rng.choice([0.0, 1.0])); rs.rr_bounce_offset = 0
    elif corner < 0.10: rs.light_intensity_multiplier = 0.0
    elif corner < 0.12: rs.max_sample_value = 0.0
    elif corner < 0.14: s.camera.vfov = float(rng.choice([1.0e-3, 3.1]))
    elif corner < 0.16: s.camera.f_stop = 0.0; rs.depth_of_field = True
    elif corner < 0.18: s.camera.clip_start, s.camera.clip_end = s.camera.clip_end, s.camera.clip_start; rs.clipping_planes = True
    elif corner < 0.20:
        for l in s.sphere_lights + s.rect_lights + s.disk_lights + s.distant_lights: l.diffuse = 0.0; l.specular = float(rng.choice([0.0, 1.0]))
    extras = {"aovs": bool(rng.uniform() < 0.3), "second_call": bool(rng.uniform() < 0.3), "big": bool(big)}
    # equivalent schedules of the library ($GATLING_OPTIONS, gi_options.h): none may change a bit
    opts = []
    if rng.uniform() < 0.5:
        for key, values in (("trace_dyn", [0, 8, 32]), ("trace_dyn_spill8", [1]), ("two_level", [1]), ("work_order", [0]), ("defer_slot", [0]), ("bounds_retire", [0]),
                            ("fused", [0]), ("path_bw", [1]), ("pool_slots", [4096, 65536]), ("bvh_collapse", [0]), ("shadow_order", [0, 1]), ("shade_variants", [0]),
                            ("merge_shade_variants", [0, 1]), ("two_stream", [0]), ("two_stream_delay", [1, 2])):
            if rng.uniform() < 0.2: opts.append(f"{key}={int(rng.choice(values))}")
    extras["options"] = ",".join(opts)
    # a row range / an interleaved row share of the image instead of all of it (multi-GPU partition of one frame)
    extras["rows"] = None
    if h >= 4 and rng.uniform() < 0.2:
        r0 = int(rng.integers(0, h - 1)); extras["rows"] = (r0, int(rng.integers(r0 + 1, h + 1)), int(rng.choice([1, 1, 2, 3])))
    # hostile geometry (include/gi_c.h "Hostile input"): non-finite / out-of-range positions make their triangles inactive, unusable instance transforms their
    # instances; non-finite shading attributes take stand-ins.  The oracle renders tests/test_hostile_inputs.py sanitised() of the same description.
    extras["hostile"] = False
    if rng.uniform() < 0.1:
        extras["hostile"] = True
        for m in s.meshes:
            if rng.uniform() < 0.6:
                v = m.vertices.copy()
                for _ in range(int(rng.integers(1, 4))):
                    field, k = str(rng.choice(["pos", "pos", "norm", "tangent", "u", "bitangentSign"])), int(rng.integers(len(v)))
                    bad = np.float32(rng.choice([np.nan, np.inf, -np.inf, 3.0e38]))
                    if field in ("u", "bitangentSign"): v[field][k] = bad if np.isinf(bad) or np.isnan(bad) else np.float32(np.nan)
                    elif field == "pos": v[field][k, int(rng.integers(3))] = bad
                    else: v[field][k, int(rng.integers(3))] = np.float32(np.nan)
                m.vertices = v
            if rng.uniform() < 0.25:
                it = np.array(m.instance_transforms, np.float32).copy(); k = int(rng.integers(len(it)))
                kind = rng.uniform()
                if kind < 0.4: it[k, int(rng.integers(3)), int(rng.integers(3))] = np.nan
                elif kind < 0.7: it[k, :3, :3] = 0.0                       # singular
                else: it[k, 2, :3] = it[k, 1, :3]                          # two equal axes: singular
                m.instance_transforms = it
    # an edit between two renders: instance transforms, visibility, a material swap
    extras["edit"] = str(rng.choice(["transforms", "visibility", "material", "mesh_transform", "mesh_remove", "light_move", "light_remove", "light_add"])) if rng.uniform() < 0.25 else None
    extras["edit_seed"] = int(rng.integers(1 << 30))
    # numerical extremes (finite, so nothing is refused): the whole scene shrunk or blown up, emission that overflows fp32 along a path, material inputs outside
    # their documented ranges.  Images may hold inf / NaN: they must be the SAME inf / NaN on both sides (the comparison takes any NaN for any NaN)
    extras["extreme"] = None
    if rng.uniform() < 0.12:
        # ("ranges" -- material inputs outside their documented ranges -- is drawn only under $GATLING_FUZZ_RANGES=1: such inputs reach float -> int conversions of
        # non-finite values, whose result the language leaves open and the host and the device fill in differently; 1 of ~1 400 such renders traced other shadow rays
        # than the oracle, with the same image.  The 6 000 out-of-range materials of `--bsdf --ranges` are identical call by call: profiles/r06zu_bsdf_ranges.log)
        kind = str(rng.choice(["scale", "emission", "ranges"])); extras["extreme"] = kind
        if kind == "ranges" and os.environ.get("GATLING_FUZZ_RANGES", "0") != "1": kind = extras["extreme"] = "emission"
        if kind == "scale":
            # (not below 1e-2: in scenes shrunk to 1e-4 units 3 of ~1 400 cases differed from the oracle by one to four path segments -- the Moeller-Trumbore test, the
            # contract's arbiter, accepts hits with determinants around 1e-20 whose position lies outside every bounding box, and which of those a walk meets depends
            # on the tree it walks: DESIGN.md section 8, profiles/r06zs_fuzz_parity_extremes.log)
            f = np.float32(10.0 ** rng.uniform(-2, 5))
            for m in s.meshes: m.transform = m.transform.copy(); m.transform[3, :3] *= f; m.transform[:3, :3] *= f
            for l in s.sphere_lights: l.pos = tuple(np.float32(l.pos) * f); l.radius = tuple(np.float32(l.radius) * f)
            for l in s.rect_lights: l.origin = tuple(np.float32(l.origin) * f); l.width *= float(f); l.height *= float(f)
            for l in s.disk_lights: l.origin = tuple(np.float32(l.origin) * f); l.radius_x *= float(f); l.radius_y *= float(f)
            c = s.camera; c.position = tuple(np.float32(c.position) * f); c.focus_distance *= float(f); c.clip_start *= float(f); c.clip_end *= float(f)
        elif kind == "emission":
            g = float(10.0 ** rng.uniform(10, 37))
            for l in s.sphere_lights + s.rect_lights + s.disk_lights + s.distant_lights: l.base_emission = tuple(float(x) * g for x in l.base_emission)
            for m in s.materials: m.params[3:6] *= np.float32(min(g, 1e30))
            rs.max_sample_value = float(rng.choice([10.0, 1.0e38, float("inf")]))
        else:
            for m in s.materials:
                for _ in range(int(rng.integers(1, 5))):
                    i = int(rng.integers(0, 64))
                    if i in (6, 14, 15, 54): continue   # switches and the cutout pair keep their meaning
                    m.params[i] = np.float32(rng.choice([-1.0, -0.25, 1.5, 7.0, 0.3, 50.0, 1.0e-6, 1.0e6]))
    # (drawn last, so that earlier seeds keep their scenes) materials handed over as MaterialX documents through the gtl shim's reader, the way hdGatling does;
    # a batch of rays through giCTraceRays
    extras["mtlx"] = {}
    if rng.uniform() < 0.25:
        for mi, m in enumerate(s.materials):
            if m.klass != MAT_DIFFUSE and not m.textures and not m.primvar_inputs and rng.uniform() < 0.6:
                extras["mtlx"][mi] = str(rng.choice(["direct", "nodegraph"]))
    # scene options of the C ABI (include/gi_c.h GI_C_SCENE_OPTION_*): the counting instantiations of the kernels, the stage timers, the schedules by option
    extras["scene_options"] = []
    if rng.uniform() < 0.25:
        for opt, values in ((1, [1]), (2, [1, 4]), (5, [0, 16]), (6, [1]), (7, [0, 1, 2])):   # COUNT_TRAVERSAL, KERNEL_TIMERS, TRACE_DYNAMIC, TWO_LEVEL, FUSED_PATH
            if rng.uniform() < 0.35: extras["scene_options"].append((opt, int(rng.choice(values))))
    extras["trace_rays"] = int(rng.integers(1, 3000)) if rng.uniform() < 0.15 else 0
    extras["trace_seed"] = int(rng.integers(1 << 30))
    return s, rs, w, h, extras


def apply_edit(desc, kind, edit_seed):
    """Edits the description the way the case's edit changes the scene (what the library's incremental update must equal) and returns the operation for the driver:
    {"op": ..., "mesh": index} or {"op": ..., "light": (list name, index)}."""
    rng = np.random.default_rng(edit_seed)
    if not desc.meshes:   # (the empty scene: nothing to edit but the lights)
        kind = "light_add"; k = 0; m = None
    else:
        k = int(rng.integers(len(desc.meshes)))
        m = desc.meshes[k]
    lights = [(name, i) for name in ("sphere_lights", "distant_lights", "rect_lights", "disk_lights") for i in range(len(getattr(desc, name)))]
    if kind in ("light_move", "light_remove") and not lights: kind = "light_add"
    if m is None and kind not in ("light_move", "light_remove"): kind = "light_add"
    if kind == "mesh_remove" and len(desc.meshes) == 1: kind = "visibility"
    if kind == "transforms":
        m.instance_transforms = np.stack([_transform(rng, spread=3.0, scale=(0.5, 1.2)) for _ in range(len(m.instance_transforms))])
    elif kind == "visibility":
        m.visible = not m.visible
    elif kind == "material":
        m.material = int(rng.integers(len(desc.materials)))
    elif kind == "mesh_transform":
        m.transform = _transform(rng)
    elif kind == "mesh_remove":
        desc.meshes.pop(k)
    elif kind == "light_add":
        desc.sphere_lights.append(SphereLight(pos=tuple(rng.uniform(-3, 3, 3)), base_emission=_color(rng, 0, 30), radius=tuple(rng.uniform(0.05, 0.6, 3))))
        return {"op": kind, "light": ("sphere_lights", len(desc.sphere_lights) - 1)}
    else:
        name, i = lights[int(rng.integers(len(lights)))]
        if kind == "light_remove":   # the reference's dense store fills the hole with the LAST element (DenseDataStore.cpp:45-68), and so does the library
            lst = getattr(desc, name); lst[i] = lst[-1]; lst.pop()
        else:
            l = getattr(desc, name)[i]
            if name == "sphere_lights": l.pos = tuple(rng.uniform(-3, 3, 3))
            elif name == "distant_lights": l.direction = tuple(rng.normal(size=3))
            else: l.origin = tuple(rng.uniform(-3, 3, 3))
            l.base_emission = _color(rng, 0, 20)
        return {"op": kind, "light": (name, i)}
    return {"op": kind, "mesh": k}
