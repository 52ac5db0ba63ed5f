"""Scene builders for the BASELINE.json configs (SURVEY.md section 8d "Concrete inputs").

``cornell_box()`` restates the *facts* of ``/root/reference/cornell.usda`` (file lines
cited below) programmatically so that tests/bench never read ``/root/reference`` at run time (it does not exist
on the GPU box).  ``tests/test_scene_cornell.py`` checks it field-by-field against the .usda through
:mod:`gatling_amd.usda` whenever the reference checkout is present.
"""
from __future__ import annotations

import numpy as np

from .meshprep import build_mesh_arrays
from .scene import (CameraDesc, DomeLight, MaterialDesc, MeshDesc, RectLight, SceneDesc, TextureBinding, MAT_DIFFUSE, MAT_OPEN_PBR,
                    MAT_USD_PREVIEW_SURFACE)
from .usda import camera_from_prim, Prim

_BOX_COUNTS = [4, 4, 4, 4, 4, 4]
_BOX_INDICES = [0, 1, 3, 2, 2, 3, 7, 6, 6, 7, 5, 4, 4, 5, 1, 0, 2, 6, 4, 0, 7, 3, 1, 5]
_BOX_NORMALS = ([(-1, -0.0, 0)] * 4 + [(0, 1, 0)] * 4 + [(1, -0.0, 0)] * 4 + [(0, -1, 0)] * 4 + [(0, 0, -1)] * 4
                + [(0, -0.0, 1)] * 4)
_UNIT_BOX = [(-1, -1, -1), (-1, -1, 1), (-1, 1, -1), (-1, 1, 1), (1, -1, -1), (1, -1, 1), (1, 1, -1), (1, 1, 1)]


def _quad_mesh(name, mid, points, normal, material):
    v, f = build_mesh_arrays(points, [4], [0, 1, 3, 2], normals=[normal] * 4, normals_interpolation="faceVarying")
    return MeshDesc(name=name, vertices=v, faces=f, material=material, id=mid, double_sided=True)


def _box_mesh(name, mid, points, material, transform=None):
    v, f = build_mesh_arrays(points, _BOX_COUNTS, _BOX_INDICES, normals=_BOX_NORMALS, normals_interpolation="faceVarying")
    m = MeshDesc(name=name, vertices=v, faces=f, material=material, id=mid, double_sided=True)
    if transform is not None:
        m.transform = np.asarray(transform, np.float64).astype(np.float32)
    return m


def cornell_box(material_class: int = MAT_USD_PREVIEW_SURFACE) -> SceneDesc:
    """The reference's example scene (cornell.usda): 8 meshes / 46 triangles, 4 UsdPreviewSurface materials,
    emissive box light (8.5, 6, 4), camera at (0,-7,0).  ``material_class=MAT_DIFFUSE`` gives config C1's
    "diffuse only" model on the same geometry."""
    s = SceneDesc()
    mk = lambda n, **kw: MaterialDesc.usd_preview_surface(name=n, klass=material_class, **kw)
    s.materials = [
        mk("/Root/Materials/Light", diffuseColor=(0.8, 0.8, 0.8), emissiveColor=(8.5, 6, 4)),  # cornell.usda:51-62
        mk("/Root/Materials/White", diffuseColor=(0.8, 0.8, 0.8)),                              # :64-74
        mk("/Root/Materials/Red", diffuseColor=(1, 0, 0)),                                      # :76-86
        mk("/Root/Materials/Green", diffuseColor=(0, 1, 0)),                                    # :88-98
    ]
    light_pts = [(-0.5, -0.5, 0.98), (-0.5, -0.5, 1), (-0.5, 0.5, 0.98), (-0.5, 0.5, 1),
                 (0.5, -0.5, 0.98), (0.5, -0.5, 1), (0.5, 0.5, 0.98), (0.5, 0.5, 1)]           # :44
    s.meshes = [
        _box_mesh("/Root/Light/Light", 0, light_pts, 0),
        _quad_mesh("/Root/BottomPlane/BottomPlane", 1, [(-1, -1, -1), (1, -1, -1), (-1, 1, -1), (1, 1, -1)], (0, 0, 1), 1),
        _quad_mesh("/Root/TopPlane/TopPlane", 2, [(1, -1, 0.99999994), (-1, -1, 1.0000001), (1, 1, 0.99999994), (-1, 1, 1.0000001)],
                   (-8.940697e-8, 0, -1), 1),
        _quad_mesh("/Root/BackPlane/BackPlane", 3, [(-1, 1, -1), (1, 1, -1), (-1, 0.99999994, 1), (1, 0.99999994, 1)],
                   (0, -1, -2.9802322e-8), 1),
        _quad_mesh("/Root/LeftPlane/LeftPlane", 4, [(-0.99999994, -1, 1), (-1, -1, -1), (-0.99999994, 1, 1), (-1, 1, -1)],
                   (1, 0, -2.9802322e-8), 2),
        _quad_mesh("/Root/RightPlane/RightPlane", 5, [(1, -1, -1), (1, -1, 1), (1, 1, -1), (1, 1, 1)], (-1, -0.0, 0), 3),
        _box_mesh("/Root/Box1/Box1", 6, _UNIT_BOX, 1,
                  [(0.28844988346099854, 0.19823384284973145, 0, 0), (-0.19823384284973145, 0.28844988346099854, 0, 0),
                   (0, 0, 0.699999988079071, 0), (-0.3499999940395355, 0.3499999940395355, -0.30000001192092896, 1)]),  # :210
        _box_mesh("/Root/Box2/Box2", 7, _UNIT_BOX, 1,
                  [(0.2831651270389557, -0.09908341616392136, 0, 0), (0.09908341616392136, 0.2831651270389557, 0, 0),
                   (0, 0, 0.3501630127429962, 0), (0.45550230145454407, -0.41113391518592834, -0.6499999761581421, 1)]),  # :229
    ]
    cam_xf = np.array([(1, 0, 0, 0), (0, -4.371138828673793e-8, 1, 0), (0, -1, -4.371138828673793e-8, 0), (0, -7, 0, 1)], np.float64)  # :15
    cam = Prim("Camera", "Camera", "/Root/Camera/Camera")
    cam.attrs = {"clippingRange": [0.1, 100.0], "focalLength": 50.0, "verticalAperture": 20.25}  # :18-24
    s.camera = camera_from_prim(cam, cam_xf)
    return s


def _look_at_camera(position, target, up, vfov_deg) -> CameraDesc:
    p, t, u = (np.asarray(x, np.float64) for x in (position, target, up))
    f = (t - p) / np.linalg.norm(t - p)
    r = np.cross(f, u); r /= np.linalg.norm(r)
    u2 = np.cross(r, f)
    return CameraDesc(position=tuple(np.float32(p)), forward=tuple(np.float32(f)), up=tuple(np.float32(u2)),
                      vfov=float(np.float32(np.deg2rad(vfov_deg))))


def random_triangle_soup(count: int = 1_000_000, seed: int = 1234, material_class: int = MAT_OPEN_PBR) -> SceneDesc:
    """Config C3: `count` independent triangles in one mesh, centres uniform in [-1,1]^3, edge
    vectors ~ N(0, 0.01^2) per component, one OpenPBR material with the defaults of open_pbr_surface.mtlx:11-92 (base colour .8,
    specular roughness .3, ior 1.5; `material_class` selects the UsdPreviewSurface / diffuse stand-ins), one 1x1 rect light at
    z=+1.5 facing -z with intensity 20, NEE on, camera at (0,-4,0) looking +y, vfov 40 degrees."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1.0, 1.0, (count, 1, 3))
    p = (c + rng.normal(0.0, 0.01, (count, 3, 3))).astype(np.float32).reshape(-1, 3)
    n = np.cross(p[1::3] - p[0::3], p[2::3] - p[0::3])
    ln = np.linalg.norm(n, axis=1, keepdims=True); ln[ln == 0] = 1.0
    n = np.repeat((n / ln).astype(np.float32), 3, axis=0)
    from .meshprep import bake_vertices
    verts = bake_vertices(p, n)
    faces = np.arange(3 * count, dtype=np.uint32).reshape(-1, 3)
    s = SceneDesc()
    if material_class == MAT_OPEN_PBR:
        s.materials = [MaterialDesc.open_pbr(name="soup")]
    else:
        s.materials = [MaterialDesc.usd_preview_surface(name="soup", diffuseColor=(0.8, 0.8, 0.8), roughness=0.3, ior=1.5, klass=material_class)]
    s.meshes = [MeshDesc(name="/Soup", vertices=verts, faces=faces, material=0, id=0, double_sided=True)]
    s.rect_lights = [RectLight(origin=(0.0, 0.0, 1.5), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(20, 20, 20), width=1.0, height=1.0)]
    s.camera = _look_at_camera((0, -4, 0), (0, 0, 0), (0, 0, 1), 40.0)
    return s


def icosphere(subdivisions: int = 3):
    """Unit icosphere: 20 * 4^subdivisions triangles (subdivisions=4 -> 5120, the C4 prototype)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [np.asarray(x, np.float64) / np.linalg.norm(x) for x in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdivisions):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    pts = np.asarray(v, np.float32)
    return pts, np.asarray(f, np.uint32)


def _parameter_sets(rng, count, material_class=None):
    """The material parameter sets of configs C4/C5: base colour ~U[0,1]^3, roughness ~U[.05,1],
    metalness in {0,1} p=.25, coat weight in {0,1} p=.25, transmission weight in {0,1} p=.125; sets with transmission are
    OpenPBR, the others alternate between OpenPBR and UsdPreviewSurface (or all use `material_class` when given)."""
    mats = []
    for i in range(count):
        color, rough = tuple(rng.uniform(0, 1, 3)), float(rng.uniform(0.05, 1.0))
        metal, coat, trans = float(rng.uniform() < 0.25), float(rng.uniform() < 0.25), float(rng.uniform() < 0.125)
        klass = material_class if material_class is not None else (MAT_OPEN_PBR if (trans > 0 or i % 2 == 0) else MAT_USD_PREVIEW_SURFACE)
        if klass == MAT_OPEN_PBR:
            mats.append(MaterialDesc.open_pbr(name=f"mat{i}", base_color=color, specular_roughness=rough, base_metalness=metal, coat_weight=coat,
                                              coat_roughness=0.05, transmission_weight=trans if material_class is None else 0.0))
        else:
            mats.append(MaterialDesc.usd_preview_surface(name=f"mat{i}", diffuseColor=color, roughness=rough, metallic=metal, clearcoat=coat,
                                                         clearcoatRoughness=0.05, klass=klass))
    return mats


def sphere_grid(grid: int = 32, subdivisions: int = 4, material_count: int = 32, seed: int = 4321, material_class=None) -> SceneDesc:
    """Config C4: grid x grid instances of one icosphere prototype, bound round-robin to
    `material_count` OpenPBR / UsdPreviewSurface parameter sets (_parameter_sets), constant (1,1,1) environment through the
    colour clear value, no analytic lights, NEE off."""
    from .meshprep import bake_vertices
    rng = np.random.default_rng(seed)
    pts, faces = icosphere(subdivisions)
    verts = bake_vertices(pts, pts / np.linalg.norm(pts, axis=1, keepdims=True))
    s = SceneDesc()
    s.materials = _parameter_sets(rng, material_count, material_class)
    per_mat = [[] for _ in range(material_count)]
    k = 0
    for gy in range(grid):
        for gx in range(grid):
            m = np.eye(4, dtype=np.float32)
            m[0, 0] = m[1, 1] = m[2, 2] = 0.45
            m[3, 0] = gx - (grid - 1) / 2.0
            m[3, 2] = gy - (grid - 1) / 2.0
            per_mat[k % material_count].append(m)
            k += 1
    for i, xf in enumerate(per_mat):
        if xf:
            s.meshes.append(MeshDesc(name=f"/Spheres/m{i}", vertices=verts, faces=faces, material=i, id=i,
                                     instance_transforms=np.stack(xf), instance_ids=np.arange(len(xf), dtype=np.int32)))
    dist = grid * 1.6
    s.camera = _look_at_camera((0, -dist, 0), (0, 0, 0), (0, 0, 1), 40.0)
    return s


def interior_scene(clutter_instances: int = 2000, subdivisions: int = 4, prototypes: int = 20, material_count: int = 50, seed: int = 9876) -> SceneDesc:
    """Config C5: a 10 x 8 x 3 m room (12 triangles) filled with `clutter_instances` randomly
    placed, scaled and rotated instances of `prototypes` lumpy icosphere meshes (20 * 4^subdivisions triangles each:
    2000 x 5120 = 10.24 M triangles at the default size), `material_count` parameter sets as in C4, four 1 x 2 m rect
    lights under the ceiling (intensity 30), NEE on.  Smaller arguments give the same structure at test size."""
    from .meshprep import bake_vertices
    rng = np.random.default_rng(seed)
    s = SceneDesc()
    s.materials = [MaterialDesc.usd_preview_surface(name="walls", diffuseColor=(0.7, 0.7, 0.7), roughness=0.8)] + _parameter_sets(rng, material_count)
    # the room: inward-facing box, x in [-5,5], y in [-4,4], z in [0,3]
    lo, hi = np.array([-5.0, -4.0, 0.0]), np.array([5.0, 4.0, 3.0])
    corners = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                        [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]], np.float32)
    quads = [((0, 1, 2, 3), (0, 0, 1)), ((7, 6, 5, 4), (0, 0, -1)), ((4, 5, 1, 0), (0, 1, 0)), ((6, 7, 3, 2), (0, -1, 0)),
             ((7, 4, 0, 3), (1, 0, 0)), ((5, 6, 2, 1), (-1, 0, 0))]
    rp, rn = [], []
    for (a, b, c, d), nrm in quads:
        for tri in ((a, b, c), (a, c, d)):
            rp += [corners[i] for i in tri]
            rn += [nrm] * 3
    s.meshes.append(MeshDesc(name="/Room", vertices=bake_vertices(np.asarray(rp, np.float32), np.asarray(rn, np.float32)),
                             faces=np.arange(36, dtype=np.uint32).reshape(-1, 3), material=0, id=0, double_sided=True))
    # clutter prototypes: icospheres with smooth radial lumps
    base, faces = icosphere(subdivisions)
    protos = []
    for _ in range(prototypes):
        k = rng.normal(0.0, 1.0, (3, 3))
        bump = 1.0 + 0.15 * np.sin(base @ k[0] * 3.0) + 0.1 * np.sin(base @ k[1] * 5.0 + 1.0) + 0.05 * np.sin(base @ k[2] * 9.0)
        pts = (base * bump[:, None]).astype(np.float32)
        fn = np.cross(pts[faces[:, 1]] - pts[faces[:, 0]], pts[faces[:, 2]] - pts[faces[:, 0]])
        vn = np.zeros_like(pts, dtype=np.float64)
        for c in range(3):
            np.add.at(vn, faces[:, c], fn)
        vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-20)
        protos.append(bake_vertices(pts, vn.astype(np.float32)))
    groups = {}
    for i in range(clutter_instances):
        proto, mat = int(rng.integers(prototypes)), 1 + int(rng.integers(material_count))
        scale = rng.uniform(0.08, 0.35) * rng.uniform(0.6, 1.4, 3)
        axis = rng.normal(0, 1, 3); axis /= np.linalg.norm(axis)
        ang = rng.uniform(0, 2 * np.pi)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = (np.diag(scale) @ R.T).astype(np.float32)  # row-vector convention: p_world = p_local . M
        m[3, :3] = (rng.uniform(-4.4, 4.4), rng.uniform(-3.4, 3.4), rng.uniform(0.3, 2.4))
        groups.setdefault((proto, mat), []).append(m)
    for n, ((proto, mat), xf) in enumerate(sorted(groups.items())):
        s.meshes.append(MeshDesc(name=f"/Clutter/p{proto}_m{mat}", vertices=protos[proto], faces=faces, material=mat, id=1 + n,
                                 instance_transforms=np.stack(xf), instance_ids=np.arange(len(xf), dtype=np.int32)))
    for lx, ly in ((-2.5, -2.0), (2.5, -2.0), (-2.5, 2.0), (2.5, 2.0)):
        s.rect_lights.append(RectLight(origin=(lx, ly, 2.95), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(30, 30, 30), width=1.0, height=2.0))  # normal = t1 x t0 = -z
    s.camera = _look_at_camera((-4.5, -3.5, 1.6), (1.5, 1.0, 1.0), (0, 0, 1), 60.0)
    return s


def textured_scene(seed: int = 7, dome: bool = True, klass_sphere: int = MAT_OPEN_PBR) -> SceneDesc:
    """Texture-path test scene: a ground quad tiled 3 x 3 in uv (exercises the wrap modes) with base-colour, roughness and
    normal maps; a uv-mapped sphere with base-colour, metallic and emission maps; a rect light; optionally an equirectangular
    dome light.  Textures are small procedural float images."""
    from .meshprep import bake_vertices
    from .scene import (TEX_BASE_COLOR, TEX_EMISSION, TEX_METALLIC, TEX_NORMAL, TEX_ROUGHNESS, TEX_WRAP_CLAMP, TEX_WRAP_CLIP,
                        TEX_WRAP_MIRRORED_REPEAT, TEX_WRAP_REPEAT)
    rng = np.random.default_rng(seed)
    s = SceneDesc()

    def tex(h, w, fn):
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.zeros((h, w, 4), np.float32)
        a[..., :3] = fn((xx + 0.5) / w, (yy + 0.5) / h)
        a[..., 3] = 1.0
        s.textures.append(a)
        return len(s.textures) - 1

    checker = tex(8, 8, lambda u, v: (0.15 + 0.7 * (((np.floor(u * 4) + np.floor(v * 4)) % 2)[..., None])) * np.array([1.0, 0.9, 0.8]))
    noise = tex(16, 16, lambda u, v: rng.uniform(0.05, 0.95, u.shape + (3,)))
    ramp = tex(4, 32, lambda u, v: np.stack([u, v, 1.0 - u], axis=-1))
    bumps = tex(16, 16, lambda u, v: np.stack([0.5 + 0.25 * np.sin(u * 2 * np.pi * 2), 0.5 + 0.25 * np.cos(v * 2 * np.pi * 3), np.full_like(u, 0.9)], axis=-1))
    glow = tex(8, 16, lambda u, v: np.stack([(u > 0.8) * 2.0, (v > 0.7) * 1.0, np.zeros_like(u)], axis=-1))

    ground = MaterialDesc.usd_preview_surface(name="ground", diffuseColor=(0.5, 0.5, 0.5), roughness=0.4)
    ground.textures = {TEX_BASE_COLOR: TextureBinding(texture=checker, wrap_s=TEX_WRAP_REPEAT, wrap_t=TEX_WRAP_MIRRORED_REPEAT),
                       TEX_ROUGHNESS: TextureBinding(texture=noise, wrap_s=TEX_WRAP_CLAMP, wrap_t=TEX_WRAP_CLIP, channel=1, scale=(0.8,) * 4, bias=(0.1,) * 4),
                       TEX_NORMAL: TextureBinding(texture=bumps, scale=(2.0, 2.0, 2.0, 1.0), bias=(-1.0, -1.0, -1.0, 0.0))}
    if klass_sphere == MAT_OPEN_PBR:
        ball = MaterialDesc.open_pbr(name="ball", base_color=(0.8, 0.8, 0.8), specular_roughness=0.25)
    else:
        ball = MaterialDesc.usd_preview_surface(name="ball", diffuseColor=(0.8, 0.8, 0.8), roughness=0.25)
    ball.textures = {TEX_BASE_COLOR: TextureBinding(texture=ramp), TEX_METALLIC: TextureBinding(texture=noise, channel=2),
                     TEX_EMISSION: TextureBinding(texture=glow, wrap_s=TEX_WRAP_CLAMP, wrap_t=TEX_WRAP_CLAMP)}
    s.materials = [ground, ball]
    gp = np.array([[-3, -3, 0], [3, -3, 0], [3, 3, 0], [-3, -3, 0], [3, 3, 0], [-3, 3, 0]], np.float32)
    guv = np.array([[-1, -1], [2, -1], [2, 2], [-1, -1], [2, 2], [-1, 2]], np.float32)  # uv range [-1, 2]: outside [0,1] on purpose
    s.meshes.append(MeshDesc(name="/Ground", vertices=bake_vertices(gp, np.tile([0, 0, 1], (6, 1)), guv), faces=np.arange(6, dtype=np.uint32).reshape(-1, 3),
                             material=0, id=0, double_sided=True))
    pts, faces = icosphere(2)
    suv = np.stack([np.arctan2(pts[:, 1], pts[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(pts[:, 2], -1, 1)) / np.pi], axis=1).astype(np.float32)
    m = np.eye(4, dtype=np.float32); m[0, 0] = m[1, 1] = m[2, 2] = 0.9; m[3, 2] = 0.9
    s.meshes.append(MeshDesc(name="/Ball", vertices=bake_vertices(pts, pts, suv), faces=faces, material=1, id=1, transform=m))
    s.rect_lights = [RectLight(origin=(0.0, 0.0, 4.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(6, 6, 6), width=2.0, height=2.0)]
    if dome:
        env = tex(16, 32, lambda u, v: np.stack([0.2 + 1.5 * v, 0.3 + 0.5 * np.sin(u * 2 * np.pi) ** 2, 0.4 + 0.6 * u], axis=-1))
        q = np.array([0.1, 0.3, -0.2, 0.9], np.float64); q /= np.linalg.norm(q)
        s.dome_light = DomeLight(texture=env, rotation=tuple(np.float32(q)), base_emission=(0.9, 1.0, 1.1))
    s.camera = _look_at_camera((0, -5.5, 2.2), (0, 0, 0.7), (0, 0, 1), 45.0)
    return s


def volume_scene(scatter=(0.6, 0.3, 0.1), anisotropy: float = 0.3, nested: bool = True) -> SceneDesc:
    """Participating-media test scene: a closed icosphere of scattering, absorbing glass (OpenPBR transmission with
    transmission_depth / transmission_scatter / anisotropy) on a diffuse floor, optionally with a smaller clear-glass
    sphere nested inside it (exercises the medium stack), lit by a rect light and the fallback dome."""
    from .meshprep import bake_vertices
    s = SceneDesc()
    s.materials = [MaterialDesc.usd_preview_surface(name="floor", diffuseColor=(0.6, 0.6, 0.6), roughness=0.7),
                   MaterialDesc.open_pbr(name="murky", base_color=(0.9, 0.9, 0.9), specular_roughness=0.1, specular_ior=1.33, transmission_weight=1.0,
                                         transmission_color=(0.8, 0.5, 0.3), transmission_depth=0.6, transmission_scatter=scatter,
                                         transmission_scatter_anisotropy=anisotropy),
                   MaterialDesc.open_pbr(name="clear", specular_roughness=0.05, specular_ior=1.6, transmission_weight=1.0,
                                         transmission_color=(0.6, 0.9, 0.7), transmission_depth=0.3)]
    fp = np.array([[-4, -4, 0], [4, -4, 0], [4, 4, 0], [-4, -4, 0], [4, 4, 0], [-4, 4, 0]], np.float32)
    s.meshes.append(MeshDesc(name="/Floor", vertices=bake_vertices(fp, np.tile([0, 0, 1], (6, 1))), faces=np.arange(6, dtype=np.uint32).reshape(-1, 3),
                             material=0, id=0, double_sided=True))
    pts, faces = icosphere(2)
    m = np.eye(4, dtype=np.float32); m[3, 2] = 1.05
    s.meshes.append(MeshDesc(name="/Murky", vertices=bake_vertices(pts, pts), faces=faces, material=1, id=1, transform=m))
    if nested:
        m2 = np.eye(4, dtype=np.float32); m2[0, 0] = m2[1, 1] = m2[2, 2] = 0.45; m2[3, :3] = (0.2, 0.0, 1.1)
        s.meshes.append(MeshDesc(name="/Clear", vertices=bake_vertices(pts, pts), faces=faces, material=2, id=2, transform=m2))
    s.rect_lights = [RectLight(origin=(0.5, -0.5, 4.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(12, 12, 12), width=1.5, height=1.5)]
    s.camera = _look_at_camera((0, -4.5, 1.6), (0, 0, 1.0), (0, 0, 1), 40.0)
    return s


def leaf_card_scene(seed: int = 5, cards: int = 12) -> SceneDesc:
    """Textured cutouts: a ground quad, a rect light and `cards` double-sided quads ("leaf cards") whose opacity comes from a
    texture -- a leaf-shaped binary mask through UsdPreviewSurface's opacityThreshold, a soft-edged mask used stochastically, and
    an OpenPBR geometry_opacity read from another channel -- so closest-hit AND shadow rays evaluate the any-hit test at each
    candidate's st (rp_main.ahit:51-60)."""
    from .meshprep import bake_vertices
    from .scene import TEX_BASE_COLOR, TEX_OPACITY, TEX_WRAP_CLAMP, TEX_WRAP_REPEAT
    rng = np.random.default_rng(seed)
    s = SceneDesc()
    yy, xx = np.mgrid[0:32, 0:32]
    u, v = (xx + 0.5) / 32, (yy + 0.5) / 32
    leaf = np.zeros((32, 32, 4), np.float32)
    inside = ((u - 0.5) / 0.42) ** 2 + ((v - 0.5) / 0.28) ** 2 < 1.0
    leaf[..., 0] = 0.15 + 0.2 * v; leaf[..., 1] = 0.45 + 0.4 * u; leaf[..., 2] = 0.1
    leaf[..., 3] = inside.astype(np.float32)                                   # alpha: binary leaf mask
    soft = np.zeros((16, 16, 4), np.float32)
    r = np.sqrt(((np.mgrid[0:16, 0:16][1] + 0.5) / 16 - 0.5) ** 2 + ((np.mgrid[0:16, 0:16][0] + 0.5) / 16 - 0.5) ** 2)
    soft[..., :3] = 0.7; soft[..., 1] = np.clip(1.2 - 2.4 * r, 0.0, 1.0); soft[..., 3] = 1.0   # green channel: radial falloff
    s.textures = [leaf, soft]
    ground = MaterialDesc.usd_preview_surface(name="ground", diffuseColor=(0.6, 0.6, 0.6), roughness=0.6)
    masked = MaterialDesc.usd_preview_surface(name="leafMasked", diffuseColor=(0.2, 0.6, 0.1), roughness=0.5, opacityThreshold=0.5)
    masked.textures = {TEX_BASE_COLOR: TextureBinding(texture=0, wrap_s=TEX_WRAP_CLAMP, wrap_t=TEX_WRAP_CLAMP),
                       TEX_OPACITY: TextureBinding(texture=0, wrap_s=TEX_WRAP_CLAMP, wrap_t=TEX_WRAP_CLAMP, channel=3)}
    veil = MaterialDesc.usd_preview_surface(name="veil", diffuseColor=(0.7, 0.3, 0.2), roughness=0.5)
    veil.textures = {TEX_OPACITY: TextureBinding(texture=1, wrap_s=TEX_WRAP_REPEAT, wrap_t=TEX_WRAP_REPEAT, channel=1, scale=(1.0, 0.9, 1.0, 1.0), bias=(0.0, 0.05, 0.0, 0.0))}
    pbr = MaterialDesc.open_pbr(name="pbrLeaf", base_color=(0.3, 0.5, 0.8), specular_roughness=0.4)
    pbr.textures = {TEX_OPACITY: TextureBinding(texture=0, channel=3)}
    s.materials = [ground, masked, veil, pbr]
    gp = np.array([[-4, -4, 0], [4, -4, 0], [4, 4, 0], [-4, -4, 0], [4, 4, 0], [-4, 4, 0]], np.float32)
    s.meshes.append(MeshDesc(name="/Ground", vertices=bake_vertices(gp, np.tile([0, 0, 1], (6, 1)), np.zeros((6, 2), np.float32)),
                             faces=np.arange(6, dtype=np.uint32).reshape(-1, 3), material=0, id=0, double_sided=True))
    quad = np.array([[-0.5, 0, -0.35], [0.5, 0, -0.35], [0.5, 0, 0.35], [-0.5, 0, -0.35], [0.5, 0, 0.35], [-0.5, 0, 0.35]], np.float32)
    quv = np.array([[0, 0], [1, 0], [1, 1], [0, 0], [1, 1], [0, 1]], np.float32)
    for i in range(cards):
        a, b = rng.uniform(0, 2 * np.pi), rng.uniform(-0.6, 0.6)
        ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
        rz = np.array([[ca, sa, 0], [-sa, ca, 0], [0, 0, 1]]); rx = np.array([[1, 0, 0], [0, cb, sb], [0, -sb, cb]])
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = (rx @ rz * rng.uniform(0.8, 1.6)).astype(np.float32)
        m[3, :3] = (rng.uniform(-1.6, 1.6), rng.uniform(-1.2, 1.2), rng.uniform(0.5, 1.8))
        s.meshes.append(MeshDesc(name=f"/Card{i}", vertices=bake_vertices(quad, np.tile([0, -1, 0], (6, 1)), quv * (2.0 if i % 3 == 1 else 1.0)),
                                 faces=np.arange(6, dtype=np.uint32).reshape(-1, 3), material=1 + i % 3, id=1 + i, double_sided=True, transform=m))
    s.rect_lights = [RectLight(origin=(0.3, -0.2, 4.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(9, 9, 9), width=1.5, height=1.5)]
    s.camera = _look_at_camera((0, -6.0, 2.6), (0, 0, 0.8), (0, 0, 1), 42.0)
    return s
