"""The flat binary scene file (``.gscn``) that the C harness ``tools/gi_render.c`` loads ("the generator
must emit both .usda and the flat binary the C harness loads").

It is the :class:`SceneDesc` the way the gi C ABI wants it -- the inputs of ``giCCreateMesh`` / ``giCCreateMaterial`` / the light
setters (reference interface ``src/gi/gtl/gi/Gi.h:86-175``) -- written sequentially, little-endian, every field 4-byte aligned:

    "GSCN" u32 version
    u32 hasSettings [u32 width, height; GiCRenderSettings (18 x 4 B, include/gi_c.h); f32 clearColor[4]]
    GiCCameraDesc (16 x f32)
    u32 nTextures   { u32 width, height; f32 rgba[h * w * 4] }
    u32 nMaterials  { str name; u32 klass; u32 nParams; f32 params[nParams];
                      9 x { i32 texture (-1: none), wrapS, wrapT, channel; f32 scale[4], bias[4]; u32 hasTransform; f32 xf[6] }; 9 x str primvarName }   (GI_C_TEX_* slots)
    u32 hasDome     [i32 texture; f32 rotation[4], baseEmission[3], diffuse, specular]
    u32 nMeshes     { str name; u32 nVertices, nFaces; i32 id; u32 flags; i32 material; u32 maxFaceId; f32 transform[16];
                      u32 nInstances; f32 instanceTransforms[nInstances * 16]; [i32 instanceIds[nInstances]]
                      GiCVertex vertices[nVertices] (48 B each); u32 faces[nFaces * 3]; [i32 faceIds[nFaces]]
                      u32 nPrimvars { primvar }; u32 nInstancerPrimvars { primvar } }
    u32 nSphere {11 f32}  u32 nDistant {9 f32}  u32 nRect {16 f32}  u32 nDisk {16 f32}
    "END!"

    str     = u32 byteLength; bytes; zero padding to a multiple of 4
    primvar = str name; i32 type, interpolation; u32 nElems; f32 | i32 data[nElems]   (i32 for the types Int .. Int4)
    flags   = 1 doubleSided | 2 leftHanded | 4 visible | 8 hasFaceIds | 16 hasInstanceIds
"""
from __future__ import annotations

import struct

import numpy as np

from .scene import (CameraDesc, DiskLight, DistantLight, DomeLight, MaterialDesc, MeshDesc, P_COUNT, Primvar, RectLight, RenderSettings,
                    SceneDesc, SphereLight, TextureBinding, VERTEX_DTYPE)

MAGIC, END, VERSION = b"GSCN", b"END!", 3  # 2 (round 4): seven texture slots (geometry_coat_normal), a texture-coordinate transform per slot; 3: nine (transmission weight / colour)
TEX_SLOTS = 9
F_DOUBLE_SIDED, F_LEFT_HANDED, F_VISIBLE, F_FACE_IDS, F_INSTANCE_IDS = 1, 2, 4, 8, 16


class _Writer:
    def __init__(self, f):
        self.f = f

    def u32(self, *v):
        self.f.write(struct.pack("<%dI" % len(v), *[int(x) for x in v]))

    def i32(self, *v):
        self.f.write(struct.pack("<%di" % len(v), *[int(x) for x in v]))

    def f32(self, *v):
        self.f.write(np.asarray(v, "<f4").tobytes())

    def arr(self, a, dtype):
        self.f.write(np.ascontiguousarray(a, dtype).tobytes())

    def string(self, s):
        b = s.encode()
        self.u32(len(b))
        self.f.write(b + b"\0" * (-len(b) % 4))


def _settings_words(rs: RenderSettings) -> bytes:
    """GiCRenderSettings, field for field (include/gi_c.h; Gi.h:139-159)."""
    return struct.pack("<4if i f I f 2I f 2i I f I f", int(rs.clipping_planes), int(rs.depth_of_field), int(rs.dome_light_camera_visible),
                       int(rs.filter_importance_sampling), float(rs.frame), int(rs.jittered_sampling), rs.light_intensity_multiplier, rs.max_bounces,
                       rs.max_sample_value, rs.max_volume_walk_length, rs.medium_stack_size, rs.meters_per_scene_unit,
                       int(rs.next_event_estimation), int(rs.progressive_accumulation), rs.rr_bounce_offset, rs.rr_inv_min_term_prob, rs.spp, 0.0)


def save_scene(path, desc: SceneDesc, settings: RenderSettings = None, width: int = 0, height: int = 0):
    with open(path, "wb") as f:
        w = _Writer(f)
        f.write(MAGIC)
        w.u32(VERSION)
        w.u32(1 if settings is not None else 0)
        if settings is not None:
            w.u32(width, height)
            f.write(_settings_words(settings))
            w.f32(*settings.clear_color)
        c = desc.camera
        w.f32(*c.position, *c.forward, *c.up, c.vfov, c.f_stop, c.focus_distance, c.focal_length, c.clip_start, c.clip_end, c.exposure)
        w.u32(len(desc.textures))
        for t in desc.textures:
            a = np.ascontiguousarray(t, "<f4")
            w.u32(a.shape[1], a.shape[0])
            f.write(a.tobytes())
        w.u32(len(desc.materials))
        for m in desc.materials:
            w.string(m.name)
            w.u32(m.klass, P_COUNT)
            w.arr(m.params, "<f4")
            for slot in range(TEX_SLOTS):
                b = m.textures.get(slot)
                if b is None:
                    b = TextureBinding()
                w.i32(b.texture, b.wrap_s, b.wrap_t, b.channel)
                w.f32(*b.scale, *b.bias)
                xf = getattr(b, "transform", None)
                w.u32(0 if xf is None else 1)
                w.f32(*(xf if xf is not None else (1.0, 0.0, 0.0, 0.0, 1.0, 0.0)))
            for slot in range(TEX_SLOTS):
                w.string(m.primvar_inputs.get(slot, ""))
        d = desc.dome_light
        w.u32(1 if d is not None else 0)
        if d is not None:
            w.i32(d.texture)
            w.f32(*d.rotation, *d.base_emission, d.diffuse, d.specular)
        w.u32(len(desc.meshes))
        for m in desc.meshes:
            w.string(m.name)
            flags = (F_DOUBLE_SIDED if m.double_sided else 0) | (F_LEFT_HANDED if m.left_handed else 0) | (F_VISIBLE if m.visible else 0) \
                | (F_FACE_IDS if m.face_ids is not None else 0) | (F_INSTANCE_IDS if m.instance_ids is not None else 0)
            w.u32(len(m.vertices), len(m.faces))
            w.i32(m.id)
            w.u32(flags)
            w.i32(m.material)
            w.u32(m.max_face_id)
            w.arr(np.asarray(m.transform, np.float32).reshape(16), "<f4")
            inst = np.ascontiguousarray(m.instance_transforms, "<f4").reshape(-1, 16)
            w.u32(len(inst))
            f.write(inst.tobytes())
            if m.instance_ids is not None:
                w.arr(m.instance_ids, "<i4")
            f.write(np.ascontiguousarray(m.vertices, VERTEX_DTYPE).tobytes())
            w.arr(np.asarray(m.faces).reshape(-1, 3), "<u4")
            if m.face_ids is not None:
                w.arr(m.face_ids, "<i4")
            for pvs in (m.primvars, m.instancer_primvars):
                w.u32(len(pvs))
                for pv in pvs:
                    data = np.ascontiguousarray(pv.data, "<i4" if int(pv.type) >= 4 else "<f4").reshape(-1)  # Int..Int4: 4-byte elements too
                    w.string(pv.name)
                    w.i32(pv.type, pv.interpolation)
                    w.u32(len(data))
                    f.write(data.tobytes())
        w.u32(len(desc.sphere_lights))
        for l in desc.sphere_lights:
            w.f32(*l.pos, *l.base_emission, *l.radius, l.diffuse, l.specular)
        w.u32(len(desc.distant_lights))
        for l in desc.distant_lights:
            w.f32(*l.direction, *l.base_emission, l.angle, l.diffuse, l.specular)
        w.u32(len(desc.rect_lights))
        for l in desc.rect_lights:
            w.f32(*l.origin, *l.t0, *l.t1, *l.base_emission, l.width, l.height, l.diffuse, l.specular)
        w.u32(len(desc.disk_lights))
        for l in desc.disk_lights:
            w.f32(*l.origin, *l.t0, *l.t1, *l.base_emission, l.radius_x, l.radius_y, l.diffuse, l.specular)
        f.write(END)


class _Reader:
    def __init__(self, data: bytes):
        self.d, self.p = data, 0

    def take(self, n):
        if self.p + n > len(self.d):
            raise ValueError("gscn: truncated file")
        b = self.d[self.p:self.p + n]
        self.p += n
        return b

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def i32(self):
        return struct.unpack("<i", self.take(4))[0]

    def arr(self, n, dtype):
        dt = np.dtype(dtype)
        return np.frombuffer(self.take(n * dt.itemsize), dt).copy()

    def f32(self, n):
        return [float(x) for x in self.arr(n, "<f4")]

    def string(self):
        n = self.u32()
        s = self.take(n).decode()
        self.take(-n % 4)
        return s


def load_scene(path):
    """-> (SceneDesc, RenderSettings or None, width, height)"""
    with open(path, "rb") as f:
        r = _Reader(f.read())
    if r.take(4) != MAGIC:
        raise ValueError("gscn: bad magic")
    if r.u32() != VERSION:
        raise ValueError("gscn: unsupported version")
    settings, width, height = None, 0, 0
    if r.u32():
        width, height = r.u32(), r.u32()
        v = struct.unpack("<4if i f I f 2I f 2i I f I f", r.take(72))
        settings = RenderSettings(clipping_planes=bool(v[0]), depth_of_field=bool(v[1]), dome_light_camera_visible=bool(v[2]),
                                  filter_importance_sampling=bool(v[3]), frame=v[4], jittered_sampling=bool(v[5]), light_intensity_multiplier=v[6],
                                  max_bounces=v[7], max_sample_value=v[8], max_volume_walk_length=v[9], medium_stack_size=v[10],
                                  meters_per_scene_unit=v[11], next_event_estimation=bool(v[12]), progressive_accumulation=bool(v[13]),
                                  rr_bounce_offset=v[14], rr_inv_min_term_prob=v[15], spp=v[16], clear_color=tuple(r.f32(4)))
    desc = SceneDesc()
    c = r.f32(16)
    desc.camera = CameraDesc(tuple(c[0:3]), tuple(c[3:6]), tuple(c[6:9]), *c[9:16])
    for _ in range(r.u32()):
        w, h = r.u32(), r.u32()
        desc.textures.append(r.arr(w * h * 4, "<f4").reshape(h, w, 4))
    for _ in range(r.u32()):
        name = r.string()
        klass, n = r.u32(), r.u32()
        m = MaterialDesc(name=name, klass=klass, params=r.arr(n, "<f4"))
        for slot in range(TEX_SLOTS):
            tex, ws, wt, ch = r.i32(), r.i32(), r.i32(), r.i32()
            sb = r.f32(8)
            has_xf = r.u32()
            xf = r.f32(6)
            if tex >= 0:
                m.textures[slot] = TextureBinding(tex, ws, wt, ch, tuple(sb[:4]), tuple(sb[4:]), tuple(float(x) for x in xf) if has_xf else None)
        for slot in range(TEX_SLOTS):
            s = r.string()
            if s:
                m.primvar_inputs[slot] = s
        desc.materials.append(m)
    if r.u32():
        tex = r.i32()
        v = r.f32(9)
        desc.dome_light = DomeLight(tex, tuple(v[0:4]), tuple(v[4:7]), v[7], v[8])
    for _ in range(r.u32()):
        name = r.string()
        nv, nf = r.u32(), r.u32()
        mid, flags, mat, max_face = r.i32(), r.u32(), r.i32(), r.u32()
        transform = r.arr(16, "<f4").reshape(4, 4)
        ni = r.u32()
        inst = r.arr(ni * 16, "<f4").reshape(ni, 4, 4)
        inst_ids = r.arr(ni, "<i4") if flags & F_INSTANCE_IDS else None
        verts = r.arr(nv, VERTEX_DTYPE)
        faces = r.arr(nf * 3, "<u4").reshape(nf, 3)
        face_ids = r.arr(nf, "<i4") if flags & F_FACE_IDS else None
        pv_lists = []
        for _k in range(2):
            pvs = []
            for _j in range(r.u32()):
                pname = r.string()
                ptype, interp = r.i32(), r.i32()
                comps = (1, 2, 3, 4)[ptype & 3]
                data = r.arr(r.u32(), "<i4" if ptype >= 4 else "<f4")
                pvs.append(Primvar(pname, ptype, interp, data.reshape(-1, comps) if comps > 1 else data))
            pv_lists.append(pvs)
        desc.meshes.append(MeshDesc(name, verts, faces, material=mat, id=mid, double_sided=bool(flags & F_DOUBLE_SIDED),
                                    left_handed=bool(flags & F_LEFT_HANDED), visible=bool(flags & F_VISIBLE), transform=transform,
                                    instance_transforms=inst, instance_ids=inst_ids, face_ids=face_ids, max_face_id=max_face,
                                    primvars=pv_lists[0], instancer_primvars=pv_lists[1]))
    for _ in range(r.u32()):
        v = r.f32(11)
        desc.sphere_lights.append(SphereLight(tuple(v[0:3]), tuple(v[3:6]), tuple(v[6:9]), v[9], v[10]))
    for _ in range(r.u32()):
        v = r.f32(9)
        desc.distant_lights.append(DistantLight(tuple(v[0:3]), tuple(v[3:6]), v[6], v[7], v[8]))
    for _ in range(r.u32()):
        v = r.f32(16)
        desc.rect_lights.append(RectLight(tuple(v[0:3]), tuple(v[3:6]), tuple(v[6:9]), tuple(v[9:12]), v[12], v[13], v[14], v[15]))
    for _ in range(r.u32()):
        v = r.f32(16)
        desc.disk_lights.append(DiskLight(tuple(v[0:3]), tuple(v[3:6]), tuple(v[6:9]), tuple(v[9:12]), v[12], v[13], v[14], v[15]))
    if r.take(4) != END:
        raise ValueError("gscn: missing end marker")
    return desc, settings, width, height
