"""ctypes binding of the CPU oracle (``oracle/libgi_oracle.so``).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module,
and only as the checker.  Nothing under ``gatling_amd/`` imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgi_oracle.so")
P_COUNT = 64


TEX_SLOT_COUNT = 9


class OrcTexBinding(C.Structure):
    _fields_ = [("texture", C.c_int32), ("wrapS", C.c_int32), ("wrapT", C.c_int32), ("channel", C.c_int32),
                ("scale", C.c_float * 4), ("bias", C.c_float * 4), ("hasTransform", C.c_int32), ("xf", C.c_float * 6)]


class OrcMaterial(C.Structure):
    _fields_ = [("klass", C.c_uint32), ("flags", C.c_uint32), ("p", C.c_float * P_COUNT), ("tex", OrcTexBinding * TEX_SLOT_COUNT),
                ("primvarInput", (C.c_char * 64) * TEX_SLOT_COUNT)]


class OrcPrimvar(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("type", C.c_int32), ("interpolation", C.c_int32), ("data", C.c_void_p), ("floatCount", C.c_uint32)]


class OrcTexture(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class OrcDomeLight(C.Structure):
    _fields_ = [("texture", C.c_int32), ("rotation", C.c_float * 4), ("baseEmission", C.c_float * 3)]


def fill_material(dst, m):
    dst.klass = m.klass
    dst.p = (C.c_float * P_COUNT)(*np.asarray(m.params, np.float32))
    for slot in range(TEX_SLOT_COUNT):
        b = getattr(m, "textures", {}).get(slot)
        dst.tex[slot].texture = -1 if b is None else int(b.texture)
        if b is not None:
            dst.tex[slot].wrapS, dst.tex[slot].wrapT, dst.tex[slot].channel = int(b.wrap_s), int(b.wrap_t), int(b.channel)
            dst.tex[slot].scale = (C.c_float * 4)(*b.scale)
            dst.tex[slot].bias = (C.c_float * 4)(*b.bias)
            xf = getattr(b, "transform", None)
            dst.tex[slot].hasTransform = 0 if xf is None else 1
            if xf is not None:
                dst.tex[slot].xf = (C.c_float * 6)(*[float(np.float32(x)) for x in xf])
        dst.primvarInput[slot].value = getattr(m, "primvar_inputs", {}).get(slot, "").encode()


class OrcMesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("vertexCount", C.c_uint32),
                ("faces", C.c_void_p), ("faceCount", C.c_uint32),
                ("id", C.c_int32), ("isDoubleSided", C.c_int32), ("isLeftHanded", C.c_int32), ("visible", C.c_int32),
                ("transform", C.c_float * 16),
                ("instanceTransforms", C.c_void_p), ("instanceCount", C.c_uint32),
                ("material", C.c_int32), ("faceIds", C.c_void_p), ("maxFaceId", C.c_uint32), ("instanceIds", C.c_void_p),
                ("primvars", C.c_void_p), ("primvarCount", C.c_uint32), ("instancerPrimvars", C.c_void_p), ("instancerPrimvarCount", C.c_uint32)]


class OrcAovs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("normal", "barycentrics", "texcoords", "opacity", "tangents", "bitangents", "thinWalled",
                                          "doubleSided", "albedo", "depth", "objectId", "faceId", "instanceId")] + [("clear", (C.c_float * 4) * 17)] + \
               [("nee", C.c_void_p), ("bounces", C.c_void_p), ("clockCycles", C.c_void_p)]


AOV_IDS = {"normal": 1, "barycentrics": 3, "texcoords": 4, "opacity": 7, "tangents": 8, "bitangents": 9, "thinWalled": 10, "objectId": 11,
           "depth": 12, "faceId": 13, "instanceId": 14, "doubleSided": 15, "albedo": 16, "nee": 2, "bounces": 5, "clockCycles": 6}


class OrcSphereLight(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("baseEmission", C.c_float * 3), ("radius", C.c_float * 3),
                ("diffuse", C.c_float), ("specular", C.c_float)]


class OrcDistantLight(C.Structure):
    _fields_ = [("direction", C.c_float * 3), ("baseEmission", C.c_float * 3), ("angle", C.c_float),
                ("diffuse", C.c_float), ("specular", C.c_float)]


class OrcRectLight(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("t0", C.c_float * 3), ("t1", C.c_float * 3), ("baseEmission", C.c_float * 3),
                ("width", C.c_float), ("height", C.c_float), ("diffuse", C.c_float), ("specular", C.c_float)]


class OrcDiskLight(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("t0", C.c_float * 3), ("t1", C.c_float * 3), ("baseEmission", C.c_float * 3),
                ("radiusX", C.c_float), ("radiusY", C.c_float), ("diffuse", C.c_float), ("specular", C.c_float)]


class OrcScene(C.Structure):
    _fields_ = [("meshes", C.c_void_p), ("meshCount", C.c_uint32),
                ("materials", C.c_void_p), ("materialCount", C.c_uint32),
                ("sphereLights", C.c_void_p), ("sphereLightCount", C.c_uint32),
                ("distantLights", C.c_void_p), ("distantLightCount", C.c_uint32),
                ("rectLights", C.c_void_p), ("rectLightCount", C.c_uint32),
                ("diskLights", C.c_void_p), ("diskLightCount", C.c_uint32),
                ("textures", C.c_void_p), ("textureCount", C.c_uint32), ("dome", C.c_void_p)]


class OrcCamera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("forward", C.c_float * 3), ("up", C.c_float * 3),
                ("vfov", C.c_float), ("fStop", C.c_float), ("focusDistance", C.c_float), ("focalLength", C.c_float),
                ("clipStart", C.c_float), ("clipEnd", C.c_float), ("exposure", C.c_float)]


class OrcSettings(C.Structure):
    _fields_ = [("clippingPlanes", C.c_int32), ("depthOfField", C.c_int32), ("domeLightCameraVisible", C.c_int32),
                ("filterImportanceSampling", C.c_int32), ("jitteredSampling", C.c_int32),
                ("nextEventEstimation", C.c_int32), ("progressiveAccumulation", C.c_int32),
                ("maxBounces", C.c_uint32), ("rrBounceOffset", C.c_uint32), ("spp", C.c_uint32), ("sampleOffset", C.c_uint32),
                ("lightIntensityMultiplier", C.c_float), ("maxSampleValue", C.c_float), ("rrInvMinTermProb", C.c_float),
                ("metersPerSceneUnit", C.c_float), ("mediumStackSize", C.c_uint32), ("maxVolumeWalkLength", C.c_uint32),
                ("clearColor", C.c_float * 4), ("frame", C.c_float)]


class OrcRegion(C.Structure):
    _fields_ = [("imageWidth", C.c_uint32), ("imageHeight", C.c_uint32), ("rowBegin", C.c_uint32), ("rowEnd", C.c_uint32)]


class OrcCounters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("shadowRays", C.c_uint64), ("hits", C.c_uint64),
                ("bounceHistogram", C.c_uint64 * 64)]


def build(force: bool = False) -> str:
    """Compiles the oracle with the committed Makefile (g++, -ffp-contract=off)."""
    src = os.path.join(_HERE, "gi_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "gi_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgi_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_rng_init.restype = C.c_uint32
        L.orc_rng_init.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_rng_next1f.restype = C.c_float
        L.orc_rng_next1f.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_hash_pcg32.restype = C.c_uint32
        L.orc_hash_pcg32.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_encode_direction.restype = C.c_uint32
        L.orc_encode_direction.argtypes = [C.POINTER(C.c_float)]
        L.orc_decode_direction.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
        L.orc_offset_ray_origin.argtypes = [C.POINTER(C.c_float)] * 3
        L.orc_fis_gauss.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.orc_sincos2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_logf.restype = C.c_float
        L.orc_logf.argtypes = [C.c_float]
        L.orc_atan2f.restype = C.c_float
        L.orc_atan2f.argtypes = [C.c_float, C.c_float]
        L.orc_acosf.restype = C.c_float
        L.orc_acosf.argtypes = [C.c_float]
        L.orc_tex_lookup.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.orc_expf.restype = C.c_float
        L.orc_expf.argtypes = [C.c_float]
        L.orc_film_reflectance.restype = C.c_float
        L.orc_film_reflectance.argtypes = [C.c_float] * 5
        L.orc_fresnel_dielectric.restype = C.c_float
        L.orc_fresnel_dielectric.argtypes = [C.c_float, C.c_float]
        L.orc_pack_half2x16.restype = C.c_uint32
        L.orc_pack_half2x16.argtypes = [C.c_float, C.c_float]
        L.orc_unpack_half2x16.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
        L.orc_orthonormal_basis.argtypes = [C.POINTER(C.c_float)] * 3
        L.orc_trace_closest.restype = C.c_int
        L.orc_trace_closest.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                                        C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_trace_batch.restype = C.c_int
        L.orc_trace_batch.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                                      C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.orc_bsdf_debug.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib = L
    return _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class PackedScene:
    """Keeps the numpy buffers alive while the C structs point into them."""

    def __init__(self, scene):
        self.keep = []
        meshes = (OrcMesh * max(1, len(scene.meshes)))()
        for i, m in enumerate(scene.meshes):
            v = np.ascontiguousarray(m.vertices)
            f = np.ascontiguousarray(m.faces, np.uint32)
            it = np.ascontiguousarray(m.instance_transforms, np.float32).reshape(-1, 16)
            self.keep += [v, f, it]
            meshes[i].vertices = v.ctypes.data
            meshes[i].vertexCount = len(v)
            meshes[i].faces = f.ctypes.data
            meshes[i].faceCount = len(f)
            meshes[i].id = m.id
            meshes[i].isDoubleSided = int(m.double_sided)
            meshes[i].isLeftHanded = int(m.left_handed)
            meshes[i].visible = int(m.visible)
            meshes[i].transform = (C.c_float * 16)(*np.asarray(m.transform, np.float32).reshape(-1))
            meshes[i].instanceTransforms = it.ctypes.data
            meshes[i].instanceCount = len(it)
            meshes[i].material = m.material
            fid = getattr(m, "face_ids", None)
            if fid is not None:
                fid = np.ascontiguousarray(fid, np.int32); self.keep.append(fid); meshes[i].faceIds = fid.ctypes.data
            meshes[i].maxFaceId = int(getattr(m, "max_face_id", 0))
            if m.instance_ids is not None:
                iid = np.ascontiguousarray(m.instance_ids, np.int32); self.keep.append(iid); meshes[i].instanceIds = iid.ctypes.data
            for attr, ptr, cnt in (("primvars", "primvars", "primvarCount"), ("instancer_primvars", "instancerPrimvars", "instancerPrimvarCount")):
                pvs = getattr(m, attr, [])
                if pvs:
                    arr = (OrcPrimvar * len(pvs))()
                    for k, pv in enumerate(pvs):
                        d = np.ascontiguousarray(pv.data, np.int32 if int(pv.type) >= 4 else np.float32).reshape(-1)  # Int..Int4 keep their bits
                        self.keep.append(d)
                        arr[k].name = pv.name.encode(); arr[k].type = int(pv.type); arr[k].interpolation = int(pv.interpolation)
                        arr[k].data = d.ctypes.data; arr[k].floatCount = len(d)
                    self.keep.append(arr)
                    setattr(meshes[i], ptr, C.addressof(arr)); setattr(meshes[i], cnt, len(pvs))
        mats = (OrcMaterial * max(1, len(scene.materials)))()
        for i, m in enumerate(scene.materials):
            fill_material(mats[i], m)
        texs = (OrcTexture * max(1, len(getattr(scene, "textures", []))))()
        for i, t in enumerate(getattr(scene, "textures", [])):
            a = np.ascontiguousarray(t, np.float32)
            assert a.ndim == 3 and a.shape[2] == 4
            self.keep.append(a)
            texs[i].rgba, texs[i].width, texs[i].height = a.ctypes.data, a.shape[1], a.shape[0]
        dome = None
        if getattr(scene, "dome_light", None) is not None:
            d = scene.dome_light
            dome = OrcDomeLight(int(d.texture), (C.c_float * 4)(*d.rotation), _f3(d.base_emission))
        sl = (OrcSphereLight * max(1, len(scene.sphere_lights)))()
        for i, l in enumerate(scene.sphere_lights):
            sl[i] = OrcSphereLight(_f3(l.pos), _f3(l.base_emission), _f3(l.radius), l.diffuse, l.specular)
        dl = (OrcDistantLight * max(1, len(scene.distant_lights)))()
        for i, l in enumerate(scene.distant_lights):
            dl[i] = OrcDistantLight(_f3(l.direction), _f3(l.base_emission), l.angle, l.diffuse, l.specular)
        rl = (OrcRectLight * max(1, len(scene.rect_lights)))()
        for i, l in enumerate(scene.rect_lights):
            rl[i] = OrcRectLight(_f3(l.origin), _f3(l.t0), _f3(l.t1), _f3(l.base_emission), l.width, l.height, l.diffuse, l.specular)
        kl = (OrcDiskLight * max(1, len(scene.disk_lights)))()
        for i, l in enumerate(scene.disk_lights):
            kl[i] = OrcDiskLight(_f3(l.origin), _f3(l.t0), _f3(l.t1), _f3(l.base_emission), l.radius_x, l.radius_y, l.diffuse, l.specular)
        self.keep += [meshes, mats, sl, dl, rl, kl, texs, dome]
        s = OrcScene()
        s.meshes = C.addressof(meshes); s.meshCount = len(scene.meshes)
        s.materials = C.addressof(mats); s.materialCount = len(scene.materials)
        s.sphereLights = C.addressof(sl); s.sphereLightCount = len(scene.sphere_lights)
        s.distantLights = C.addressof(dl); s.distantLightCount = len(scene.distant_lights)
        s.rectLights = C.addressof(rl); s.rectLightCount = len(scene.rect_lights)
        s.diskLights = C.addressof(kl); s.diskLightCount = len(scene.disk_lights)
        s.textures = C.addressof(texs); s.textureCount = len(getattr(scene, "textures", []))
        s.dome = C.addressof(dome) if dome is not None else None
        self.c = s


def _camera(cam) -> OrcCamera:
    return OrcCamera(_f3(cam.position), _f3(cam.forward), _f3(cam.up), cam.vfov, cam.f_stop, cam.focus_distance,
                     cam.focal_length, cam.clip_start, cam.clip_end, cam.exposure)


def _settings(rs, sample_offset=0) -> OrcSettings:
    s = OrcSettings()
    s.clippingPlanes = int(rs.clipping_planes)
    s.depthOfField = int(rs.depth_of_field)
    s.domeLightCameraVisible = int(rs.dome_light_camera_visible)
    s.filterImportanceSampling = int(rs.filter_importance_sampling)
    s.jitteredSampling = int(rs.jittered_sampling)
    s.nextEventEstimation = int(rs.next_event_estimation)
    s.progressiveAccumulation = int(rs.progressive_accumulation)
    s.maxBounces = rs.max_bounces
    s.rrBounceOffset = rs.rr_bounce_offset
    s.spp = rs.spp
    s.sampleOffset = sample_offset
    s.frame = float(getattr(rs, "frame", 0.0))
    s.lightIntensityMultiplier = rs.light_intensity_multiplier
    s.maxSampleValue = rs.max_sample_value
    s.rrInvMinTermProb = rs.rr_inv_min_term_prob
    s.metersPerSceneUnit = rs.meters_per_scene_unit
    s.mediumStackSize = rs.medium_stack_size
    s.maxVolumeWalkLength = rs.max_volume_walk_length
    s.clearColor = (C.c_float * 4)(*rs.clear_color)
    return s


def render(scene, settings, width, height, rows=None, sample_offset=0, prev_color=None, threads=1, row_list=None):
    """Renders with the oracle.  Returns (color float32 [rows,width,4] with row 0 = bottom, counters dict).  `row_list`: explicit image rows instead of a
    contiguous range (one scene preparation for rows that are not adjacent)."""
    L = lib()
    ps = PackedScene(scene)
    r0, r1 = rows if rows is not None else (0, height)
    cam, st = _camera(scene.camera), _settings(settings, sample_offset)
    cnt = OrcCounters()
    prev = None
    if prev_color is not None:
        prev = np.ascontiguousarray(prev_color, np.float32)
    if row_list is not None:
        rl = np.ascontiguousarray(row_list, np.uint32)
        out = np.zeros((len(rl), width, 4), np.float32)
        rg = OrcRegion(width, height, 0, height)
        L.orc_render_rows.restype = C.c_int
        L.orc_render_rows.argtypes = [C.c_void_p] * 4 + [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        rc = L.orc_render_rows(C.addressof(ps.c), C.addressof(cam), C.addressof(st), C.addressof(rg), len(rl), rl.ctypes.data,
                               prev.ctypes.data if prev is not None else None, out.ctypes.data, C.addressof(cnt), threads)
    else:
        out = np.zeros((r1 - r0, width, 4), np.float32)
        rg = OrcRegion(width, height, r0, r1)
        rc = L.orc_render(C.addressof(ps.c), C.addressof(cam), C.addressof(st), C.addressof(rg),
                          prev.ctypes.data if prev is not None else None, out.ctypes.data, C.addressof(cnt), threads)
    if rc != 0:
        raise RuntimeError(f"orc_render failed with code {rc}")
    counters = {"samples": cnt.samples, "segments": cnt.segments, "shadow_rays": cnt.shadowRays, "hits": cnt.hits,
                "bounce_histogram": [int(x) for x in cnt.bounceHistogram]}
    return out, counters


def trace_rays(scene, origins, dirs, t_min=0.0, t_max=3.0e38):
    L = lib()
    ps = PackedScene(scene)
    o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    n = len(o)
    tuv = np.zeros((n, 3), np.float32)
    ip = np.zeros((n, 2), np.int32)
    FP = C.POINTER(C.c_float)
    L.orc_trace_batch(C.addressof(ps.c), n, o.ctypes.data_as(FP), d.ctypes.data_as(FP), t_min, t_max, tuv.ctypes.data_as(FP),
                      ip.ctypes.data_as(C.POINTER(C.c_int32)))
    return tuv, ip


def bsdf_debug(material, items):
    """items: float32 [n,22] (normal, tangentU, tangentV, geomNormal, k1, k2, xi[4]) -> float32 [n,15]."""
    L = lib()
    m = OrcMaterial()
    fill_material(m, material)
    for slot in range(TEX_SLOT_COUNT):
        m.tex[slot].texture = -1  # the BSDF known-answer entry point evaluates the constant parameter block
    a = np.ascontiguousarray(items, np.float32).reshape(-1, 22)
    out = np.zeros((len(a), 15), np.float32)
    FP = C.POINTER(C.c_float)
    L.orc_bsdf_debug(C.addressof(m), len(a), a.ctypes.data_as(FP), out.ctypes.data_as(FP))
    return out


def render_aovs(scene, settings, width, height, names, clear_values=None, sample_offset=0, prev=None, rows=None):
    """Non-colour AOVs through the oracle.  names: subset of AOV_IDS; clear_values: {name: 4 floats or 1 int};
    prev: {name: array} previous-call contents for the accumulating AOVs.  Returns {name: array}."""
    L = lib()
    L.orc_render_aovs.restype = C.c_int
    L.orc_render_aovs.argtypes = [C.c_void_p] * 5
    ps = PackedScene(scene)
    r0, r1 = rows if rows is not None else (0, height)
    n = (r1 - r0) * width
    A = OrcAovs()
    out = {}
    for name in names:
        aid = AOV_IDS[name]
        cv = (clear_values or {}).get(name, 0)
        if name in ("objectId", "faceId", "instanceId"):
            arr = np.zeros(n, np.int32)
            A.clear[aid][0] = np.frombuffer(np.int32(cv).tobytes(), np.float32)[0]
        elif name == "depth":
            arr = np.zeros(n, np.float32)
            A.clear[aid][0] = float(cv)
        else:
            arr = np.zeros((n, 4), np.float32)
            cv4 = (list(cv) + [0, 0, 0, 0])[:4] if hasattr(cv, "__len__") else [cv] * 4
            for k in range(4):
                A.clear[aid][k] = float(cv4[k])
            if prev and name in prev:
                arr[:] = np.asarray(prev[name], np.float32).reshape(n, 4)
        out[name] = arr
        setattr(A, name, arr.ctypes.data)
    cam, st = _camera(scene.camera), _settings(settings, sample_offset)
    rg = OrcRegion(width, height, r0, r1)
    rc = L.orc_render_aovs(C.addressof(ps.c), C.addressof(cam), C.addressof(st), C.addressof(rg), C.addressof(A))
    if rc != 0:
        raise RuntimeError(f"orc_render_aovs failed with code {rc}")
    return {k: (v.reshape(r1 - r0, width, 4) if v.ndim == 2 else v.reshape(r1 - r0, width)) for k, v in out.items()}


def tex_lookup(texture, u, v, wrap_u=1, wrap_v=1):
    """tex_lookup_float4_2d of the oracle's texture runtime on a float32 [h, w, 4] image."""
    a = np.ascontiguousarray(texture, np.float32)
    t = OrcTexture(a.ctypes.data, a.shape[1], a.shape[0])
    out = (C.c_float * 4)()
    lib().orc_tex_lookup(C.addressof(t), float(u), float(v), int(wrap_u), int(wrap_v), out)
    return np.array(out[:], np.float32)
