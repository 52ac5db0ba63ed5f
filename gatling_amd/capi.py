"""ctypes binding of the product library ``gatling_amd/libgatling_gi.so`` (C ABI: ``include/gi_c.h``).

This is the harness-side mirror of the reference's gi interface
(``/root/reference/src/gi/gtl/gi/Gi.h:199-261``): same entry points, same argument meaning.  There is no CPU
fallback -- if the HIP library is missing or no GPU is visible, loading / ``giCInitialize`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .scene import P_COUNT, SceneDesc, RenderSettings

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgatling_gi.so")
if os.environ.get("GATLING_GI_LIB"):  # experiments: a variant build of the same library (tools/build_variant.py), e.g. other kernel compile flags
    LIB_PATH = os.path.abspath(os.environ["GATLING_GI_LIB"])

GI_C_OK = 0
AOV_COLOR = 0
FORMAT_INT32, FORMAT_FLOAT32, FORMAT_FLOAT32_VEC4 = 0, 1, 2
OPTION_COUNT_TRAVERSAL, OPTION_KERNEL_TIMERS, OPTION_POOL_SLOTS, OPTION_SAMPLE_BUFFER_MB, OPTION_TRACE_DYNAMIC, OPTION_TWO_LEVEL, OPTION_FUSED_PATH, OPTION_DEVICES = 1, 2, 3, 4, 5, 6, 7, 8


class GiCCameraDesc(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("forward", C.c_float * 3), ("up", C.c_float * 3), ("vfov", C.c_float),
                ("fStop", C.c_float), ("focusDistance", C.c_float), ("focalLength", C.c_float), ("clipStart", C.c_float),
                ("clipEnd", C.c_float), ("exposure", C.c_float)]


class GiCMeshDesc(C.Structure):
    _fields_ = [("faceCount", C.c_uint32), ("faces", C.c_void_p), ("faceIds", C.c_void_p), ("id", C.c_int32),
                ("isDoubleSided", C.c_int32), ("isLeftHanded", C.c_int32), ("name", C.c_char_p), ("maxFaceId", C.c_uint32),
                ("vertexCount", C.c_uint32), ("vertices", C.c_void_p)]


class GiCRenderSettings(C.Structure):
    _fields_ = [("clippingPlanes", C.c_int32), ("depthOfField", C.c_int32), ("domeLightCameraVisible", C.c_int32),
                ("filterImportanceSampling", C.c_int32), ("frame", C.c_float), ("jitteredSampling", C.c_int32),
                ("lightIntensityMultiplier", C.c_float), ("maxBounces", C.c_uint32), ("maxSampleValue", C.c_float),
                ("maxVolumeWalkLength", C.c_uint32), ("mediumStackSize", C.c_uint32), ("metersPerSceneUnit", C.c_float),
                ("nextEventEstimation", C.c_int32), ("progressiveAccumulation", C.c_int32), ("rrBounceOffset", C.c_uint32),
                ("rrInvMinTermProb", C.c_float), ("spp", C.c_uint32), ("time", C.c_float)]


class GiCAovBinding(C.Structure):
    _fields_ = [("aovId", C.c_int32), ("clearValue", C.c_uint8 * 16), ("renderBuffer", C.c_void_p)]


class GiCRenderParams(C.Structure):
    _fields_ = [("aovBindings", C.POINTER(GiCAovBinding)), ("aovBindingCount", C.c_uint32), ("camera", GiCCameraDesc),
                ("domeLight", C.c_void_p), ("renderSettings", GiCRenderSettings), ("scene", C.c_void_p),
                ("rowBegin", C.c_uint32), ("rowEnd", C.c_uint32), ("rowStride", C.c_uint32)]


class GiCMaterialDesc(C.Structure):
    _fields_ = [("klass", C.c_uint32), ("flags", C.c_uint32), ("p", C.c_float * P_COUNT)]


class GiCRenderStats(C.Structure):
    _fields_ = [("renderMs", C.c_double), ("bvhBuildMs", C.c_double), ("uploadMs", C.c_double), ("traceMs", C.c_double),
                ("shadeMs", C.c_double), ("raygenMs", C.c_double), ("shadowMs", C.c_double),
                ("samples", C.c_uint64), ("segments", C.c_uint64), ("shadowRays", C.c_uint64), ("nodesVisited", C.c_uint64),
                ("trisTested", C.c_uint64), ("shadowNodesVisited", C.c_uint64), ("shadowTrisTested", C.c_uint64),
                ("iterations", C.c_uint32), ("traceLaunches", C.c_uint32), ("nodeCount", C.c_uint32), ("triangleCount", C.c_uint32),
                ("fusedPath", C.c_uint32), ("batches", C.c_uint32), ("poolSlots", C.c_uint32), ("inactiveTriangleCount", C.c_uint32)]


# every symbol include/gi_c.h declares: (name, restype, argtypes)
_P, _F, _U, _I = C.c_void_p, C.c_float, C.c_uint32, C.c_int32
_FP = C.POINTER(C.c_float)
class GiCTextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba", C.c_void_p)]


class GiCTextureBinding(C.Structure):
    _fields_ = [("texture", C.c_void_p), ("wrapS", C.c_int32), ("wrapT", C.c_int32), ("channel", C.c_int32),
                ("scale", C.c_float * 4), ("bias", C.c_float * 4)]


class GiCPrimvarData(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_int32), ("interpolation", C.c_int32), ("data", C.c_void_p), ("dataSize", C.c_uint64)]



# asset reader / image loader hooks (include/gi_c.h: giCRegisterAssetReader, giCSetImageLoader)
ASSET_OPEN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_char_p)
ASSET_SIZE = C.CFUNCTYPE(C.c_uint64, C.c_void_p, C.c_void_p)
ASSET_DATA = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p)
ASSET_CLOSE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


class GiCAssetReader(C.Structure):
    _fields_ = [("user", C.c_void_p), ("open", ASSET_OPEN), ("size", ASSET_SIZE), ("data", ASSET_DATA), ("close", ASSET_CLOSE)]


class GiCDecodedImage(C.Structure):
    _fields_ = [("format", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("reserved", C.c_uint32), ("pixels", C.c_void_p), ("handle", C.c_void_p)]


IMAGE_LOAD = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_int32, C.POINTER(GiCDecodedImage))
IMAGE_RELEASE = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(GiCDecodedImage))
IMAGE_RGBA8_UNORM, IMAGE_RGB16_FLOAT, IMAGE_RGBA16_FLOAT, IMAGE_R32_FLOAT, IMAGE_RGBA32_FLOAT = 1, 2, 3, 4, 5


class GiCImageLoader(C.Structure):
    _fields_ = [("user", C.c_void_p), ("load", IMAGE_LOAD), ("release", IMAGE_RELEASE)]


SYMBOLS = [
    ("giCInitialize", C.c_int, [C.c_int]), ("giCInitializeDevices", C.c_int, [C.POINTER(C.c_int32), C.c_uint32]), ("giCGetDeviceCount", C.c_uint32, []), ("giCGetApiVersion", C.c_uint32, []), ("giCGetDevicePeerAccess", C.c_int32, [C.c_uint32]), ("giCTerminate", None, []), ("giCGetLastError", C.c_char_p, []),
    ("giCCreateMaterial", _P, [_P, C.c_char_p, C.POINTER(GiCMaterialDesc)]), ("giCDestroyMaterial", None, [_P]),
    ("giCCreateMesh", _P, [_P, C.POINTER(GiCMeshDesc)]), ("giCSetMeshTransform", None, [_P, _FP]),
    ("giCSetMeshInstanceTransforms", None, [_P, _U, _FP]), ("giCSetMeshInstanceIds", None, [_P, _U, C.POINTER(C.c_int32)]),
    ("giCSetMeshMaterial", None, [_P, _P]), ("giCSetMeshVisibility", None, [_P, _I]), ("giCDestroyMesh", None, [_P]),
    ("giCRender", C.c_int, [C.POINTER(GiCRenderParams)]),
    ("giCCreateScene", _P, []), ("giCDestroyScene", None, [_P]),
    ("giCCreateSphereLight", _P, [_P]), ("giCDestroySphereLight", None, [_P, _P]), ("giCSetSphereLightPosition", None, [_P, _FP]),
    ("giCSetSphereLightBaseEmission", None, [_P, _FP]), ("giCSetSphereLightRadius", None, [_P, _F, _F, _F]),
    ("giCSetSphereLightDiffuseSpecular", None, [_P, _F, _F]),
    ("giCCreateDistantLight", _P, [_P]), ("giCDestroyDistantLight", None, [_P, _P]), ("giCSetDistantLightDirection", None, [_P, _FP]),
    ("giCSetDistantLightBaseEmission", None, [_P, _FP]), ("giCSetDistantLightAngle", None, [_P, _F]),
    ("giCSetDistantLightDiffuseSpecular", None, [_P, _F, _F]),
    ("giCCreateRectLight", _P, [_P]), ("giCDestroyRectLight", None, [_P, _P]), ("giCSetRectLightOrigin", None, [_P, _FP]),
    ("giCSetRectLightTangents", None, [_P, _FP, _FP]), ("giCSetRectLightBaseEmission", None, [_P, _FP]),
    ("giCSetRectLightDimensions", None, [_P, _F, _F]), ("giCSetRectLightDiffuseSpecular", None, [_P, _F, _F]),
    ("giCCreateDiskLight", _P, [_P]), ("giCDestroyDiskLight", None, [_P, _P]), ("giCSetDiskLightOrigin", None, [_P, _FP]),
    ("giCSetDiskLightTangents", None, [_P, _FP, _FP]), ("giCSetDiskLightBaseEmission", None, [_P, _FP]),
    ("giCSetDiskLightRadius", None, [_P, _F, _F]), ("giCSetDiskLightDiffuseSpecular", None, [_P, _F, _F]),
    ("giCCreateDomeLight", _P, [_P, C.c_char_p]), ("giCDestroyDomeLight", None, [_P]), ("giCSetDomeLightRotation", None, [_P, _FP]),
    ("giCSetDomeLightBaseEmission", None, [_P, _FP]), ("giCSetDomeLightDiffuseSpecular", None, [_P, _F, _F]),
    ("giCSetDomeLightTexture", None, [_P, _P]),
    ("giCCreateTexture", _P, [_P, C.POINTER(GiCTextureDesc)]), ("giCDestroyTexture", None, [_P]),
    ("giCCreateTextureFromFile", _P, [_P, C.c_char_p, _I]),
    ("giCDebugDecodeImage", C.c_int, [C.c_char_p, _I, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _FP, C.c_uint64]),
    ("giCRegisterAssetReader", None, [C.POINTER(GiCAssetReader)]), ("giCSetImageLoader", None, [C.POINTER(GiCImageLoader)]),
    ("giCSetMaterialTexture", C.c_int, [_P, _I, C.POINTER(GiCTextureBinding)]),
    ("giCSetMaterialTextureTransform", C.c_int, [_P, _I, _FP]),
    ("giCSetMeshPrimvars", C.c_int, [_P, _U, C.POINTER(GiCPrimvarData)]), ("giCSetMeshInstancerPrimvars", C.c_int, [_P, _U, C.POINTER(GiCPrimvarData)]),
    ("giCSetMaterialPrimvarInput", C.c_int, [_P, _I, C.c_char_p]),
    ("giCCreateRenderBuffer", _P, [_U, _U, _I]), ("giCDestroyRenderBuffer", None, [_P]), ("giCGetRenderBufferMem", _P, [_P]),
    ("giCGetRenderBufferDeviceMem", _P, [_P]), ("giCSetRenderBufferDeviceOnly", None, [_P, _I]),
    ("giCGetRenderStats", C.c_int, [_P, C.POINTER(GiCRenderStats)]), ("giCSetSceneOption", C.c_int, [_P, _I, _I]),
    ("giCTraceRays", C.c_int, [_P, _U, _FP, _FP, _F, _F, _FP, C.POINTER(C.c_int32)]),
    ("giCDebugEvalBsdf", C.c_int, [C.POINTER(GiCMaterialDesc), _U, _FP, _FP]),
    ("giCDebugShadeClass", C.c_int, [C.POINTER(GiCMaterialDesc)]),
    ("giCDebugCheckSqrt", C.c_int64, [C.c_uint32, C.c_uint64]),
    ("giCDebugValidateBvh", C.c_int, [_FP, _U, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("giCDebugValidatePartitionedBvh", C.c_int, [_FP, _U, _U, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("giCDebugTexRuntime", C.c_int, [_FP, _U, _U, _U, _U, _FP, _FP]),
]

_lib = None


def load_library():
    """Loads libgatling_gi.so and types every entry point.  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _fp(values):
    arr = (C.c_float * len(values))(*[float(v) for v in values])
    return arr


class GiError(RuntimeError):
    pass


_initialized = False


def initialize(device: int = 0, devices=None):
    """One giCInitialize per process (Gi.cpp:244-259).  `devices` = list of HIP ordinals -> giCInitializeDevices: whole-frame renders are then dealt
    row-wise to all of them inside the library (the first is the primary); $GATLING_DEVICES does the same for callers that pass nothing."""
    global _initialized
    L = load_library()
    if not _initialized:
        if devices:
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            rc = L.giCInitializeDevices(arr, len(devices))
        else:
            rc = L.giCInitialize(device)
        if rc != GI_C_OK:
            raise GiError("giCInitialize failed: " + L.giCGetLastError().decode())
        _initialized = True
    return L


def _camera(cam) -> GiCCameraDesc:
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    return GiCCameraDesc(f3(cam.position), f3(cam.forward), f3(cam.up), cam.vfov, cam.f_stop, cam.focus_distance,
                         cam.focal_length, cam.clip_start, cam.clip_end, cam.exposure)


def _settings(rs: RenderSettings) -> GiCRenderSettings:
    s = GiCRenderSettings()
    s.clippingPlanes = int(rs.clipping_planes); s.depthOfField = int(rs.depth_of_field)
    s.domeLightCameraVisible = int(rs.dome_light_camera_visible); s.filterImportanceSampling = int(rs.filter_importance_sampling)
    s.frame = float(getattr(rs, "frame", 0.0)); s.jitteredSampling = int(rs.jittered_sampling); s.lightIntensityMultiplier = rs.light_intensity_multiplier
    s.maxBounces = rs.max_bounces; s.maxSampleValue = rs.max_sample_value; s.maxVolumeWalkLength = rs.max_volume_walk_length
    s.mediumStackSize = rs.medium_stack_size; s.metersPerSceneUnit = rs.meters_per_scene_unit
    s.nextEventEstimation = int(rs.next_event_estimation); s.progressiveAccumulation = int(rs.progressive_accumulation)
    s.rrBounceOffset = rs.rr_bounce_offset; s.rrInvMinTermProb = rs.rr_inv_min_term_prob; s.spp = rs.spp; s.time = 0.0
    return s


class Scene:
    """Feeds a :class:`SceneDesc` through the gi C ABI the way hdGatling feeds Hydra prims through Gi.h."""

    def __init__(self, desc: SceneDesc, device: int = 0, mtlx_materials=None):
        """`mtlx_materials`: {material index: MaterialX document string} -- those materials are created through the gtl shim's MaterialX reader
        (gtl::giCreateMaterialFromMtlxStr, what hdGatling calls) instead of giCCreateMaterial with the parameter block."""
        self.L = initialize(device)
        L = self.L
        self.desc = desc
        self.handle = L.giCCreateScene()
        if not self.handle:
            raise GiError("giCCreateScene failed")
        self.materials, self.meshes, self.lights, self.textures, self.dome = [], [], [], [], None
        for t in getattr(desc, "textures", []):
            a = np.ascontiguousarray(t, np.float32)
            td = GiCTextureDesc(a.shape[1], a.shape[0], a.ctypes.data)
            h = L.giCCreateTexture(self.handle, C.byref(td))  # copies the pixels
            if not h:
                raise GiError("giCCreateTexture failed: " + L.giCGetLastError().decode())
            self.textures.append(h)
        for mi, m in enumerate(desc.materials):
            if mtlx_materials and mi in mtlx_materials:
                L.gtlCreateMaterialFromMtlxStrC.restype = C.c_void_p
                L.gtlCreateMaterialFromMtlxStrC.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
                h = L.gtlCreateMaterialFromMtlxStrC(self.handle, m.name.encode(), mtlx_materials[mi].encode())
                if not h:
                    raise GiError("gtl::giCreateMaterialFromMtlxStr refused the document of material %d" % mi)
                self.materials.append(h)
                continue
            md = GiCMaterialDesc(m.klass, 0, (C.c_float * P_COUNT)(*np.asarray(m.params, np.float32)))
            h = L.giCCreateMaterial(self.handle, m.name.encode(), C.byref(md))
            if not h:
                raise GiError("giCCreateMaterial failed: " + L.giCGetLastError().decode())
            for slot, b in getattr(m, "textures", {}).items():
                tb = GiCTextureBinding(self.textures[b.texture], int(b.wrap_s), int(b.wrap_t), int(b.channel), (C.c_float * 4)(*b.scale), (C.c_float * 4)(*b.bias))
                if L.giCSetMaterialTexture(h, int(slot), C.byref(tb)) != GI_C_OK:
                    raise GiError("giCSetMaterialTexture failed: " + L.giCGetLastError().decode())
                if getattr(b, "transform", None) is not None and L.giCSetMaterialTextureTransform(h, int(slot), _fp(b.transform)) != GI_C_OK:
                    raise GiError("giCSetMaterialTextureTransform failed: " + L.giCGetLastError().decode())
            for slot, name in getattr(m, "primvar_inputs", {}).items():
                if L.giCSetMaterialPrimvarInput(h, int(slot), name.encode()) != GI_C_OK:
                    raise GiError("giCSetMaterialPrimvarInput failed: " + L.giCGetLastError().decode())
            self.materials.append(h)
        if getattr(desc, "dome_light", None) is not None:
            d = desc.dome_light
            self.dome = L.giCCreateDomeLight(self.handle, b"")
            if d.texture >= 0:
                L.giCSetDomeLightTexture(self.dome, self.textures[d.texture])
            L.giCSetDomeLightRotation(self.dome, _fp(d.rotation)); L.giCSetDomeLightBaseEmission(self.dome, _fp(d.base_emission))
            L.giCSetDomeLightDiffuseSpecular(self.dome, d.diffuse, d.specular)
        for m in desc.meshes:
            v = np.ascontiguousarray(m.vertices)
            f = np.ascontiguousarray(m.faces, np.uint32)
            fid = np.ascontiguousarray(m.face_ids, np.int32) if m.face_ids is not None else None
            d = GiCMeshDesc(len(f), f.ctypes.data, fid.ctypes.data if fid is not None else None, m.id, int(m.double_sided),
                            int(m.left_handed), m.name.encode(), int(m.max_face_id), len(v), v.ctypes.data)
            h = L.giCCreateMesh(self.handle, C.byref(d))
            if not h:
                raise GiError("giCCreateMesh failed: " + L.giCGetLastError().decode())
            L.giCSetMeshTransform(h, _fp(np.asarray(m.transform, np.float32).reshape(-1)))
            it = np.ascontiguousarray(m.instance_transforms, np.float32).reshape(-1, 16)
            L.giCSetMeshInstanceTransforms(h, len(it), it.ctypes.data_as(_FP))
            if m.instance_ids is not None:
                ids = np.ascontiguousarray(m.instance_ids, np.int32)
                L.giCSetMeshInstanceIds(h, len(ids), ids.ctypes.data_as(C.POINTER(C.c_int32)))
            for attr, fn in (("primvars", L.giCSetMeshPrimvars), ("instancer_primvars", L.giCSetMeshInstancerPrimvars)):
                pvs = getattr(m, attr, [])
                if pvs:
                    arr, keep = (GiCPrimvarData * len(pvs))(), []
                    for k, pv in enumerate(pvs):
                        d = np.ascontiguousarray(pv.data, np.int32 if int(pv.type) >= 4 else np.float32).reshape(-1)  # Int..Int4 (Gi.h:76-79)
                        keep.append(d)
                        arr[k] = GiCPrimvarData(pv.name.encode(), int(pv.type), int(pv.interpolation), d.ctypes.data, d.nbytes)
                    if fn(h, len(pvs), arr) != GI_C_OK:  # copies the data
                        raise GiError("giCSetMesh*Primvars failed: " + L.giCGetLastError().decode())
            if m.material >= 0:
                L.giCSetMeshMaterial(h, self.materials[m.material])
            L.giCSetMeshVisibility(h, int(m.visible))
            self.meshes.append(h)
        for l in desc.sphere_lights:
            h = L.giCCreateSphereLight(self.handle)
            L.giCSetSphereLightPosition(h, _fp(l.pos)); L.giCSetSphereLightBaseEmission(h, _fp(l.base_emission))
            L.giCSetSphereLightRadius(h, *[float(x) for x in l.radius]); L.giCSetSphereLightDiffuseSpecular(h, l.diffuse, l.specular)
            self.lights.append(("sphere", h))
        for l in desc.distant_lights:
            h = L.giCCreateDistantLight(self.handle)
            L.giCSetDistantLightDirection(h, _fp(l.direction)); L.giCSetDistantLightBaseEmission(h, _fp(l.base_emission))
            L.giCSetDistantLightAngle(h, l.angle); L.giCSetDistantLightDiffuseSpecular(h, l.diffuse, l.specular)
            self.lights.append(("distant", h))
        for l in desc.rect_lights:
            h = L.giCCreateRectLight(self.handle)
            L.giCSetRectLightOrigin(h, _fp(l.origin)); L.giCSetRectLightTangents(h, _fp(l.t0), _fp(l.t1))
            L.giCSetRectLightBaseEmission(h, _fp(l.base_emission)); L.giCSetRectLightDimensions(h, l.width, l.height)
            L.giCSetRectLightDiffuseSpecular(h, l.diffuse, l.specular)
            self.lights.append(("rect", h))
        for l in desc.disk_lights:
            h = L.giCCreateDiskLight(self.handle)
            L.giCSetDiskLightOrigin(h, _fp(l.origin)); L.giCSetDiskLightTangents(h, _fp(l.t0), _fp(l.t1))
            L.giCSetDiskLightBaseEmission(h, _fp(l.base_emission)); L.giCSetDiskLightRadius(h, l.radius_x, l.radius_y)
            L.giCSetDiskLightDiffuseSpecular(h, l.diffuse, l.specular)
            self.lights.append(("disk", h))
        self._buffers = {}

    def set_mesh_transform(self, mesh_index: int, matrix):
        """giSetMeshTransform (Gi.h:213): a transform-only edit -- the next render updates the scene incrementally."""
        m = np.ascontiguousarray(matrix, np.float32).reshape(16)
        self.desc.meshes[mesh_index].transform = m.reshape(4, 4).copy()
        self.L.giCSetMeshTransform(self.meshes[mesh_index], m.ctypes.data_as(_FP))

    def set_mesh_instance_transforms(self, mesh_index: int, transforms):
        """giSetMeshInstanceTransforms (Gi.h:214)."""
        t = np.ascontiguousarray(transforms, np.float32).reshape(-1, 16)
        self.desc.meshes[mesh_index].instance_transforms = t.reshape(-1, 4, 4).copy()
        self.L.giCSetMeshInstanceTransforms(self.meshes[mesh_index], len(t), t.ctypes.data_as(_FP))

    def set_option(self, option: int, value: int):
        if self.L.giCSetSceneOption(self.handle, option, value) != GI_C_OK:
            raise GiError("giCSetSceneOption failed")

    def color_buffer(self, width, height):
        key = ("color", width, height)
        if key not in self._buffers:
            rb = self.L.giCCreateRenderBuffer(width, height, FORMAT_FLOAT32_VEC4)
            if not rb:
                raise GiError("giCCreateRenderBuffer failed: " + self.L.giCGetLastError().decode())
            self._buffers[key] = rb
        return self._buffers[key]

    def render(self, settings: RenderSettings, width: int, height: int, rows=None, device_only=False, row_stride=1, copy=True):
        """One giCRender call.  Returns the colour AOV as float32 [rows, width, 4] (row 0 = bottom) -- the library-owned host
        memory of giCGetRenderBufferMem copied out, or with ``copy=False`` a VIEW of it (the reference's contract, Gi.cpp:3003-3006:
        the pointer stays valid, its contents are those of the last render, until the buffer is destroyed) -- or None when ``device_only``."""
        L = self.L
        rb = self.color_buffer(width, height)
        L.giCSetRenderBufferDeviceOnly(rb, int(device_only))
        binding = GiCAovBinding()
        binding.aovId = AOV_COLOR
        clear = np.asarray(settings.clear_color, np.float32)
        C.memmove(binding.clearValue, clear.ctypes.data, 16)
        binding.renderBuffer = rb
        p = GiCRenderParams()
        p.aovBindings = C.pointer(binding)
        p.aovBindingCount = 1
        p.camera = _camera(self.desc.camera)
        p.domeLight = self.dome
        p.renderSettings = _settings(settings)
        p.scene = self.handle
        r0, r1 = rows if rows is not None else (0, height)
        p.rowBegin, p.rowEnd, p.rowStride = r0, r1, row_stride
        if L.giCRender(C.byref(p)) != GI_C_OK:
            raise GiError("giCRender failed: " + L.giCGetLastError().decode())
        if device_only:
            return None
        mem = L.giCGetRenderBufferMem(rb)
        full = np.ctypeslib.as_array(C.cast(mem, C.POINTER(C.c_float)), shape=(height, width, 4))
        return full[r0:r1:row_stride].copy() if copy else full[r0:r1:row_stride]

    AOVS = {"normal": (1, FORMAT_FLOAT32_VEC4), "nee": (2, FORMAT_FLOAT32_VEC4), "barycentrics": (3, FORMAT_FLOAT32_VEC4),
            "texcoords": (4, FORMAT_FLOAT32_VEC4), "bounces": (5, FORMAT_FLOAT32_VEC4), "clockCycles": (6, FORMAT_FLOAT32_VEC4), "opacity": (7, FORMAT_FLOAT32_VEC4),
            "tangents": (8, FORMAT_FLOAT32_VEC4), "bitangents": (9, FORMAT_FLOAT32_VEC4), "thinWalled": (10, FORMAT_FLOAT32_VEC4),
            "objectId": (11, FORMAT_INT32), "depth": (12, FORMAT_FLOAT32), "faceId": (13, FORMAT_INT32), "instanceId": (14, FORMAT_INT32),
            "doubleSided": (15, FORMAT_FLOAT32_VEC4), "albedo": (16, FORMAT_FLOAT32_VEC4)}

    def render_aovs(self, settings: RenderSettings, width: int, height: int, names, clear_values=None, with_color=True, rows=None, row_stride=1):
        """One giCRender call with the colour AOV (optional) plus the named non-colour AOVs bound (Gi.h:36-56, 161-166).
        Returns {name: array}; vec3 AOVs come back as [h, w, 4] (the shader writes .xyz only), ids / depth as [h, w]."""
        L = self.L
        r0, r1 = rows if rows is not None else (0, height)
        bindings = []
        bufs = {}
        if with_color:
            bufs["color"] = (self.color_buffer(width, height), FORMAT_FLOAT32_VEC4, np.asarray(settings.clear_color, np.float32).tobytes(), AOV_COLOR)
        for name in names:
            aid, fmt = self.AOVS[name]
            key = ("aov", name, width, height)
            if key not in self._buffers:
                self._buffers[key] = L.giCCreateRenderBuffer(width, height, fmt)
            cv = (clear_values or {}).get(name, 0)
            if fmt == FORMAT_INT32:
                raw = np.int32(cv).tobytes() + b"\0" * 12
            elif fmt == FORMAT_FLOAT32:
                raw = np.float32(cv).tobytes() + b"\0" * 12
            else:
                cv4 = (list(cv) + [0, 0, 0, 0])[:4] if hasattr(cv, "__len__") else [cv] * 4
                raw = np.asarray(cv4, np.float32).tobytes()
            bufs[name] = (self._buffers[key], fmt, raw, aid)
        arr = (GiCAovBinding * len(bufs))()
        for i, (name, (rb, fmt, raw, aid)) in enumerate(bufs.items()):
            arr[i].aovId = aid
            C.memmove(arr[i].clearValue, raw, 16)
            arr[i].renderBuffer = rb
        p = GiCRenderParams()
        p.aovBindings = C.cast(arr, C.POINTER(GiCAovBinding)); p.aovBindingCount = len(bufs)
        p.camera = _camera(self.desc.camera); p.domeLight = self.dome; p.renderSettings = _settings(settings); p.scene = self.handle
        p.rowBegin, p.rowEnd, p.rowStride = r0, r1, row_stride
        if L.giCRender(C.byref(p)) != GI_C_OK:
            raise GiError("giCRender failed: " + L.giCGetLastError().decode())
        out = {}
        for name, (rb, fmt, raw, aid) in bufs.items():
            mem = L.giCGetRenderBufferMem(rb)
            if fmt == FORMAT_FLOAT32_VEC4:
                a = np.ctypeslib.as_array(C.cast(mem, C.POINTER(C.c_float)), shape=(height, width, 4))
            elif fmt == FORMAT_FLOAT32:
                a = np.ctypeslib.as_array(C.cast(mem, C.POINTER(C.c_float)), shape=(height, width))
            else:
                a = np.ctypeslib.as_array(C.cast(mem, C.POINTER(C.c_int32)), shape=(height, width))
            out[name] = a[r0:r1:row_stride].copy()
        return out

    def device_pointer(self, width, height) -> int:
        return int(self.L.giCGetRenderBufferDeviceMem(self.color_buffer(width, height)))

    def stats(self) -> dict:
        s = GiCRenderStats()
        if self.L.giCGetRenderStats(self.handle, C.byref(s)) != GI_C_OK:
            raise GiError("giCGetRenderStats failed")
        return {name: getattr(s, name) for name, _ in GiCRenderStats._fields_}

    def trace_rays(self, origins, dirs, t_min=0.0, t_max=3.0e38):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(o)
        tuv = np.zeros((n, 3), np.float32)
        ip = np.zeros((n, 2), np.int32)
        rc = self.L.giCTraceRays(self.handle, n, o.ctypes.data_as(_FP), d.ctypes.data_as(_FP), t_min, t_max,
                                 tuv.ctypes.data_as(_FP), ip.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc < 0:
            raise GiError("giCTraceRays failed: " + self.L.giCGetLastError().decode())
        return tuv, ip

    def close(self):
        if self.handle:
            L = self.L
            for rb in self._buffers.values():
                L.giCDestroyRenderBuffer(rb)
            self._buffers = {}
            destroy = {"sphere": L.giCDestroySphereLight, "distant": L.giCDestroyDistantLight, "rect": L.giCDestroyRectLight,
                       "disk": L.giCDestroyDiskLight}
            for kind, h in self.lights:
                destroy[kind](self.handle, h)
            for h in self.meshes:
                L.giCDestroyMesh(h)
            for h in self.materials:
                L.giCDestroyMaterial(h)
            if self.dome:
                L.giCDestroyDomeLight(self.dome)
            for h in self.textures:
                L.giCDestroyTexture(h)
            self.lights, self.meshes, self.materials, self.textures, self.dome = [], [], [], [], None
            self.L.giCDestroyScene(self.handle)
            self.handle = None


def shade_class(material) -> int:
    """giCDebugShadeClass: the k_shade variant (shade class) of an untextured material; host only."""
    md = GiCMaterialDesc(material.klass, 0, (C.c_float * P_COUNT)(*np.asarray(material.params, np.float32)))
    return int(load_library().giCDebugShadeClass(C.byref(md)))


def bsdf_debug(material, items, device: int = 0):
    """Device-side closed-form BSDF sample/evaluate; items float32 [n,22] -> float32 [n,15] (see include/gi_c.h)."""
    L = initialize(device)
    md = GiCMaterialDesc(material.klass, 0, (C.c_float * P_COUNT)(*np.asarray(material.params, np.float32)))
    a = np.ascontiguousarray(items, np.float32).reshape(-1, 22)
    out = np.zeros((len(a), 15), np.float32)
    if L.giCDebugEvalBsdf(C.byref(md), len(a), a.ctypes.data_as(_FP), out.ctypes.data_as(_FP)) != GI_C_OK:
        raise GiError("giCDebugEvalBsdf failed: " + L.giCGetLastError().decode())
    return out
