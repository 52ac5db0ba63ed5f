"""Neutral scene description shared by the product binding (capi.py) and the test oracle binding.

The fields mirror the *inputs* of the reference's gi API (``/root/reference/src/gi/gtl/gi/Gi.h``):
``GiVertex`` (:110-118), ``GiMeshDesc`` (:124-137), ``GiCameraDesc`` (:96-108), ``GiRenderSettings``
(:139-159) and the light setters (:226-256).  Defaults of :class:`RenderSettings` are the Hydra render
setting defaults of ``src/hdGatling/renderDelegate.cpp:93-110``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

# GiVertex: pos[3], u, norm[3], v, tangent[3], bitangentSign  (48 bytes)
VERTEX_DTYPE = np.dtype([
    ("pos", np.float32, 3), ("u", np.float32),
    ("norm", np.float32, 3), ("v", np.float32),
    ("tangent", np.float32, 3), ("bitangentSign", np.float32),
])
assert VERTEX_DTYPE.itemsize == 48

# material classes (closed-form BSDFs, see DESIGN.md "Materials")
MAT_DIFFUSE = 0
MAT_USD_PREVIEW_SURFACE = 1
MAT_OPEN_PBR = 2

# parameter block indices (float p[64]); must match include/gi_c.h and oracle/gi_oracle.h
P_BASE_COLOR = 0
P_EMISSION = 3
P_USE_SPECULAR_WORKFLOW = 6
P_SPECULAR_COLOR = 7
P_METALLIC = 10
P_ROUGHNESS = 11
P_CLEARCOAT = 12
P_CLEARCOAT_ROUGHNESS = 13
P_OPACITY = 14
P_OPACITY_THRESHOLD = 15
P_IOR = 16
P_BASE_WEIGHT = 17
P_SPECULAR_WEIGHT = 18
P_COAT_COLOR = 19
P_COAT_IOR = 22
P_TRANSMISSION_WEIGHT = 23
P_TRANSMISSION_COLOR = 24
P_DIFFUSE_ROUGHNESS = 27
P_TRANSMISSION_DEPTH = 28
P_TRANSMISSION_SCATTER = 29           # 3
P_TRANSMISSION_SCATTER_ANISOTROPY = 47  # (32..46 hold the device's derived constants)
P_COAT_DARKENING = 48
P_FUZZ_WEIGHT = 49
P_FUZZ_COLOR = 50                    # 3
P_FUZZ_ROUGHNESS = 53
P_THIN_WALLED = 54
P_SUBSURFACE_WEIGHT = 55
P_SUBSURFACE_COLOR = 56              # 3
P_SUBSURFACE_ANISOTROPY = 59
P_SUBSURFACE_RADIUS = 32             # (slots 32..35 of the USER block; on the device these indices hold derived constants, written after the inputs were read)
P_SUBSURFACE_RADIUS_SCALE = 33       # 3
P_SPECULAR_ANISOTROPY = 60
P_COAT_ANISOTROPY = 61
P_COAT_ROTATION = 36                 # geometry_coat_tangent as a turn of the tangent, in turns (USER block only, like 32..35)
P_SPECULAR_ROTATION = 37             # geometry_tangent likewise: the tangent of the dielectric / conductor lobes
P_THIN_FILM_WEIGHT = 62
P_THIN_FILM_THICKNESS = 63           # micrometres
P_THIN_FILM_IOR = 6                  # OpenPBR class only (the slot is useSpecularWorkflow for UsdPreviewSurface)
P_COUNT = 64


@dataclass
class TextureBinding:
    """One textured material input, UsdUVTexture semantics: value = texel * scale + bias, sampled at the hit's st."""
    texture: int = -1                      # index into SceneDesc.textures
    wrap_s: int = 1                        # TEX_WRAP_* (mdl_types.glsl:117-120): 0 clamp, 1 repeat, 2 mirrored repeat, 3 clip
    wrap_t: int = 1
    channel: int = 0                       # scalar inputs (roughness, metallic): channel of the scaled/biased texel
    scale: tuple = (1.0, 1.0, 1.0, 1.0)
    bias: tuple = (0.0, 0.0, 0.0, 0.0)
    transform: Optional[tuple] = None      # UsdTransform2d upstream of `st`, folded: s' = (xf0 s + xf1 t) + xf2, t' = (xf3 s + xf4 t) + xf5 (see usd_transform_2d)


def usd_transform_2d(rotation_deg=0.0, scale=(1.0, 1.0), translation=(0.0, 0.0)):
    """The six floats of a UsdTransform2d node (UsdPreviewSurface specification: result = in * scale, rotated counter-clockwise by `rotation` degrees about the
    origin, + translation).  cos / sin in double precision, rounded once to float32 -- the gtl shim's MaterialX reader (gtl_shim.cpp) does the same."""
    import math
    rad = float(np.float32(rotation_deg)) * 3.14159265358979323846 / 180.0   # (the inputs are float32 values of a document: converted first, then double arithmetic)
    c, s = math.cos(rad), math.sin(rad)
    sx, sy = float(np.float32(scale[0])), float(np.float32(scale[1]))
    return tuple(float(np.float32(x)) for x in (c * sx, -s * sy, float(translation[0]), s * sx, c * sy, float(translation[1])))


TEX_WRAP_CLAMP, TEX_WRAP_REPEAT, TEX_WRAP_MIRRORED_REPEAT, TEX_WRAP_CLIP = 0, 1, 2, 3
TEX_TRANSMISSION_WEIGHT, TEX_TRANSMISSION_COLOR = 7, 8  # OpenPBR transmission_weight (scalar) / transmission_color (rgb; tints the surface when transmission_depth is 0)
TEX_BASE_COLOR, TEX_EMISSION, TEX_ROUGHNESS, TEX_METALLIC, TEX_NORMAL, TEX_OPACITY, TEX_COAT_NORMAL, TEX_SLOT_COUNT = 0, 1, 2, 3, 4, 5, 6, 9  # TEX_COAT_NORMAL: OpenPBR geometry_coat_normal


@dataclass
class MaterialDesc:
    name: str = "material"
    klass: int = MAT_USD_PREVIEW_SURFACE
    params: np.ndarray = field(default_factory=lambda: np.zeros(P_COUNT, np.float32))
    textures: dict = field(default_factory=dict)   # TEX_* slot -> TextureBinding
    primvar_inputs: dict = field(default_factory=dict)   # TEX_* slot -> primvar name (UsdPrimvarReader; scene_data_lookup_*)

    @staticmethod
    def usd_preview_surface(name="mat", diffuseColor=(0.18, 0.18, 0.18), emissiveColor=(0, 0, 0),
                            useSpecularWorkflow=0, specularColor=(0, 0, 0), metallic=0.0, roughness=0.5,
                            clearcoat=0.0, clearcoatRoughness=0.01, opacity=1.0, opacityThreshold=0.0,
                            ior=1.5, klass=MAT_USD_PREVIEW_SURFACE) -> "MaterialDesc":
        """UsdPreviewSurface inputs with the spec's fallback values."""
        p = np.zeros(P_COUNT, np.float32)
        p[P_BASE_COLOR:P_BASE_COLOR + 3] = diffuseColor
        p[P_EMISSION:P_EMISSION + 3] = emissiveColor
        p[P_USE_SPECULAR_WORKFLOW] = useSpecularWorkflow
        p[P_SPECULAR_COLOR:P_SPECULAR_COLOR + 3] = specularColor
        p[P_METALLIC] = metallic
        p[P_ROUGHNESS] = roughness
        p[P_CLEARCOAT] = clearcoat
        p[P_CLEARCOAT_ROUGHNESS] = clearcoatRoughness
        p[P_OPACITY] = opacity
        p[P_OPACITY_THRESHOLD] = opacityThreshold
        p[P_IOR] = ior
        return MaterialDesc(name=name, klass=klass, params=p)


def _open_pbr(name="mat", base_weight=1.0, base_color=(0.8, 0.8, 0.8), base_metalness=0.0, specular_weight=1.0,
              specular_color=(1, 1, 1), specular_roughness=0.3, specular_ior=1.5, transmission_weight=0.0,
              transmission_color=(1, 1, 1), transmission_depth=0.0, coat_weight=0.0, coat_color=(1, 1, 1), coat_roughness=0.0,
              coat_ior=1.6, emission_luminance=0.0, emission_color=(1, 1, 1), base_diffuse_roughness=0.0,
              transmission_scatter=(0, 0, 0), transmission_scatter_anisotropy=0.0, coat_darkening=1.0, fuzz_weight=0.0,
              fuzz_color=(1, 1, 1), fuzz_roughness=0.5, geometry_thin_walled=False, subsurface_weight=0.0, subsurface_color=(0.8, 0.8, 0.8),
              subsurface_scatter_anisotropy=0.0, specular_roughness_anisotropy=0.0, coat_roughness_anisotropy=0.0,
              thin_film_weight=0.0, thin_film_thickness=0.5, thin_film_ior=1.4, subsurface_radius=1.0, subsurface_radius_scale=(1.0, 0.5, 0.25),
              geometry_opacity=1.0, coat_rotation=0.0, specular_rotation=0.0) -> MaterialDesc:
    """open_pbr_surface inputs with the defaults of src/gi/mtlx/open_pbr_surface.mtlx:11-92 (the lobes this core implements)."""
    p = np.zeros(P_COUNT, np.float32)
    p[P_BASE_COLOR:P_BASE_COLOR + 3] = base_color
    p[P_EMISSION:P_EMISSION + 3] = np.float32(emission_luminance) * np.asarray(emission_color, np.float32)
    p[P_SPECULAR_COLOR:P_SPECULAR_COLOR + 3] = specular_color
    p[P_METALLIC] = base_metalness
    p[P_ROUGHNESS] = specular_roughness
    p[P_CLEARCOAT] = coat_weight
    p[P_CLEARCOAT_ROUGHNESS] = coat_roughness
    p[P_OPACITY] = geometry_opacity
    p[P_IOR] = specular_ior
    p[P_BASE_WEIGHT] = base_weight
    p[P_SPECULAR_WEIGHT] = specular_weight
    p[P_COAT_COLOR:P_COAT_COLOR + 3] = coat_color
    p[P_COAT_IOR] = coat_ior
    p[P_TRANSMISSION_WEIGHT] = transmission_weight
    p[P_TRANSMISSION_COLOR:P_TRANSMISSION_COLOR + 3] = transmission_color
    p[P_DIFFUSE_ROUGHNESS] = base_diffuse_roughness
    p[P_TRANSMISSION_DEPTH] = transmission_depth
    p[P_TRANSMISSION_SCATTER:P_TRANSMISSION_SCATTER + 3] = transmission_scatter
    p[P_TRANSMISSION_SCATTER_ANISOTROPY] = transmission_scatter_anisotropy
    p[P_COAT_DARKENING] = coat_darkening
    p[P_FUZZ_WEIGHT] = fuzz_weight                     # fuzz (sheen) layer over the coat (open_pbr_surface.mtlx:569-581)
    p[P_FUZZ_COLOR:P_FUZZ_COLOR + 3] = fuzz_color
    p[P_FUZZ_ROUGHNESS] = fuzz_roughness
    p[P_THIN_WALLED] = 1.0 if geometry_thin_walled else 0.0
    p[P_SUBSURFACE_WEIGHT] = subsurface_weight         # thin-walled form (open_pbr_surface.mtlx:140-196) always; volumetric form (:182-192) in renders with a medium stack
    p[P_SUBSURFACE_RADIUS] = subsurface_radius         # mean free path = radius * radius_scale per channel (:47-50, 182-186)
    p[P_SUBSURFACE_RADIUS_SCALE:P_SUBSURFACE_RADIUS_SCALE + 3] = subsurface_radius_scale
    p[P_SUBSURFACE_COLOR:P_SUBSURFACE_COLOR + 3] = subsurface_color
    p[P_SUBSURFACE_ANISOTROPY] = subsurface_scatter_anisotropy
    p[P_SPECULAR_ANISOTROPY] = specular_roughness_anisotropy   # open_pbr_anisotropy (open_pbr_surface.mtlx:133-136, 552-555): highlights stretched along the tangent
    p[P_COAT_ANISOTROPY] = coat_roughness_anisotropy
    p[P_COAT_ROTATION] = coat_rotation                 # geometry_coat_tangent (:91, 561) = rotate3d(Tworld, 360 * coat_rotation degrees, N)
    p[P_SPECULAR_ROTATION] = specular_rotation         # geometry_tangent (:89) likewise
    p[P_THIN_FILM_WEIGHT] = thin_film_weight           # thin film on the dielectric and metal lobes (open_pbr_surface.mtlx:300-304, 404-431, 450-464)
    p[P_THIN_FILM_THICKNESS] = thin_film_thickness
    p[P_THIN_FILM_IOR] = thin_film_ior
    return MaterialDesc(name=name, klass=MAT_OPEN_PBR, params=p)


MaterialDesc.open_pbr = staticmethod(_open_pbr)


PRIMVAR_FLOAT, PRIMVAR_VEC2, PRIMVAR_VEC3, PRIMVAR_VEC4 = 0, 1, 2, 3                       # GiPrimvarType (Gi.h:76-79)
PRIMVAR_INT, PRIMVAR_INT2, PRIMVAR_INT3, PRIMVAR_INT4 = 4, 5, 6, 7
INTERP_CONSTANT, INTERP_INSTANCE, INTERP_UNIFORM, INTERP_VERTEX = 0, 1, 2, 3                # GiPrimvarInterpolation (Gi.h:81-84)


@dataclass
class Primvar:
    """GiPrimvarData (Gi.h:86-92): float data, [n] or [n, components]."""
    name: str
    type: int
    interpolation: int
    data: np.ndarray


@dataclass
class MeshDesc:
    name: str
    vertices: np.ndarray            # VERTEX_DTYPE[N]
    faces: np.ndarray               # uint32 [M,3]
    material: int = 0               # index into SceneDesc.materials
    id: int = 0
    double_sided: bool = False
    left_handed: bool = False
    visible: bool = True
    transform: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))  # USD row-vector convention
    instance_transforms: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32)[None].copy())
    instance_ids: Optional[np.ndarray] = None
    face_ids: Optional[np.ndarray] = None   # int32 per face (GiMeshDesc.faceIds); None = zeros
    max_face_id: int = 0
    primvars: list = field(default_factory=list)            # GiMeshDesc.primvars
    instancer_primvars: list = field(default_factory=list)  # giSetMeshInstancerPrimvars


@dataclass
class SphereLight:
    pos: tuple = (0, 0, 0)
    base_emission: tuple = (0, 0, 0)
    radius: tuple = (0.5, 0.5, 0.5)
    diffuse: float = 1.0
    specular: float = 1.0


@dataclass
class DistantLight:
    direction: tuple = (0, 0, 0)
    base_emission: tuple = (0, 0, 0)
    angle: float = 0.0
    diffuse: float = 1.0
    specular: float = 1.0


@dataclass
class RectLight:
    origin: tuple = (0, 0, 0)
    t0: tuple = (1, 0, 0)
    t1: tuple = (0, 1, 0)
    base_emission: tuple = (0, 0, 0)
    width: float = 1.0
    height: float = 1.0
    diffuse: float = 1.0
    specular: float = 1.0


@dataclass
class DiskLight:
    origin: tuple = (0, 0, 0)
    t0: tuple = (1, 0, 0)
    t1: tuple = (0, 1, 0)
    base_emission: tuple = (0, 0, 0)
    radius_x: float = 0.5
    radius_y: float = 0.5
    diffuse: float = 1.0
    specular: float = 1.0


@dataclass
class CameraDesc:
    position: tuple = (0, 0, 0)
    forward: tuple = (0, 0, -1)
    up: tuple = (0, 1, 0)
    vfov: float = 0.8
    f_stop: float = 0.0
    focus_distance: float = 0.0
    focal_length: float = 5.0
    clip_start: float = 0.1
    clip_end: float = 100.0
    exposure: float = 0.0


@dataclass
class RenderSettings:
    """Defaults: src/hdGatling/renderDelegate.cpp:93-110; colour clear value :229."""
    spp: int = 1
    max_bounces: int = 13
    rr_bounce_offset: int = 3
    rr_inv_min_term_prob: float = 0.95
    max_sample_value: float = 10.0
    filter_importance_sampling: bool = True
    depth_of_field: bool = False
    light_intensity_multiplier: float = 1.0
    next_event_estimation: bool = False
    clipping_planes: bool = False
    medium_stack_size: int = 0
    frame: float = 0.0              # GiRenderSettings.frame (Gi.h:144): the FRAME scene data
    max_volume_walk_length: int = 7
    jittered_sampling: bool = True
    meters_per_scene_unit: float = 1.0
    progressive_accumulation: bool = True
    dome_light_camera_visible: bool = True
    clear_color: tuple = (1.0, 1.0, 1.0, 1.0)


@dataclass
class DomeLight:
    """Gi.cpp:2943-2976: equirectangular environment texture, rotation quaternion (x, y, z, w), emission multiplier."""
    texture: int = -1                      # index into SceneDesc.textures (the reference loads it from a file path)
    rotation: tuple = (0.0, 0.0, 0.0, 1.0)
    base_emission: tuple = (1.0, 1.0, 1.0)
    diffuse: float = 1.0
    specular: float = 1.0


@dataclass
class SceneDesc:
    meshes: List[MeshDesc] = field(default_factory=list)
    materials: List[MaterialDesc] = field(default_factory=list)
    sphere_lights: List[SphereLight] = field(default_factory=list)
    distant_lights: List[DistantLight] = field(default_factory=list)
    rect_lights: List[RectLight] = field(default_factory=list)
    disk_lights: List[DiskLight] = field(default_factory=list)
    textures: List[np.ndarray] = field(default_factory=list)   # float32 [h, w, 4], linear, row 0 first (v = 0 side)
    dome_light: Optional[DomeLight] = None
    camera: CameraDesc = field(default_factory=CameraDesc)

    def triangle_count(self) -> int:
        return int(sum(len(m.faces) * len(m.instance_transforms) for m in self.meshes if m.visible))
