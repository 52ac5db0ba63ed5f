"""gatling_amd -- MI355X-native wavefront path-tracing core behind gatling's ``gi`` boundary.

The product is the C-ABI shared library built from ``gatling_amd/csrc`` (HIP kernels for gfx950 + host
scene/BVH code, declared in ``include/gi_c.h``).  The Python modules here are the harness-side mirror of the
reference's gi interface (ctypes binding, scene description, a small .usda reader, scene generators).
"""
from .scene import (CameraDesc, DiskLight, DistantLight, MaterialDesc, MeshDesc, RectLight, RenderSettings,
                    SceneDesc, SphereLight, MAT_DIFFUSE, MAT_USD_PREVIEW_SURFACE, MAT_OPEN_PBR)

__all__ = ["CameraDesc", "DiskLight", "DistantLight", "MaterialDesc", "MeshDesc", "RectLight", "RenderSettings",
           "SceneDesc", "SphereLight", "MAT_DIFFUSE", "MAT_USD_PREVIEW_SURFACE", "MAT_OPEN_PBR"]
