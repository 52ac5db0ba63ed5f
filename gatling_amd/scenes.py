"""Scene builders for the BASELINE.json configs (SURVEY.md section 8d "Concrete inputs").

``cornell_box()`` restates the *facts* of ``/root/reference/cornell.usda`` (file lines
cited below) programmatically so that tests/bench never read ``/root/reference`` at run time (it does not exist
on the GPU box).  ``tests/test_scene_cornell.py`` checks it field-by-field against the .usda through
:mod:`gatling_amd.usda` whenever the reference checkout is present.
"""
from __future__ import annotations

import numpy as np

from .meshprep import build_mesh_arrays
from .scene import (CameraDesc, MaterialDesc, MeshDesc, RectLight, SceneDesc, MAT_DIFFUSE, MAT_OPEN_PBR,
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
