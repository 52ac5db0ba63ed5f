"""The programmatic cornell scene must equal what the usda reader derives from the reference's cornell.usda."""
import os

import numpy as np
import pytest

from gatling_amd.meshprep import build_mesh_arrays, fan_triangulate
from gatling_amd.scenes import cornell_box
from gatling_amd.usda import load_usda, parse_usda

REF = "/root/reference/cornell.usda"


def test_cornell_facts():
    s = cornell_box()
    assert len(s.meshes) == 8 and len(s.materials) == 4 and s.triangle_count() == 46
    assert sum(len(m.vertices) for m in s.meshes) == 138  # de-indexed: 3 unique vertices per triangle
    assert all(m.double_sided for m in s.meshes)
    assert tuple(s.materials[0].params[3:6]) == (8.5, 6.0, 4.0)


@pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not present (GPU box)")
def test_cornell_matches_reference_usda():
    a, b = cornell_box(), load_usda(REF)
    assert len(a.meshes) == len(b.meshes)
    for ma, mb in zip(a.meshes, b.meshes):
        assert ma.name == mb.name and ma.material == mb.material and ma.double_sided == mb.double_sided
        assert ma.vertices.tobytes() == mb.vertices.tobytes() and np.array_equal(ma.faces, mb.faces)
        assert np.array_equal(ma.transform, mb.transform)
    for ma, mb in zip(a.materials, b.materials):
        assert ma.name == mb.name and np.array_equal(ma.params, mb.params)
    assert a.camera == b.camera


def test_fan_triangulation_order():
    tris, fvs = fan_triangulate([4, 3, 5], [0, 1, 3, 2, 4, 5, 6, 7, 8, 9, 10, 11])
    assert tris.tolist() == [[0, 1, 3], [0, 3, 2], [4, 5, 6], [7, 8, 9], [7, 9, 10], [7, 10, 11]]  # quad [0,1,3,2] -> (0,1,3),(0,3,2)
    assert fvs.tolist() == [[0, 1, 2], [0, 2, 3], [4, 5, 6], [7, 8, 9], [7, 9, 10], [7, 10, 11]]
    tl, _ = fan_triangulate([4], [0, 1, 2, 3], left_handed=True)
    assert tl.tolist() == [[0, 2, 1], [0, 3, 2]]


def test_face_varying_deindex_and_tangents():
    v, f = build_mesh_arrays([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)], [4], [0, 1, 2, 3], normals=[(0, 0, 1)] * 4,
                             normals_interpolation="faceVarying")
    assert len(v) == 6 and f.tolist() == [[0, 1, 2], [3, 4, 5]]
    t, n = v["tangent"], v["norm"]
    assert np.allclose(np.einsum("ij,ij->i", t, n), 0) and np.all(v["bitangentSign"] == 1.0)


def test_usda_parser_subset():
    root = parse_usda('''#usda 1.0
( defaultPrim = "A" upAxis = "Z" )
def Xform "A" { matrix4d xformOp:transform = ( (1,0,0,0),(0,1,0,0),(0,0,1,0),(1,2,3,1) )
  def Mesh "M" ( prepend apiSchemas = ["MaterialBindingAPI"] ) { uniform bool doubleSided = 1
    int[] faceVertexCounts = [3]  int[] faceVertexIndices = [0,1,2]  rel material:binding = </A/Mat>
    point3f[] points = [(0,0,0),(1,0,0),(0,1,-2.5e-1)] normal3f[] normals = [(0,0,1),(0,0,1),(0,0,1)] ( interpolation = "vertex" )
    uniform token info:id = "x" token outputs:surface } }''')
    mesh = root.children[0].children[0]
    assert mesh.type == "Mesh" and mesh.attrs["doubleSided"] == 1.0 and mesh.attrs["points"][2] == [0.0, 1.0, -0.25]
    assert mesh.attrs["material:binding"] == ("path", "/A/Mat") and mesh.attr_meta["normals"]["interpolation"] == "vertex"
    assert mesh.attrs["info:id"] == "x" and mesh.attrs["outputs:surface"] is None
