"""Host-side geometry derivation that defines the *inputs* of the gi boundary.

In the reference this work is done by the Hydra delegate (which stays as-is) before ``giCreateMesh``:
``/root/reference/src/hdGatling/mesh.cpp`` -- fan triangulation through ``HdMeshUtil``
(:848), de-indexing when any primvar is faceVarying (:634, 916-921), smooth fallback normals (:895-907),
Duff fallback tangents when there are no texcoords (:231-261) and the bitangent sign (:83-86).  The test /
bench harness has no OpenUSD, so this module restates those steps for the meshes it builds.
"""
from __future__ import annotations

import numpy as np

from .scene import VERTEX_DTYPE


def fan_triangulate(face_vertex_counts, face_vertex_indices, left_handed=False):
    """Returns (tri_point_indices [T,3], tri_fv_indices [T,3]) -- HdMeshUtil::ComputeTriangleIndices order.

    Face [a,b,c,d] -> (a,b,c), (a,c,d).  For left-handed orientation Hydra flips the winding.
    ``tri_fv_indices`` index the flat face-vertex array (for faceVarying primvars).
    """
    counts = np.asarray(face_vertex_counts, np.int64)
    idx = np.asarray(face_vertex_indices, np.int64)
    tris, fvs = [], []
    base = 0
    for n in counts:
        for k in range(1, n - 1):
            a, b, c = base, base + k, base + k + 1
            if left_handed:
                b, c = c, b
            fvs.append((a, b, c))
            tris.append((idx[a], idx[b], idx[c]))
        base += n
    return (np.asarray(tris, np.uint32).reshape(-1, 3), np.asarray(fvs, np.int64).reshape(-1, 3))


def smooth_normals(points, tris):
    """Hd_SmoothNormals: per-vertex sum of (unnormalised, i.e. area-weighted) face normals, normalised."""
    p = np.asarray(points, np.float32)
    t = np.asarray(tris, np.int64)
    fn = np.cross(p[t[:, 1]] - p[t[:, 0]], p[t[:, 2]] - p[t[:, 0]]).astype(np.float32)
    n = np.zeros_like(p)
    for k in range(3):
        np.add.at(n, t[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    return (n / ln).astype(np.float32)


def duff_basis(n):
    """Duff et al. orthonormal basis, vectorised; mesh.cpp:231-239 (== common.glsl:128-137)."""
    n = np.asarray(n, np.float32)
    one = np.float32(1.0)
    s = np.where(n[:, 2] >= 0, one, -one).astype(np.float32)
    a = (-one / (s + n[:, 2])).astype(np.float32)
    b = (n[:, 0] * n[:, 1] * a).astype(np.float32)
    t = np.stack([one + s * n[:, 0] * n[:, 0] * a, s * b, -s * n[:, 0]], axis=1).astype(np.float32)
    bt = np.stack([b, s + n[:, 1] * n[:, 1] * a, -n[:, 1]], axis=1).astype(np.float32)
    return t, bt


def bake_vertices(points, normals, texcoords=None):
    """_BakeMeshGeometry (mesh.cpp:281-331) with the fallback-tangent path (no authored tangents)."""
    p = np.asarray(points, np.float32).reshape(-1, 3)
    n = np.asarray(normals, np.float32).reshape(-1, 3)
    assert len(p) == len(n)
    if texcoords is None:
        t, bt = duff_basis(n)
        uv = np.zeros((len(p), 2), np.float32)
    else:
        # Texture-space tangents (Lengyel, mesh.cpp:93-223) are not needed by the harness scenes yet.
        t, bt = duff_basis(n)
        uv = np.asarray(texcoords, np.float32).reshape(-1, 2)
    sign = np.where(np.einsum("ij,ij->i", np.cross(t, bt), n) > 0, 1.0, -1.0).astype(np.float32)
    v = np.zeros(len(p), VERTEX_DTYPE)
    v["pos"] = p
    v["norm"] = n
    v["tangent"] = t
    v["bitangentSign"] = sign
    v["u"] = uv[:, 0]
    v["v"] = uv[:, 1]
    return v


def build_mesh_arrays(points, face_vertex_counts, face_vertex_indices, normals=None,
                      normals_interpolation="vertex", left_handed=False, texcoords=None):
    """Returns (vertices VERTEX_DTYPE[N], faces uint32[T,3]) the way hdGatling would hand them to giCreateMesh."""
    points = np.asarray(points, np.float32).reshape(-1, 3)
    tris, fvs = fan_triangulate(face_vertex_counts, face_vertex_indices, left_handed)
    if normals is not None and normals_interpolation == "faceVarying":
        # de-index everything: three unique vertices per triangle (mesh.cpp:916-921, 1011-1017)
        normals = np.asarray(normals, np.float32).reshape(-1, 3)
        pts = points[tris.reshape(-1).astype(np.int64)]
        nrm = normals[fvs.reshape(-1)]
        faces = np.arange(len(pts), dtype=np.uint32).reshape(-1, 3)
        uv = np.asarray(texcoords, np.float32).reshape(-1, 2)[tris.reshape(-1).astype(np.int64)] if texcoords is not None else None
        return bake_vertices(pts, nrm, uv), faces
    if normals is None:
        normals = smooth_normals(points, tris)
    return bake_vertices(points, np.asarray(normals, np.float32).reshape(-1, 3), texcoords), tris.astype(np.uint32)
