"""Experiment: how much does ray ORDER change the traversal kernels' time?

  rocprofv3 --kernel-trace -d out -o t --output-format csv -- python tools/exp_ray_order.py [cornell | soup N | instances | interior]
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d out2 -o t -- python tools/exp_ray_order.py <same>       (L2 hit rate per order)
  python tools/exp_ray_order_report.py <stdout of the first run> out/.../t_kernel_trace.csv [out2/.../t_results.db]

(k_trace / k_trace_dyn dispatch durations, in call order: the first three dispatches are the bounces that generate the rays, then one per order.)

Measured on MI355X (r01): 3.74 M secondary rays (bounces 1-3 mixed): random order 341 us, sorted by octant 331, by origin triangle
308, by (triangle, octant) 290, by (4x4x4 origin cell, octant) 298 -- at most -15 %, less than a sorting pass over the ray records
costs (>= 60 us for 4 M rays at 5 TB/s); camera rays cost 37 ps/ray, secondary rays 77-91 ps/ray whatever their order.
`... exp_ray_order.py soup 1000000` (k_trace_dyn, scene beyond LDS): random 2061 us, best order (16^3 Morton cell, octant) 1848 us.
r03: `instances` = C4 (1 024 instanced icospheres, 376 MB flat BVH: L2 hit 0.45) and `interior` = C5 (10.24 M triangles, beyond the Infinity Cache): profiles/r03*_ray_order.txt."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatling_amd import capi
from gatling_amd.scenes import cornell_box

mode = sys.argv[1] if len(sys.argv) > 1 else "cornell"
if mode == "soup":   # a scene beyond LDS: k_trace_dyn, incoherent node/triangle fetches
    from gatling_amd.scenes import random_triangle_soup
    desc = random_triangle_soup(int(sys.argv[2]) if len(sys.argv) > 2 else 1000000, seed=1234)
elif mode == "instances":
    from gatling_amd.scenes import sphere_grid
    desc = sphere_grid(32, 4, 32)
elif mode == "interior":
    from gatling_amd.scenes import interior_scene
    desc = interior_scene()
else:
    desc = cornell_box()
sc = capi.Scene(desc)
rng = np.random.default_rng(1)
N = 4 << 20
# per flattened instance (meshes in order, instances in order -- the order giCTraceRays numbers them): mesh index + object-to-world matrix
inst_mesh, inst_M = [], []
for mi, m in enumerate(desc.meshes):
    its = np.asarray(m.instance_transforms, np.float64).reshape(-1, 4, 4)
    for I in its:
        inst_mesh.append(mi); inst_M.append(np.asarray(m.transform, np.float64).reshape(4, 4) @ I)
inst_mesh = np.asarray(inst_mesh); inst_M = np.asarray(inst_M)
mesh_P = [m.vertices["pos"].astype(np.float64) for m in desc.meshes]
mesh_F = [np.asarray(m.faces, np.int64) for m in desc.meshes]
mesh_nf = np.asarray([len(f) for f in mesh_F]); inst_first = np.concatenate([[0], np.cumsum(mesh_nf[inst_mesh])])


def hit_normals(inst, prim):
    """Geometric normals of the hit triangles (row-vector convention: world = local @ M)."""
    n = np.zeros((len(inst), 3))
    for mi in np.unique(inst_mesh[inst]):
        sel = np.nonzero(inst_mesh[inst] == mi)[0]
        f = mesh_F[mi][prim[sel]]
        M = inst_M[inst[sel]][:, :3, :3]
        e1 = np.einsum("ij,ijk->ik", mesh_P[mi][f[:, 1]] - mesh_P[mi][f[:, 0]], M)
        e2 = np.einsum("ij,ijk->ik", mesh_P[mi][f[:, 2]] - mesh_P[mi][f[:, 0]], M)
        c = np.cross(e1, e2)
        n[sel] = c / np.maximum(np.linalg.norm(c, axis=1, keepdims=True), 1e-30)
    return n


cam = desc.camera
fwd = np.asarray(cam.forward, np.float64); up = np.asarray(cam.up, np.float64); right = np.cross(fwd, up)
w, h = 1920, 1080
pix = np.arange(N) % (w * h)
px, py = pix % w, pix // w
th = np.tan(cam.vfov / 2)
x = ((px + rng.random(N)) / w * 2 - 1) * th * w / h
y = ((py + rng.random(N)) / h * 2 - 1) * th
d = fwd[None] + x[:, None] * right[None] + y[:, None] * up[None]
d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.broadcast_to(np.asarray(cam.position, np.float64), d.shape).copy()

def bounce(o, d, label):
    tuv, ip = sc.trace_rays(o.astype(np.float32), d.astype(np.float32))
    hit = ip[:, 0] >= 0
    inst, prim = np.clip(ip[:, 0], 0, None), np.clip(ip[:, 1], 0, None)
    tri = inst_first[inst] + prim
    n = hit_normals(inst, prim)
    n = np.where((np.einsum("ij,ij->i", n, d) > 0)[:, None], -n, n)
    p = o + d * tuv[:, :1] + n * (1e-4 * max(1.0, float(np.abs(o).max())))
    # cosine-weighted direction around n
    u1, u2 = rng.random(len(o)), rng.random(len(o))
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    a = np.where(np.abs(n[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    t = np.cross(a, n); t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(n, t)
    nd = t * (r * np.cos(phi))[:, None] + b * (r * np.sin(phi))[:, None] + n * np.sqrt(1 - u1)[:, None]
    print(label, "hit fraction", hit.mean(), flush=True)
    return p[hit], nd[hit], tri[hit], inst[hit]

o1, d1, t1, i1 = bounce(o, d, "camera rays (sorted by pixel)")
o2, d2, t2, i2 = bounce(o1, d1, "bounce-1 rays (pixel order)")
o3, d3, t3, i3 = bounce(o2, d2, "bounce-2 rays (pixel order)")
S_o = np.concatenate([o1, o2, o3])[:N]; S_d = np.concatenate([d1, d2, d3])[:N]; S_t = np.concatenate([t1, t2, t3])[:N]; S_i = np.concatenate([i1, i2, i3])[:N]
perm = rng.permutation(len(S_o))
S_o, S_d, S_t, S_i = S_o[perm], S_d[perm], S_t[perm], S_i[perm]
octant = (S_d[:, 0] >= 0) * 1 + (S_d[:, 1] >= 0) * 2 + (S_d[:, 2] >= 0) * 4
lo, hi = S_o.min(0), S_o.max(0)
cell = np.clip(((S_o - lo) / (hi - lo + 1e-9) * 4).astype(np.int64), 0, 3)
morton = cell[:, 0] * 16 + cell[:, 1] * 4 + cell[:, 2]
cell16 = np.clip(((S_o - lo) / (hi - lo + 1e-9) * 16).astype(np.int64), 0, 15)
def part1by2(v):
    v = (v | (v << 8)) & 0x00f00f; v = (v | (v << 4)) & 0x0c30c3; v = (v | (v << 2)) & 0x249249
    return v
morton16 = part1by2(cell16[:, 0]) | (part1by2(cell16[:, 1]) << 1) | (part1by2(cell16[:, 2]) << 2)
cell2 = np.clip(((S_o - lo) / (hi - lo + 1e-9) * 2).astype(np.int64), 0, 1)
key2 = cell2[:, 0] * 4 + cell2[:, 1] * 2 + cell2[:, 2]
orders = {"random mix": np.arange(len(S_o)), "by 2x2x2 cell (8 keys)": np.argsort(key2, kind="stable"), "by 2x2x2 cell, octant (64 keys)": np.lexsort((octant, key2)),
          "by 4x4x4 cell (64 keys)": np.argsort(morton, kind="stable"), "by octant": np.argsort(octant, kind="stable"), "by origin triangle": np.argsort(S_t, kind="stable"),
          "by triangle, octant": np.lexsort((octant, S_t)), "by 4x4x4 cell, octant": np.lexsort((octant, morton)), "by octant, cell": np.lexsort((morton, octant)),
          "by 16^3 morton cell, octant": np.lexsort((octant, morton16)), "by octant, 16^3 morton": np.lexsort((morton16, octant)),
          "by origin instance": np.argsort(S_i, kind="stable"), "by origin instance, octant": np.lexsort((octant, S_i))}
if mode in ("instances", "interior"):  # the orders a zero-copy bucketing could produce (8 / 64 keys) + the full sorts as upper bounds
    keep = ("random mix", "by 2x2x2 cell (8 keys)", "by octant", "by 2x2x2 cell, octant (64 keys)", "by 4x4x4 cell, octant", "by 16^3 morton cell, octant", "by origin instance, octant")
    orders = {k: v for k, v in orders.items() if k in keep}
for name, idx in orders.items():
    sc.trace_rays(S_o[idx].astype(np.float32), S_d[idx].astype(np.float32))
    print("order:", name, len(idx), flush=True)
sc.close()
