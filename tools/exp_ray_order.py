"""Experiment: how much does ray ORDER change k_trace's time on the LDS-resident cornell scene?

  rocprofv3 --kernel-trace -d out -o t --output-format csv -- python tools/exp_ray_order.py     (k_trace dispatch durations, in call order)

Measured on MI355X (r01): 3.74 M secondary rays (bounces 1-3 mixed): random order 341 us, sorted by octant 331, by origin triangle
308, by (triangle, octant) 290, by (4x4x4 origin cell, octant) 298 -- at most -15 %, less than a sorting pass over the ray records
costs (>= 60 us for 4 M rays at 5 TB/s); camera rays cost 37 ps/ray, secondary rays 77-91 ps/ray whatever their order.
`... exp_ray_order.py soup 1000000` (k_trace_dyn, scene beyond LDS): random 2061 us, best order (16^3 Morton cell, octant) 1848 us."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatling_amd import capi
from gatling_amd.scenes import cornell_box

if len(sys.argv) > 1 and sys.argv[1] == "soup":   # a scene beyond LDS: k_trace_dyn, incoherent node/triangle fetches
    from gatling_amd.scenes import random_triangle_soup
    desc = random_triangle_soup(int(sys.argv[2]) if len(sys.argv) > 2 else 1000000, seed=1234)
else:
    desc = cornell_box()
sc = capi.Scene(desc)
rng = np.random.default_rng(1)
N = 4 << 20
# world-space triangles in device order: meshes in order, one instance each, faces in order
tris = []
for m in desc.meshes:
    M = np.asarray(m.transform, np.float64) @ np.asarray(m.instance_transforms, np.float64).reshape(-1, 4, 4)[0]
    P = np.c_[m.vertices["pos"].astype(np.float64), np.ones(len(m.vertices))] @ M
    tris.append(P[:, :3][np.asarray(m.faces, np.int64)])
first = np.cumsum([0] + [len(t) for t in tris])
T = np.concatenate(tris)
NRM = np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]); NRM /= np.linalg.norm(NRM, axis=1, keepdims=True)
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
    tri = first[np.clip(ip[:, 0], 0, None)] + np.clip(ip[:, 1], 0, None)
    n = NRM[tri]
    n = np.where((np.einsum("ij,ij->i", n, d) > 0)[:, None], -n, n)
    p = o + d * tuv[:, :1] + n * 1e-4
    # cosine-weighted direction around n
    u1, u2 = rng.random(len(o)), rng.random(len(o))
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    a = np.where(np.abs(n[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    t = np.cross(a, n); t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(n, t)
    nd = t * (r * np.cos(phi))[:, None] + b * (r * np.sin(phi))[:, None] + n * np.sqrt(1 - u1)[:, None]
    print(label, "hit fraction", hit.mean(), flush=True)
    return p[hit], nd[hit], tri[hit]

o1, d1, t1 = bounce(o, d, "camera rays (sorted by pixel)")
o2, d2, t2 = bounce(o1, d1, "bounce-1 rays (pixel order)")
o3, d3, t3 = bounce(o2, d2, "bounce-2 rays (pixel order)")
S_o = np.concatenate([o1, o2, o3])[:N]; S_d = np.concatenate([d1, d2, d3])[:N]; S_t = np.concatenate([t1, t2, t3])[:N]
perm = rng.permutation(len(S_o))
S_o, S_d, S_t = S_o[perm], S_d[perm], S_t[perm]
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
          "by 16^3 morton cell, octant": np.lexsort((octant, morton16)), "by octant, 16^3 morton": np.lexsort((morton16, octant))}
for name, idx in orders.items():
    sc.trace_rays(S_o[idx].astype(np.float32), S_d[idx].astype(np.float32))
    print("order:", name, len(idx), flush=True)
sc.close()
