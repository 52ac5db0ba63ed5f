import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gatling_amd import capi
from gatling_amd.scene import MaterialDesc
from oracle import orc
rng = np.random.default_rng(5); n = 4096
nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
t = np.cross(nrm, rng.normal(size=(n, 3))); t /= np.linalg.norm(t, axis=1, keepdims=True); b = np.cross(nrm, t)
def hemi():
    v = rng.normal(size=(n, 3)); v[:, 2] = np.abs(v[:, 2]) + 0.05; v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v[:, :1] * t + v[:, 1:2] * b + v[:, 2:3] * nrm
items = np.concatenate([nrm, t, b, nrm, hemi(), hemi(), rng.uniform(size=(n, 4))], axis=1).astype(np.float32)
m = MaterialDesc.usd_preview_surface(diffuseColor=(0.7, 0.4, 0.2), roughness=0.5)
g, r = capi.bsdf_debug(m, items), orc.bsdf_debug(m, items)
names = ["k2x","k2y","k2z","opx","opy","opz","pdf","ev","dx","dy","dz","gx","gy","gz","epdf"]
for j, nm in enumerate(names):
    neq = (g[:, j].view(np.uint32) != r[:, j].view(np.uint32))
    print(nm, int(neq.sum()), float(np.abs(g[:, j] - r[:, j]).max()))
ev = r[:, 7]
for e in np.unique(ev):
    sel = ev == e
    print("event", e, int(sel.sum()), "k2 mismatches", int((g[sel, 0:3].view(np.uint32) != r[sel, 0:3].view(np.uint32)).any(axis=1).sum()))
