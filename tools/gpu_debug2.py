import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gatling_amd import capi
from gatling_amd.scene import MaterialDesc, MAT_DIFFUSE, RenderSettings
from gatling_amd.scenes import cornell_box
from oracle import orc
rng = np.random.default_rng(5); n = 4096
nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
t = np.cross(nrm, rng.normal(size=(n, 3))); t /= np.linalg.norm(t, axis=1, keepdims=True); b = np.cross(nrm, t)
def hemi():
    v = rng.normal(size=(n, 3)); v[:, 2] = np.abs(v[:, 2]) + 0.05; v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v[:, :1] * t + v[:, 1:2] * b + v[:, 2:3] * nrm
xi = rng.uniform(size=(n, 4))
for quant in (False, True):
    x = xi.copy()
    if quant: x = np.floor(x * 2**23) / 2**23
    items = np.concatenate([nrm, t, b, nrm, hemi(), hemi(), x], axis=1).astype(np.float32)
    m = MaterialDesc.usd_preview_surface(diffuseColor=(0.7, 0.4, 0.2), klass=MAT_DIFFUSE)
    g, r = capi.bsdf_debug(m, items), orc.bsdf_debug(m, items)
    neq = (g[:, :3].view(np.uint32) != r[:, :3].view(np.uint32)).any(axis=1)
    print("quant", quant, "diffuse k2 mismatches", int(neq.sum()), "pdf mismatches", int((g[:,6].view(np.uint32)!=r[:,6].view(np.uint32)).sum()))
    idx = np.nonzero(neq)[0][:3]
    for i in idx:
        print(" item", i, "xi", items[i, 18:20], "gpu", g[i, :3], "ref", r[i, :3], "pdf", g[i,6], r[i,6])
# clip case
desc = cornell_box(MAT_DIFFUSE); desc.camera.clip_start, desc.camera.clip_end = 6.5, 7.8
rs = RenderSettings(spp=4, max_bounces=6, clipping_planes=True)
sc = capi.Scene(desc); img = sc.render(rs, 96, 54); st = sc.stats(); sc.close()
ref, cnt = orc.render(desc, rs, 96, 54)
d = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
print("clip: seg", st["segments"], cnt["segments"], "diff pixels", np.argwhere(d)[:10].tolist(), [ (img[y,x].tolist(), ref[y,x].tolist()) for y,x in np.argwhere(d)[:4]])
for spp in (1,):
    rs = RenderSettings(spp=1, max_bounces=1, clipping_planes=True)
    sc = capi.Scene(desc); img = sc.render(rs, 96, 54); st = sc.stats(); sc.close()
    ref, cnt = orc.render(desc, rs, 96, 54)
    d = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    print("clip 1spp 1b: diff pixels", np.argwhere(d)[:10].tolist())
