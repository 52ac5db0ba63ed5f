"""Differential campaign: random renders (tests/fuzz_scenes.py) through the C ABI on the GPU and through the oracle, compared bit for bit (test infrastructure).

  python tests/fuzz_parity.py 0:500 [--threads 64] [--log gpurun_out/fuzz.log]      renders: seeds first:last (exclusive) or a comma list
  python tests/fuzz_parity.py 129,219 --reduce                                      shrink differing cases to what still differs and print what is left
  python tests/fuzz_parity.py 0:20000 --bsdf                                        one random material on 2 048 random frames / directions per seed
  python tests/fuzz_parity.py 0:4000 --concurrent 4                                 four host threads render scenes of their own at the same time
  python tests/fuzz_parity.py 0:1500 --scale 12                                     the same cases on images 12 times as wide and as high
  GATLING_DEVICES=0,0,0 python tests/fuzz_parity.py 0:4000                          every whole-frame render dealt to three device contexts

Per render case: the colour image of one giCRender call == the oracle's, the frame's segment / shadow-ray / sample counts, and as the seed decides a second
(progressive) call, the 13 (whole frames: 16) non-colour AOVs, a row range or an interleaved row share, an edit of the live scene followed by a third call, a batch
of rays through giCTraceRays, materials handed over as MaterialX documents, scene options, one of the schedules of gi_options.h, hostile geometry (the oracle then
renders tests/test_hostile_inputs.py sanitised() of the description).  Prints one line per case and a summary; exit status 1 if any case differs.  The campaigns
that were run, and what they found, are listed in DESIGN.md section 8 (logs under profiles/)."""
from __future__ import annotations

import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fuzz_scenes import apply_edit, random_case  # noqa: E402

AOV_NAMES = ["normal", "barycentrics", "texcoords", "opacity", "tangents", "bitangents", "thinWalled", "objectId", "depth", "faceId", "instanceId", "doubleSided", "albedo"]
PATH_AOV_NAMES = ["nee", "bounces", "clockCycles"]
AOV_CLEAR = {"nee": (0.25, 0.5, 0.75, 0.0), "normal": (0.5, 0.5, 0.5, 0.5), "objectId": -1, "faceId": -1, "instanceId": -1, "depth": 1.0, "albedo": (0.1, 0.2, 0.3, 0.0)}


def differing(a, b):
    """Number of pixels whose bits differ (NaNs of any payload count as equal to NaNs: the host's and the device's default NaN differ in sign)."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape: return -1
    if a.dtype.kind == "f":
        ne = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    else:
        ne = a != b
    return int(ne.any(axis=-1).sum()) if a.ndim == 3 else int(ne.sum())


def apply_to_scene(gi, sc, desc, op):
    """The edit of tests/fuzz_scenes.py apply_edit through the C ABI, on the live scene."""
    import ctypes as C
    L, kind = sc.L, op["op"]
    fp = lambda v: (C.c_float * len(v))(*[float(x) for x in v])  # noqa: E731
    if "mesh" in op:
        k = op["mesh"]
        if kind == "mesh_remove":
            L.giCDestroyMesh(sc.meshes.pop(k)); return
        m = desc.meshes[k]
        if kind == "transforms": sc.set_mesh_instance_transforms(k, m.instance_transforms)
        elif kind == "visibility": L.giCSetMeshVisibility(sc.meshes[k], int(m.visible))
        elif kind == "material": L.giCSetMeshMaterial(sc.meshes[k], sc.materials[m.material])
        else: sc.set_mesh_transform(k, m.transform)
        return
    name, i = op["light"]
    short = name.split("_")[0]
    if kind == "light_add":
        l = desc.sphere_lights[i]; h = L.giCCreateSphereLight(sc.handle)
        L.giCSetSphereLightPosition(h, fp(l.pos)); L.giCSetSphereLightBaseEmission(h, fp(l.base_emission))
        L.giCSetSphereLightRadius(h, *[float(x) for x in l.radius]); L.giCSetSphereLightDiffuseSpecular(h, l.diffuse, l.specular)
        # (the wrapper keeps its lights in creation order: spheres first)
        at = sum(1 for kk, _ in sc.lights if kk == "sphere")
        sc.lights.insert(at, ("sphere", h)); return
    idx = [j for j, (kk, _) in enumerate(sc.lights) if kk == short][i]
    h = sc.lights[idx][1]
    if kind == "light_remove":
        {"sphere": L.giCDestroySphereLight, "distant": L.giCDestroyDistantLight, "rect": L.giCDestroyRectLight, "disk": L.giCDestroyDiskLight}[short](sc.handle, h)
        sc.lights.pop(idx); return
    l = getattr(desc, name)[i]
    if short == "sphere": L.giCSetSphereLightPosition(h, fp(l.pos)); L.giCSetSphereLightBaseEmission(h, fp(l.base_emission))
    elif short == "distant": L.giCSetDistantLightDirection(h, fp(l.direction)); L.giCSetDistantLightBaseEmission(h, fp(l.base_emission))
    elif short == "rect": L.giCSetRectLightOrigin(h, fp(l.origin)); L.giCSetRectLightBaseEmission(h, fp(l.base_emission))
    else: L.giCSetDiskLightOrigin(h, fp(l.origin)); L.giCSetDiskLightBaseEmission(h, fp(l.base_emission))


def run_case(gi, orc, seed, threads=8, use_options=True):
    """{"seed", "status": "same" | "differs" | "refused" | "error", "detail", ...} of one case.  `use_options` False: the case runs under the library's default
    schedule ($GATLING_OPTIONS is process-wide: concurrent cases cannot each have their own)."""
    import dataclasses
    desc, rs, w, h, ex = random_case(seed)
    if not use_options: ex["options"] = ""
    if _SCALE[0] > 1:   # --scale: the same case on an image `scale` times as wide and as high (more work than the path pool holds: batches, chunked sample buffers)
        S = _SCALE[0]; w, h = w * S, h * S
        if ex.get("rows"): ex["rows"] = (ex["rows"][0] * S, ex["rows"][1] * S, ex["rows"][2])
    rows = ex.get("rows")
    info = {"seed": seed, "tris": desc.triangle_count(), "w": w, "h": h, "spp": rs.spp, "bounces": rs.max_bounces, "nee": rs.next_event_estimation,
            "media": rs.medium_stack_size, "materials": len(desc.materials), "big": ex["big"], "aovs": ex["aovs"], "second": ex["second_call"],
            "rows": rows, "edit": ex.get("edit"), "hostile": ex.get("hostile", False), "extreme": ex.get("extreme") or "-", "mtlx": len(ex.get("mtlx") or {}), "rays": ex.get("trace_rays", 0), "scene_options": "/".join(f"{o}={v}" for o, v in ex.get("scene_options") or []) or "-", "options": ex.get("options") or "-"}
    if rows: r0, r1, stride = rows
    else: r0, r1, stride = 0, h, 1
    row_list = list(range(r0, r1, stride))
    t0 = time.perf_counter()
    if use_options: os.environ["GATLING_OPTIONS"] = ex.get("options") or ""
    try:
        docs = None
        if ex.get("mtlx"):
            from gatling_amd.mtlx_writer import material_to_mtlx
            docs = {mi: material_to_mtlx(desc.materials[mi], form) for mi, form in ex["mtlx"].items()}
        try:
            sc = gi.Scene(desc, mtlx_materials=docs)
        except Exception as e:  # the host refused the scene: the oracle has no say
            return dict(info, status="refused", detail=str(e)[:200])
        try:
            try:
                for opt, val in ex.get("scene_options") or []: sc.set_option(opt, val)
                rays = None
                if ex.get("trace_rays"):   # closest hits of a batch of rays (giCTraceRays), before anything was rendered
                    rr = np.random.default_rng(ex["trace_seed"]); nr = ex["trace_rays"]
                    ro = rr.uniform(-4, 4, (nr, 3)).astype(np.float32); rd = rr.normal(size=(nr, 3)); rd = (rd / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32)
                    rays = (ro, rd, sc.trace_rays(ro, rd))
                kw = {"rows": (r0, r1), "row_stride": stride}
                # (the colour AOV is bound in the same call; whole-frame cases also bind the three AOVs that follow whole paths: NEE, Bounces, ClockCycles)
                names = AOV_NAMES + (PATH_AOV_NAMES if not rows else [])
                aov = sc.render_aovs(rs, w, h, names, AOV_CLEAR, **kw) if ex["aovs"] else None
                img = aov["color"] if aov is not None else sc.render(rs, w, h, **kw)
                st = sc.stats()
                img2 = sc.render(rs, w, h, **kw) if ex["second_call"] else None
                img3 = None
                if ex.get("edit"):
                    op = apply_edit(desc, ex["edit"], ex["edit_seed"])   # (sc.desc IS desc: the oracle renders the edited description)
                    apply_to_scene(gi, sc, desc, op)
                    rs3 = dataclasses.replace(rs, progressive_accumulation=False)
                    img3 = sc.render(rs3, w, h, **kw)
            except gi.GiError as e:
                return dict(info, status="refused", detail=str(e)[:200])
        finally:
            sc.close()
    finally:
        if use_options: os.environ["GATLING_OPTIONS"] = ""
    info["gpu_s"] = round(time.perf_counter() - t0, 3)
    with _ORACLE_LOCK:   # (the oracle is test infrastructure with state of its own: one case at a time, also under --concurrent)
        return _compare(orc, seed, desc, rs, w, h, ex, info, threads, rows, row_list, r0, r1, stride, img, st, img2, img3, aov, rays)


_ORACLE_LOCK = __import__("threading").Lock()
_SCALE = [1]
_HARNESS = []
_BSDF_RANGES = [False]


def _compare(orc, seed, desc, rs, w, h, ex, info, threads, rows, row_list, r0, r1, stride, img, st, img2, img3, aov, rays):
    import dataclasses
    t0 = time.perf_counter()
    problems = []
    if ex.get("edit"):   # the oracle sees the scene as it was before the edit first
        before, _, _, _, _ = random_case(seed)
    else:
        before = desc
    if ex.get("hostile"):   # the oracle renders the documented reading of hostile geometry, written with ordinary triangles
        from test_hostile_inputs import sanitised
        before = sanitised(before); desc = sanitised(desc)
    okw = {"threads": threads, "row_list": row_list} if rows else {"threads": threads}
    ref, cnt = orc.render(before, rs, w, h, **okw)
    bad = differing(img, ref)
    if bad: problems.append(f"colour: {bad} of {len(row_list) * w} pixels")
    if (st["segments"], st["shadowRays"], st["samples"]) != (cnt["segments"], cnt["shadow_rays"], cnt["samples"]):
        problems.append(f"counts: segments {st['segments']} / {cnt['segments']}, shadow rays {st['shadowRays']} / {cnt['shadow_rays']}, samples {st['samples']} / {cnt['samples']}")
    if img2 is not None:
        ref2, _ = orc.render(before, rs, w, h, sample_offset=rs.spp, prev_color=ref, **okw)
        bad = differing(img2, ref2)
        if bad: problems.append(f"second call: {bad} pixels")
    if aov is not None:
        names = AOV_NAMES + (PATH_AOV_NAMES if not rows else [])
        refa = orc.render_aovs(before, rs, w, h, names, AOV_CLEAR)
        for name in names:
            full = np.asarray(refa[name]); full = full.reshape((h, w, 4) if full.size == h * w * 4 else (h, w))
            got = np.asarray(aov[name])
            if name in ("nee", "bounces"): got, full = got[..., :3], full[..., :3]   # (the shader writes .xyz)
            bad = differing(got, full[r0:r1:stride])
            if bad: problems.append(f"aov {name}: {bad}")
    if rays is not None:
        ro, rd, (tuv, ip) = rays
        rtuv, rip = orc.trace_rays(before, ro, rd)
        hit = rip[:, 0] >= 0
        if not np.array_equal(ip, rip): problems.append(f"trace_rays: {int((ip != rip).any(axis=1).sum())} of {len(ip)} hits name another triangle")
        elif not np.array_equal(tuv[hit].view(np.uint32), rtuv[hit].view(np.uint32)): problems.append("trace_rays: t / u / v differ")
    if img3 is not None:
        ref3, _ = orc.render(desc, dataclasses.replace(rs, progressive_accumulation=False), w, h, **okw)
        bad = differing(img3, ref3)
        if bad: problems.append(f"after the edit ({ex['edit']}): {bad} pixels")
    info["cpu_s"] = round(time.perf_counter() - t0, 3)
    info["finite"] = bool(np.isfinite(img).all())
    return dict(info, status="differs" if problems else "same", detail="; ".join(problems))


def _differs(gi, orc, desc, rs, w, h, threads):
    try:
        sc = gi.Scene(desc)
    except Exception:
        return False
    try:
        img = sc.render(rs, w, h); st = sc.stats()
    except gi.GiError:
        return False
    finally:
        sc.close()
    if _HOSTILE[0]:
        from test_hostile_inputs import sanitised
        desc = sanitised(desc)
    ref, cnt = orc.render(desc, rs, w, h, threads=threads)
    return differing(img, ref) != 0 or (st["segments"], st["shadowRays"]) != (cnt["segments"], cnt["shadow_rays"])


_HOSTILE = [False]   # reduce_case: the case under reduction has hostile geometry (the oracle renders its sanitised form)


def reduce_case(gi, orc, seed, threads=8):
    """Greedy reduction of a differing case: drops lights, meshes, instances, texture bindings, resets material parameters and render settings one at a time while
    the colour image (or the counts) still differ; prints what is left."""
    import copy
    import dataclasses
    from gatling_amd.scene import MaterialDesc, MAT_OPEN_PBR
    desc, rs, w, h, ex = random_case(seed)
    os.environ["GATLING_OPTIONS"] = ex.get("options") or ""
    _HOSTILE[0] = bool(ex.get("hostile"))
    print(f"== seed {seed}: hostile {ex.get('hostile')} options {ex.get('options') or '-'} rows {ex.get('rows')} edit {ex.get('edit')}")
    if not _differs(gi, orc, desc, rs, w, h, threads):
        print(f"seed {seed}: the first colour call does not differ (second call / AOVs only)"); return
    def attempt(mut):
        nonlocal desc, rs
        d2, r2 = copy.deepcopy(desc), dataclasses.replace(rs)
        if mut(d2, r2) is False: return False
        if _differs(gi, orc, d2, r2, w, h, threads): desc, rs = d2, r2; return True
        return False
    changed = True
    while changed:
        changed = False
        for name in ("sphere_lights", "distant_lights", "rect_lights", "disk_lights"):
            k = 0
            while k < len(getattr(desc, name)):
                if attempt(lambda d, r: getattr(d, name).pop(k)): changed = True
                else: k += 1
        if desc.dome_light is not None and attempt(lambda d, r: setattr(d, "dome_light", None)): changed = True
        k = 0
        while k < len(desc.meshes):
            if len(desc.meshes) > 1 and attempt(lambda d, r: d.meshes.pop(k)): changed = True
            else: k += 1
        for k, m in enumerate(desc.meshes):
            if len(m.instance_transforms) > 1:
                def one(d, r, k=k):
                    d.meshes[k].instance_transforms = d.meshes[k].instance_transforms[:1].copy(); d.meshes[k].instance_ids = None
                if attempt(one): changed = True
            if not np.array_equal(m.instance_transforms[0], np.eye(4, dtype=np.float32)):
                def ident(d, r, k=k): d.meshes[k].instance_transforms = np.eye(4, dtype=np.float32)[None].copy(); d.meshes[k].instance_ids = None
                if attempt(ident): changed = True
            if not np.array_equal(m.transform, np.eye(4, dtype=np.float32)):
                if attempt(lambda d, r, k=k: setattr(d.meshes[k], "transform", np.eye(4, dtype=np.float32))): changed = True
            for flag, plain in (("double_sided", False), ("left_handed", False)):
                if getattr(m, flag) != plain and attempt(lambda d, r, k=k, flag=flag, plain=plain: setattr(d.meshes[k], flag, plain)): changed = True
            for attr in ("primvars", "instancer_primvars"):
                j = 0
                while j < len(getattr(desc.meshes[k], attr)):
                    if attempt(lambda d, r, k=k, attr=attr, j=j: getattr(d.meshes[k], attr).pop(j)): changed = True
                    else: j += 1
            if len(m.faces) > 1:   # halve the faces
                for half in (0, 1):
                    def cut(d, r, k=k, half=half):
                        f = d.meshes[k].faces; n = len(f) // 2
                        d.meshes[k].faces = (f[:n] if half == 0 else f[n:]).copy()
                        if d.meshes[k].face_ids is not None: d.meshes[k].face_ids = (d.meshes[k].face_ids[:n] if half == 0 else d.meshes[k].face_ids[n:]).copy()
                    if attempt(cut): changed = True; break
        used = sorted({m.material for m in desc.meshes})
        for mi in used:
            mat = desc.materials[mi]
            for slot in list(mat.textures):
                if attempt(lambda d, r, mi=mi, slot=slot: d.materials[mi].textures.pop(slot)): changed = True
            for slot in list(mat.primvar_inputs):
                if attempt(lambda d, r, mi=mi, slot=slot: d.materials[mi].primvar_inputs.pop(slot)): changed = True
            ref = MaterialDesc.open_pbr() if mat.klass == MAT_OPEN_PBR else MaterialDesc.usd_preview_surface(klass=mat.klass)
            for i in range(len(mat.params)):
                if mat.params[i] != ref.params[i]:
                    def reset(d, r, mi=mi, i=i, v=ref.params[i]): d.materials[mi].params[i] = v
                    if attempt(reset): changed = True
        plain = RenderSettingsDefaults()
        for f, v in plain.items():
            if getattr(rs, f) != v and attempt(lambda d, r, f=f, v=v: setattr(r, f, v)): changed = True
        for f in ("spp", "max_bounces"):
            while getattr(rs, f) > (1 if f == "spp" else 0) and attempt(lambda d, r, f=f: setattr(r, f, getattr(r, f) - 1)): changed = True
    np.set_printoptions(precision=9, suppress=False, linewidth=200)
    print(f"== seed {seed} reduced: {w}x{h}", rs)
    print("camera", desc.camera)
    for name in ("sphere_lights", "distant_lights", "rect_lights", "disk_lights", "dome_light"):
        if getattr(desc, name): print(name, getattr(desc, name))
    for m in desc.meshes:
        print("mesh", m.name, "faces", len(m.faces), "material", m.material, "double_sided", m.double_sided, "left_handed", m.left_handed, "instances", len(m.instance_transforms),
              "transform", m.transform.reshape(-1).tolist())
        if len(m.faces) <= 2:
            for f in m.faces: print("   face", [(m.vertices[i]["pos"].tolist(), m.vertices[i]["norm"].tolist(), m.vertices[i]["tangent"].tolist(), float(m.vertices[i]["u"]),
                                                 float(m.vertices[i]["v"]), float(m.vertices[i]["bitangentSign"])) for i in f])
        if _HOSTILE[0]:
            print("   instance transforms", np.asarray(m.instance_transforms).reshape(-1, 16).tolist())
            v = m.vertices
            badv = [i for i in range(len(v)) if not (np.isfinite(v["pos"][i]).all() and np.isfinite(v["norm"][i]).all() and np.isfinite(v["tangent"][i]).all()
                                                   and np.isfinite(v["u"][i]) and np.isfinite(v["v"][i]) and np.isfinite(v["bitangentSign"][i]) and (np.abs(v["pos"][i]) <= 1e18).all())]
            print("   hostile vertices", [(i, v[i].tolist()) for i in badv[:6]], "used by faces", [int(k) for k in range(len(m.faces)) if set(m.faces[k].tolist()) & set(badv)][:8])
        for attr in ("primvars", "instancer_primvars"):
            for pv in getattr(m, attr): print("  ", attr, pv.name, "type", pv.type, "interp", pv.interpolation, "n", len(np.asarray(pv.data).reshape(-1)), np.asarray(pv.data).reshape(-1)[:8])
    for mi in sorted({m.material for m in desc.meshes}):
        mat = desc.materials[mi]
        ref = MaterialDesc.open_pbr() if mat.klass == MAT_OPEN_PBR else MaterialDesc.usd_preview_surface(klass=mat.klass)
        print("material", mi, "klass", mat.klass, "non-default params", {i: float(mat.params[i]) for i in range(len(mat.params)) if mat.params[i] != ref.params[i]},
              "textures", {k: v for k, v in mat.textures.items()}, "primvar inputs", mat.primvar_inputs)
    sc = gi.Scene(desc)
    try:
        img = sc.render(rs, w, h); st = sc.stats()
    finally:
        sc.close()
    if _HOSTILE[0]:
        from test_hostile_inputs import sanitised
        ref, cnt = orc.render(sanitised(desc), rs, w, h, threads=threads)
    else:
        ref, cnt = orc.render(desc, rs, w, h, threads=threads)
    ne = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1)
    ys, xs = np.nonzero(ne)
    print("differing pixels", int(ne.sum()), "counts", (st["segments"], st["shadowRays"]), (cnt["segments"], cnt["shadow_rays"]))
    for y, x in list(zip(ys, xs))[:4]:
        print("  pixel", x, y, "hip", img[y, x].tolist(), "oracle", ref[y, x].tolist())


def bsdf_case(gi, orc, seed, n=2048):
    """One random material (tests/fuzz_scenes.py _material, no bindings) on n random shading frames, view / light directions and random numbers -- grazing and
    below-horizon directions, geometric normals off the shading normal, random numbers at 0 and just below 1, both facings -- through giCDebugEvalBsdf and through
    the oracle's entry points: the 15 outputs per item (sampled direction, weight, pdf, event; evaluated diffuse / glossy / pdf) bit for bit."""
    from fuzz_scenes import _material
    rng = np.random.default_rng([0xb5df, seed])
    mat = _material(rng, 0, 0)
    mat.primvar_inputs = {}
    if _BSDF_RANGES[0]:   # --ranges: material inputs outside their documented ranges (finite: nothing is refused)
        for _ in range(int(rng.integers(1, 5))):
            i = int(rng.integers(0, 64))
            if i not in (6, 14, 15, 54): mat.params[i] = np.float32(rng.choice([-1.0, -0.25, 1.5, 7.0, 0.3, 50.0, 1.0e-6, 1.0e6]))
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    t = np.cross(nrm, rng.normal(size=(n, 3))); t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(nrm, t)

    def direction(kind):
        v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        z = np.abs(v[:, 2])
        if kind == "grazing": z = z * 10.0 ** rng.uniform(-6, 0, n)
        elif kind == "any": z = v[:, 2]
        v[:, 2] = z; v /= np.linalg.norm(v, axis=1, keepdims=True)
        return v[:, :1] * t + v[:, 1:2] * b + v[:, 2:3] * nrm
    k1 = np.where(rng.uniform(size=(n, 1)) < 0.3, direction("grazing"), direction("up"))
    k2 = np.where(rng.uniform(size=(n, 1)) < 0.3, direction("grazing"), np.where(rng.uniform(size=(n, 1)) < 0.3, direction("any"), direction("up")))
    gn = nrm + rng.normal(scale=0.3, size=(n, 3)) * (rng.uniform(size=(n, 1)) < 0.5); gn /= np.linalg.norm(gn, axis=1, keepdims=True)
    xi = rng.uniform(size=(n, 4))
    edge = rng.uniform(size=(n, 3))
    xi[:, :3] = np.where(edge < 0.03, 0.0, np.where(edge < 0.06, np.float32(1.0) - np.float32(2.0 ** -24), xi[:, :3]))
    xi[:, 3] = rng.uniform(size=n) < 0.3   # back-facing items
    items = np.concatenate([nrm, t, b, gn, k1, k2, xi], axis=1).astype(np.float32)
    got = gi.bsdf_debug(mat, items)
    want = orc.bsdf_debug(mat, items)
    ne = (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
    rows = np.nonzero(ne.any(axis=1))[0]
    detail = ""
    if len(rows):
        r = int(rows[0])
        detail = f"{len(rows)} of {n} items; first: item {r} columns {np.nonzero(ne[r])[0].tolist()} hip {got[r].tolist()} oracle {want[r].tolist()} input {items[r].tolist()} klass {mat.klass} params {[float(x) for x in mat.params]}"
    return {"seed": seed, "klass": mat.klass, "finite": bool(np.isfinite(got).all()), "status": "differs" if len(rows) else "same", "detail": detail}


def gscn_case(orc, seed, threads, tmpdir):
    """The case written to a scene file (gatling_amd/scenefile.py), rendered by the plain-C client tools/gi_render (strict C99 over include/gi_c.h, its own parser of
    the file) in a process of its own, against the oracle on the description in memory: the file format, its two implementations and the client on the whole generator."""
    import subprocess
    from gatling_amd.scenefile import save_scene
    from test_scenefile import _build_harness
    desc, rs, w, h, ex = random_case(seed)
    info = {"seed": seed, "tris": desc.triangle_count(), "w": w, "h": h, "spp": rs.spp, "bounces": rs.max_bounces, "hostile": ex["hostile"]}
    if ex["hostile"]:
        from test_hostile_inputs import sanitised
        ref_desc = sanitised(desc)
    else:
        ref_desc = desc
    path, out = os.path.join(tmpdir, f"s{seed}.gscn"), os.path.join(tmpdir, f"o{seed}.raw")
    save_scene(path, desc, rs, w, h)
    if not _HARNESS: _HARNESS.append(_build_harness())   # (compiled once per campaign)
    run = subprocess.run([_HARNESS[0], path, out], capture_output=True, text=True, timeout=300)
    try:
        if run.returncode != 0:
            return dict(info, status="refused", detail=(run.stdout + run.stderr).strip().replace("\n", " | ")[-300:])
        img = np.fromfile(out, np.float32).reshape(h, w, 4)
    finally:
        for f in (path, out):
            if os.path.exists(f): os.remove(f)
    ref, _ = orc.render(ref_desc, rs, w, h, threads=threads)
    bad = differing(img, ref)
    return dict(info, status="differs" if bad else "same", detail=f"colour: {bad} of {w * h} pixels" if bad else "")


def RenderSettingsDefaults():
    return {"next_event_estimation": False, "medium_stack_size": 0, "depth_of_field": False, "clipping_planes": False, "filter_importance_sampling": False,
            "jittered_sampling": False, "light_intensity_multiplier": 1.0, "max_sample_value": 1.0e6, "rr_bounce_offset": 100, "meters_per_scene_unit": 1.0,
            "dome_light_camera_visible": True, "clear_color": (0.0, 0.0, 0.0, 0.0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("seeds", help="first:last (exclusive) or a comma list")
    ap.add_argument("--threads", type=int, default=min(64, os.cpu_count() or 8))
    ap.add_argument("--log", default=None)
    ap.add_argument("--reduce", action="store_true", help="reduce each differing case of the list to what still differs")
    ap.add_argument("--concurrent", type=int, default=1, help="this many host threads run cases at the same time, each on scenes of its own (the library serialises "
                    "nothing but what shares state: include/gi_c.h); the oracle then runs single-threaded per case")
    ap.add_argument("--ranges", action="store_true", help="with --bsdf: material inputs outside their documented ranges")
    ap.add_argument("--gscn", action="store_true", help="each case through a scene file and the plain-C client tools/gi_render instead of the ctypes binding")
    ap.add_argument("--scale", type=int, default=1, help="render every case `scale` times as wide and as high")
    ap.add_argument("--bsdf", action="store_true", help="the seeds are BSDF cases (one random material on 2 048 random frames / directions each) instead of renders")
    a = ap.parse_args()
    seeds = list(range(*map(int, a.seeds.split(":")))) if ":" in a.seeds else [int(x) for x in a.seeds.split(",")]
    from gatling_amd import capi as gi
    from oracle import orc
    orc.build(); orc.lib(); gi.initialize(0)
    _SCALE[0] = max(1, a.scale); _BSDF_RANGES[0] = a.ranges
    if a.reduce:
        for seed in seeds:
            try: reduce_case(gi, orc, seed, a.threads)
            except Exception: traceback.print_exc()
        return 0
    import tempfile
    tmpdir = tempfile.mkdtemp(prefix="fuzz_gscn_") if a.gscn else None
    log = open(a.log, "w") if a.log else None
    tally = {}
    t0 = time.perf_counter()
    def one(seed):
        try:
            if a.bsdf: return bsdf_case(gi, orc, seed)
            if a.gscn: return gscn_case(orc, seed, a.threads, tmpdir)
            return run_case(gi, orc, seed, a.threads if a.concurrent == 1 else 1, use_options=a.concurrent == 1)
        except Exception:
            return {"seed": seed, "status": "error", "detail": traceback.format_exc(limit=3).replace("\n", " | ")[-400:]}
    if a.concurrent > 1:
        from concurrent.futures import ThreadPoolExecutor
        os.environ["GATLING_OPTIONS"] = ""
        results = ThreadPoolExecutor(max_workers=a.concurrent).map(one, seeds)
    else:
        results = (one(seed) for seed in seeds)
    for r in results:
        tally[r["status"]] = tally.get(r["status"], 0) + 1
        line = " ".join(f"{k}={v}" for k, v in r.items())
        if r["status"] != "same" or log is None: print(line, flush=True)
        if log: log.write(line + "\n"); log.flush()
    summary = f"# {len(seeds)} cases in {time.perf_counter() - t0:.0f} s: " + ", ".join(f"{v} {k}" for k, v in sorted(tally.items()))
    print(summary)
    if log: log.write(summary + "\n")
    return 1 if tally.get("differs", 0) or tally.get("error", 0) else 0


if __name__ == "__main__":
    sys.exit(main())
