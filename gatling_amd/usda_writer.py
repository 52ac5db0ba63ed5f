"""Writes a :class:`SceneDesc` as ``.usda`` so that the generated configurations (C3-C5, ``gatling_amd/scenes.py``) can be handed
to the reference's own ``gatling <scene.usd> <render.png>`` on a box that has it ("the generator must emit
both .usda and the flat binary the C harness loads"; the binary is :mod:`gatling_amd.scenefile`).

What the stage holds is what hdGatling reads back into the same gi calls (reference derivations cited per prim):

* ``Camera``: transform rows (right, up, -forward, position); ``focalLength`` / ``verticalAperture`` in tenths of a scene unit so
  that ``renderPass.cpp:191-228`` recovers ``vfov`` and ``focalLength``;
* ``Material``: one ``Shader`` with ``info:id = "UsdPreviewSurface"`` or ``"ND_open_pbr_surface_surfaceshader"`` and constant
  inputs; inputs bound to a primvar get a ``UsdPrimvarReader_*`` shader (``scene_data_lookup_*``).  Texture bindings are float
  pixel arrays without a file behind them and are NOT written (the constant value stays);
* ``Mesh``: triangles, ``points`` / ``normals`` / ``primvars:st`` with vertex interpolation, further float primvars; a mesh with
  several instances becomes a ``class`` prototype plus one ``instanceable`` Xform per instance (native instancing, which Hydra
  turns into the instancer path ``instancer.cpp:203-340``), each carrying ``M_prim * M_instance``;
* lights: ``intensity 1``, ``color`` = base emission, ``normalize`` off, so ``light.cpp:58-94`` yields the same base emission; the
  transform rows are (t0 * sx, t1 * sy, normal, origin) -- ``light.cpp:226-233, 289-296``;
* dome light: ``inputs:texture:file`` names a Radiance ``.hdr`` written next to the stage (RGBE: 8-bit mantissas, lossy).

The reader for the same subset is :mod:`gatling_amd.usda` (round trip: ``tests/test_usda_writer.py``)."""
from __future__ import annotations

import math
import os
import re

import numpy as np

from .scene import (INTERP_CONSTANT, INTERP_UNIFORM, INTERP_VERTEX, MAT_OPEN_PBR, MAT_USD_PREVIEW_SURFACE, P_BASE_COLOR, P_BASE_WEIGHT,
                    P_CLEARCOAT, P_CLEARCOAT_ROUGHNESS, P_COAT_COLOR, P_COAT_ROTATION, P_SPECULAR_ROTATION, P_COAT_DARKENING, P_COAT_IOR, P_DIFFUSE_ROUGHNESS, P_FUZZ_COLOR, P_FUZZ_ROUGHNESS, P_FUZZ_WEIGHT, P_THIN_WALLED, P_SUBSURFACE_WEIGHT, P_SUBSURFACE_COLOR, P_SUBSURFACE_ANISOTROPY, P_SUBSURFACE_RADIUS, P_SUBSURFACE_RADIUS_SCALE, P_SPECULAR_ANISOTROPY, P_COAT_ANISOTROPY, P_THIN_FILM_WEIGHT, P_THIN_FILM_THICKNESS, P_THIN_FILM_IOR, P_EMISSION, P_IOR, P_METALLIC, P_OPACITY,
                    P_OPACITY_THRESHOLD, P_ROUGHNESS, P_SPECULAR_COLOR, P_SPECULAR_WEIGHT, P_TRANSMISSION_COLOR, P_TRANSMISSION_DEPTH,
                    P_TRANSMISSION_SCATTER, P_TRANSMISSION_SCATTER_ANISOTROPY, P_TRANSMISSION_WEIGHT, P_USE_SPECULAR_WORKFLOW, SceneDesc,
                    TEX_BASE_COLOR, TEX_EMISSION, TEX_METALLIC, TEX_NORMAL, TEX_ROUGHNESS, TEX_TRANSMISSION_COLOR, TEX_TRANSMISSION_WEIGHT)

OPEN_PBR_ID = "ND_open_pbr_surface_surfaceshader"
# (input name, parameter index, components) -- the inputs this core implements (open_pbr_surface.mtlx:11-92)
OPEN_PBR_INPUTS = [("base_weight", P_BASE_WEIGHT, 1), ("base_color", P_BASE_COLOR, 3), ("base_diffuse_roughness", P_DIFFUSE_ROUGHNESS, 1),
                   ("base_metalness", P_METALLIC, 1), ("specular_weight", P_SPECULAR_WEIGHT, 1), ("specular_color", P_SPECULAR_COLOR, 3),
                   ("specular_roughness", P_ROUGHNESS, 1), ("specular_ior", P_IOR, 1), ("transmission_weight", P_TRANSMISSION_WEIGHT, 1),
                   ("transmission_color", P_TRANSMISSION_COLOR, 3), ("transmission_depth", P_TRANSMISSION_DEPTH, 1),
                   ("transmission_scatter", P_TRANSMISSION_SCATTER, 3), ("transmission_scatter_anisotropy", P_TRANSMISSION_SCATTER_ANISOTROPY, 1),
                   ("coat_weight", P_CLEARCOAT, 1), ("coat_color", P_COAT_COLOR, 3), ("coat_roughness", P_CLEARCOAT_ROUGHNESS, 1),
                   ("coat_ior", P_COAT_IOR, 1), ("coat_darkening", P_COAT_DARKENING, 1), ("fuzz_weight", P_FUZZ_WEIGHT, 1), ("fuzz_color", P_FUZZ_COLOR, 3),
                   ("fuzz_roughness", P_FUZZ_ROUGHNESS, 1), ("geometry_thin_walled", P_THIN_WALLED, 0), ("geometry_opacity", P_OPACITY, 1),
                   ("subsurface_weight", P_SUBSURFACE_WEIGHT, 1), ("subsurface_color", P_SUBSURFACE_COLOR, 3), ("subsurface_scatter_anisotropy", P_SUBSURFACE_ANISOTROPY, 1),
                   ("subsurface_radius", P_SUBSURFACE_RADIUS, 1), ("subsurface_radius_scale", P_SUBSURFACE_RADIUS_SCALE, 3),
                   ("specular_roughness_anisotropy", P_SPECULAR_ANISOTROPY, 1), ("coat_roughness_anisotropy", P_COAT_ANISOTROPY, 1), ("coat_rotation", P_COAT_ROTATION, 1), ("specular_rotation", P_SPECULAR_ROTATION, 1),
                   ("thin_film_weight", P_THIN_FILM_WEIGHT, 1), ("thin_film_thickness", P_THIN_FILM_THICKNESS, 1), ("thin_film_ior", P_THIN_FILM_IOR, 1)]
UPS_INPUTS = [("diffuseColor", P_BASE_COLOR, 3), ("emissiveColor", P_EMISSION, 3), ("useSpecularWorkflow", P_USE_SPECULAR_WORKFLOW, 0),
              ("specularColor", P_SPECULAR_COLOR, 3), ("metallic", P_METALLIC, 1), ("roughness", P_ROUGHNESS, 1), ("clearcoat", P_CLEARCOAT, 1),
              ("clearcoatRoughness", P_CLEARCOAT_ROUGHNESS, 1), ("opacity", P_OPACITY, 1), ("opacityThreshold", P_OPACITY_THRESHOLD, 1), ("ior", P_IOR, 1)]
# texturable slot -> (UsdPreviewSurface input, OpenPBR input, primvar reader type, value type)
SLOT_INPUTS = {TEX_BASE_COLOR: ("diffuseColor", "base_color", "float3", "color3f"), TEX_EMISSION: ("emissiveColor", "emission_color", "float3", "color3f"),
               TEX_ROUGHNESS: ("roughness", "specular_roughness", "float", "float"), TEX_METALLIC: ("metallic", "base_metalness", "float", "float"),
               TEX_NORMAL: ("normal", "geometry_normal", "float3", "normal3f"),
               TEX_TRANSMISSION_WEIGHT: (None, "transmission_weight", "float", "float"),     # (OpenPBR only: UsdPreviewSurface has no transmission)
               TEX_TRANSMISSION_COLOR: (None, "transmission_color", "float3", "color3f")}
PRIMVAR_TYPES = {0: "float", 1: "float2", 2: "float3", 3: "float4"}
PRIMVAR_INTERP = {INTERP_CONSTANT: "constant", INTERP_UNIFORM: "uniform", INTERP_VERTEX: "vertex"}


def _f(x) -> str:
    return "%.9g" % float(x)


def _tuple(v) -> str:
    return "(" + ", ".join(_f(x) for x in v) + ")"


def _matrix(m) -> str:
    m = np.asarray(m, np.float64).reshape(4, 4)
    return "( " + ", ".join(_tuple(r) for r in m) + " )"


def _tuples(a, comps) -> str:
    a = np.asarray(a, np.float32).reshape(-1, comps)
    if comps == 1:
        return "[" + ", ".join(_f(x) for x in a[:, 0]) + "]"
    return "[" + ", ".join(_tuple(r) for r in a) + "]"


def _ident(name: str, prefix: str) -> str:
    s = re.sub(r"[^A-Za-z0-9_]", "_", name.strip("/").split("/")[-1] or "x")
    return f"{prefix}_{s}"


def _xform_lines(m, ind) -> list:
    return [f'{ind}matrix4d xformOp:transform = {_matrix(m)}', f'{ind}uniform token[] xformOpOrder = ["xformOp:transform"]']


def write_hdr(path, pixels):
    """Flat (uncompressed) Radiance RGBE; row 0 of ``pixels`` (the v = 0 side) is written LAST: the file is top to bottom."""
    px = np.asarray(pixels, np.float32)[::-1, :, :3]
    mx = px.max(axis=2)
    e = np.where(mx > 1e-32, np.floor(np.log2(np.maximum(mx, 1e-38))) + 1, -128).astype(np.int32)
    scale = np.where(mx > 1e-32, np.ldexp(np.float32(256.0), -e), 0.0).astype(np.float32)
    rgbe = np.zeros(px.shape[:2] + (4,), np.uint8)
    rgbe[..., :3] = np.clip(px * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(mx > 1e-32, e + 128, 0).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (px.shape[0], px.shape[1]))
        f.write(rgbe.tobytes())


def _material_lines(i, m, root) -> list:
    name = _ident(m.name, f"m{i}")
    path = f"{root}/Materials/{name}"
    out = [f'        def Material "{name}"', "        {"]
    if m.klass == MAT_OPEN_PBR:
        out.append(f"            token outputs:mtlx:surface.connect = <{path}/Shader.outputs:surface>")
    else:
        out.append(f"            token outputs:surface.connect = <{path}/Shader.outputs:surface>")
    out.append(f"            custom int gatling:materialClass = {int(m.klass)}")
    out += ['            def Shader "Shader"', "            {"]
    p = np.asarray(m.params, np.float32)
    bound = {}
    for slot, pv in m.primvar_inputs.items():
        ups, opbr, rtype, vtype = SLOT_INPUTS[slot]
        if m.klass != MAT_OPEN_PBR and ups is None:
            continue
        bound[opbr if m.klass == MAT_OPEN_PBR else ups] = (slot, pv, rtype, vtype)
    if m.klass == MAT_OPEN_PBR:
        out.append(f'                uniform token info:id = "{OPEN_PBR_ID}"')
        inputs = list(OPEN_PBR_INPUTS)
        lum = float(p[P_EMISSION:P_EMISSION + 3].max())
        out.append(f"                float inputs:emission_luminance = {_f(lum)}")
        col = p[P_EMISSION:P_EMISSION + 3] / np.float32(lum) if lum > 0 else np.ones(3, np.float32)
        if "emission_color" not in bound:
            out.append(f"                color3f inputs:emission_color = {_tuple(col)}")
    else:
        out.append('                uniform token info:id = "UsdPreviewSurface"')
        inputs = list(UPS_INPUTS)
    for key, idx, comps in inputs:
        if key in bound:
            continue
        if comps == 3:
            out.append(f"                color3f inputs:{key} = {_tuple(p[idx:idx + 3])}")
        elif comps == 0:
            out.append(f"                int inputs:{key} = {int(p[idx])}")
        else:
            out.append(f"                float inputs:{key} = {_f(p[idx])}")
    for key, (slot, pv, rtype, vtype) in bound.items():
        out.append(f"                {vtype} inputs:{key}.connect = <{path}/pv{slot}.outputs:result>")
    out += ["                token outputs:surface", "            }"]
    for key, (slot, pv, rtype, vtype) in bound.items():
        out += [f'            def Shader "pv{slot}"', "            {", f'                uniform token info:id = "UsdPrimvarReader_{rtype}"',
                f'                string inputs:varname = "{pv}"', f"                {rtype} outputs:result", "            }"]
    if m.textures:
        out.append(f"            # {len(m.textures)} texture binding(s) not written: float pixel arrays have no file behind them")
    out.append("        }")
    return out, path


def _mesh_body(m, mat_path, ind) -> list:
    v = m.vertices
    faces = np.asarray(m.faces, np.int64).reshape(-1, 3)
    out = [f"{ind}int[] faceVertexCounts = [" + ", ".join(["3"] * len(faces)) + "]",
           f"{ind}int[] faceVertexIndices = [" + ", ".join(str(int(x)) for x in faces.reshape(-1)) + "]",
           f"{ind}point3f[] points = {_tuples(v['pos'], 3)}",
           f"{ind}normal3f[] normals = {_tuples(v['norm'], 3)} (", f'{ind}    interpolation = "vertex"', f"{ind})",
           f"{ind}texCoord2f[] primvars:st = {_tuples(np.stack([v['u'], v['v']], axis=1), 2)} (", f'{ind}    interpolation = "vertex"', f"{ind})",
           f"{ind}uniform bool doubleSided = {1 if m.double_sided else 0}",
           f'{ind}uniform token orientation = "{"leftHanded" if m.left_handed else "rightHanded"}"',
           f'{ind}uniform token subdivisionScheme = "none"']
    if not m.visible:
        out.append(f'{ind}token visibility = "invisible"')
    if mat_path:
        out.append(f"{ind}rel material:binding = <{mat_path}>")
    for pv in m.primvars:
        if pv.interpolation in PRIMVAR_INTERP and pv.type in PRIMVAR_TYPES:
            comps = pv.type + 1
            out += [f"{ind}{PRIMVAR_TYPES[pv.type]}[] primvars:{pv.name} = {_tuples(pv.data, comps)} (",
                    f'{ind}    interpolation = "{PRIMVAR_INTERP[pv.interpolation]}"', f"{ind})"]
    return out


def write_usda(path, desc: SceneDesc, aspect: float = 16.0 / 9.0, up_axis: str = "Z"):
    root = "/Root"
    L = ["#usda 1.0", "(", '    defaultPrim = "Root"', "    metersPerUnit = 1", f'    upAxis = "{up_axis}"', ")", "", 'def Xform "Root"', "{"]
    # ---- camera
    c = desc.camera
    fwd = np.asarray(c.forward, np.float64); fwd /= np.linalg.norm(fwd)
    up = np.asarray(c.up, np.float64)
    right = np.cross(fwd, up); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    cam = np.eye(4); cam[0, :3] = right; cam[1, :3] = up; cam[2, :3] = -fwd; cam[3, :3] = c.position
    focal = float(c.focal_length) * 10.0                                # GfCamera::FOCAL_LENGTH_UNIT = 0.1
    vap = 2.0 * float(c.focal_length) * math.tan(float(c.vfov) * 0.5) * 10.0
    L += ['    def Camera "Camera"', "    {", f"        float2 clippingRange = ({_f(c.clip_start)}, {_f(c.clip_end)})", f"        float focalLength = {_f(focal)}",
          f"        float verticalAperture = {_f(vap)}", f"        float horizontalAperture = {_f(vap * aspect)}", f"        float fStop = {_f(c.f_stop)}",
          f"        float focusDistance = {_f(c.focus_distance)}", f"        float exposure = {_f(c.exposure)}"] + _xform_lines(cam, "        ") + ["    }"]
    # ---- materials
    L += ['    def Scope "Materials"', "    {"]
    mat_paths = []
    for i, m in enumerate(desc.materials):
        lines, p = _material_lines(i, m, root)
        L += lines
        mat_paths.append(p)
    L.append("    }")
    # ---- meshes
    for i, m in enumerate(desc.meshes):
        name = _ident(m.name, f"mesh{i}")
        mat = mat_paths[m.material] if 0 <= m.material < len(mat_paths) else None
        inst = np.asarray(m.instance_transforms, np.float64).reshape(-1, 4, 4)
        prim = np.asarray(m.transform, np.float64).reshape(4, 4)
        api = ["        prepend apiSchemas = [\"MaterialBindingAPI\"]"]
        if len(inst) == 1:
            L += [f'    def Mesh "{name}" (', *api, "    )", "    {"] + _mesh_body(m, mat, "        ") + _xform_lines(prim @ inst[0], "        ") + ["    }"]
        else:
            L += [f'    class Xform "_proto_{name}"', "    {", '        def Mesh "geo" (', *["    " + a for a in api], "        )", "        {"]
            L += _mesh_body(m, mat, "            ") + ["        }", "    }", f'    def Xform "{name}"', "    {"]
            for j, it in enumerate(inst):
                L += [f'        def Xform "inst{j}" (', "            instanceable = true", f"            inherits = <{root}/_proto_{name}>", "        )", "        {"]
                L += _xform_lines(prim @ it, "            ") + ["        }"]
            L.append("    }")
    # ---- lights
    def light(kind, name, xf, extra, l):
        out = [f'    def {kind} "{name}"', "    {", "        float inputs:intensity = 1", f"        color3f inputs:color = {_tuple(l.base_emission)}",
               "        bool inputs:normalize = 0", f"        float inputs:diffuse = {_f(l.diffuse)}", f"        float inputs:specular = {_f(l.specular)}"]
        return out + ["        " + e for e in extra] + _xform_lines(xf, "        ") + ["    }"]

    def frame(t0, t1, sx, sy, origin):
        t0 = np.asarray(t0, np.float64); t1 = np.asarray(t1, np.float64)
        m = np.eye(4); m[0, :3] = t0 * sx; m[1, :3] = t1 * sy; m[2, :3] = np.cross(t0, t1); m[3, :3] = origin
        return m
    for i, l in enumerate(desc.sphere_lights):
        m = np.diag([l.radius[0], l.radius[1], l.radius[2], 1.0]); m[3, :3] = l.pos
        L += light("SphereLight", f"sphereLight{i}", m, ["float inputs:radius = 1"], l)
    for i, l in enumerate(desc.distant_lights):
        d = np.asarray(l.direction, np.float64); d /= np.linalg.norm(d)
        a = np.array([1.0, 0, 0]) if abs(d[0]) < 0.9 else np.array([0, 1.0, 0])
        x = np.cross(a, -d); x /= np.linalg.norm(x)
        m = np.eye(4); m[0, :3] = x; m[1, :3] = np.cross(-d, x); m[2, :3] = -d
        L += light("DistantLight", f"distantLight{i}", m, [f"float inputs:angle = {_f(math.degrees(l.angle))}"], l)
    for i, l in enumerate(desc.rect_lights):
        L += light("RectLight", f"rectLight{i}", frame(l.t0, l.t1, 1.0, 1.0, l.origin), [f"float inputs:width = {_f(l.width)}", f"float inputs:height = {_f(l.height)}"], l)
    for i, l in enumerate(desc.disk_lights):
        L += light("DiskLight", f"diskLight{i}", frame(l.t0, l.t1, l.radius_x, l.radius_y, l.origin), ["float inputs:radius = 1"], l)
    d = desc.dome_light
    if d is not None:
        extra = []
        if 0 <= d.texture < len(desc.textures):
            hdr = os.path.splitext(os.path.basename(str(path)))[0] + "_dome.hdr"
            write_hdr(os.path.join(os.path.dirname(str(path)), hdr), desc.textures[d.texture])
            extra.append(f"asset inputs:texture:file = @./{hdr}@")
        # light.cpp:388-390 hands (x, y, z, -w) of the transform's rotation quaternion to giSetDomeLightRotation
        x, y, z, w = [float(q) for q in d.rotation]
        w = -w
        rot = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w), 0], [2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w), 0],
                        [2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y), 0], [0, 0, 0, 1.0]])
        L += light("DomeLight", "domeLight", rot, extra, d)
    L += ["}", ""]
    with open(path, "w") as f:
        f.write("\n".join(L))
