"""A tiny reader for the subset of ``.usda`` that ``/root/reference/cornell.usda`` (and the harness's own
generated scenes) use: nested ``def`` prims, ``xformOp:transform`` matrices, ``Mesh`` topology / points /
normals, ``Camera`` attributes, ``Material`` -> ``UsdPreviewSurface`` constant inputs and
``material:binding`` relationships.  OpenUSD (``pxr``) is not installable in this environment; in the
product the Hydra delegate does this ingestion and stays as-is.

The derivations that hdGatling performs between USD and the gi boundary are restated in
:mod:`gatling_amd.meshprep` (mesh) and :func:`camera_from_prim` (``renderPass.cpp:191-228``).
"""
from __future__ import annotations

import math
import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from .meshprep import build_mesh_arrays
from .scene import CameraDesc, MaterialDesc, MeshDesc, SceneDesc, MAT_USD_PREVIEW_SURFACE

_TOKEN = re.compile(r"""
    \s+ | \#[^\n]* |
    (?P<str>"(?:[^"\\]|\\.)*") |
    (?P<path><[^>]*>) |
    (?P<asset>@[^@]*@) |
    (?P<num>[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)) |
    (?P<id>[A-Za-z_][A-Za-z0-9_:.\[\]]*) |
    (?P<p>[(){}\[\]=,])
""", re.X)


def _tokenize(text: str):
    pos, out = 0, []
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise ValueError(f"usda: cannot tokenise at {text[pos:pos + 40]!r}")
        pos = m.end()
        k = m.lastgroup
        if k:
            out.append((k, m.group(k)))
    return out


@dataclass
class Prim:
    type: str
    name: str
    path: str
    meta: dict = field(default_factory=dict)
    attrs: Dict[str, object] = field(default_factory=dict)
    attr_meta: Dict[str, dict] = field(default_factory=dict)
    children: List["Prim"] = field(default_factory=list)
    specifier: str = "def"


class _Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def expect(self, val):
        k, v = self.next()
        if v != val:
            raise ValueError(f"usda: expected {val!r}, got {v!r}")

    def value(self):
        k, v = self.peek()
        if v == "(":
            return self.seq("(", ")")
        if v == "[":
            return self.seq("[", "]")
        self.next()
        if k == "num":
            return float(v)
        if k == "str":
            return v[1:-1]
        if k == "path":
            return ("path", v[1:-1])
        if k == "asset":
            return ("asset", v[1:-1])
        return v  # identifier / token

    def seq(self, o, c):
        self.expect(o)
        items = []
        while self.peek()[1] != c:
            items.append(self.value())
            if self.peek()[1] == ",":
                self.next()
        self.expect(c)
        return items

    def metadata(self):
        """( key = value ... ) blocks; 'prepend/append' list-op keywords are accepted and ignored."""
        meta = {}
        self.expect("(")
        while self.peek()[1] != ")":
            k, v = self.next()
            if k == "str":  # doc string
                continue
            if v in ("prepend", "append", "add", "delete"):
                k, v = self.next()
            self.expect("=")
            meta[v] = self.value()
        self.expect(")")
        return meta

    def prim_body(self, prim: Prim):
        self.expect("{")
        while self.peek()[1] != "}":
            k, v = self.peek()
            if v in ("def", "over", "class"):
                prim.children.append(self.prim(prim.path))
                continue
            # attribute / relationship:  [custom] [uniform] type name [= value] [(meta)]
            words = []
            while self.peek()[1] not in ("=", "}", "(") and self.peek()[0] in ("id",):
                words.append(self.next()[1])
                if len(words) >= 2 and words[-2] not in ("custom", "uniform", "varying", "prepend", "append"):
                    break
            name = words[-1]
            val = None
            if self.peek()[1] == "=":
                self.next()
                val = self.value()
            prim.attrs[name] = val
            if self.peek()[1] == "(":
                prim.attr_meta[name] = self.metadata()
        self.expect("}")

    def prim(self, parent_path):
        spec = self.next()[1]  # def / over / class
        k, v = self.next()
        typ = ""
        if k == "id":
            typ = v
            k, v = self.next()
        name = v[1:-1]
        p = Prim(typ, name, f"{parent_path}/{name}", specifier=spec)
        if self.peek()[1] == "(":
            p.meta = self.metadata()
        self.prim_body(p)
        return p

    def stage(self):
        root = Prim("", "", "")
        if self.peek()[1] == "(":
            root.meta = self.metadata()
        while self.peek()[0] is not None:
            if self.peek()[1] in ("def", "over", "class"):
                root.children.append(self.prim(""))
            else:
                self.next()
        return root


def parse_usda(text: str) -> Prim:
    if text.startswith("#usda"):
        text = text[text.index("\n"):]
    return _Parser(_tokenize(text)).stage()


def _matrix(val) -> np.ndarray:
    return np.asarray(val, np.float64).reshape(4, 4)


def camera_from_prim(prim: Prim, world: np.ndarray) -> CameraDesc:
    """HdGatlingRenderPass::_ConstructGiCamera, renderPass.cpp:191-228 (doubles, then cast to float)."""
    pos = np.array([0.0, 0.0, 0.0, 1.0]) @ world
    fwd = np.array([0.0, 0.0, -1.0, 0.0]) @ world
    up = np.array([0.0, 1.0, 0.0, 0.0]) @ world
    fwd = fwd[:3] / np.linalg.norm(fwd[:3])
    up = up[:3] / np.linalg.norm(up[:3])
    a = prim.attrs
    aperture = np.float32(float(a.get("verticalAperture", 15.2908)) * 0.1)   # GfCamera::APERTURE_UNIT
    focal = np.float32(float(a.get("focalLength", 50.0)) * 0.1)              # GfCamera::FOCAL_LENGTH_UNIT
    vfov = np.float32(2.0) * np.float32(math.atan(float(aperture / (np.float32(2.0) * focal))))
    clip = a.get("clippingRange", [1.0, 1000000.0])
    return CameraDesc(position=tuple(np.float32(pos[:3])), forward=tuple(np.float32(fwd)), up=tuple(np.float32(up)),
                      vfov=float(vfov), f_stop=float(a.get("fStop", 0.0)), focus_distance=float(a.get("focusDistance", 0.0)),
                      focal_length=float(focal), clip_start=float(clip[0]), clip_end=float(clip[1]),
                      exposure=float(a.get("exposure", 0.0)))


def _material_from_prim(prim: Prim, klass: int) -> MaterialDesc:
    """UsdPreviewSurface / open_pbr_surface constant inputs; inputs connected to a UsdPrimvarReader become primvar bindings.
    ``custom int gatling:materialClass`` (written by usda_writer) overrides the class chosen by the caller."""
    from .usda_writer import OPEN_PBR_ID, OPEN_PBR_INPUTS, SLOT_INPUTS
    from .scene import MAT_OPEN_PBR, P_EMISSION
    if prim.attrs.get("gatling:materialClass") is not None:
        klass = int(prim.attrs["gatling:materialClass"])
    shaders = {c.name: c for c in prim.children if c.type == "Shader"}
    opbr = next((c for c in shaders.values() if c.attrs.get("info:id") == OPEN_PBR_ID), None)
    shader = opbr or next((c for c in shaders.values() if c.attrs.get("info:id") == "UsdPreviewSurface"), None)
    if opbr is not None:
        m = MaterialDesc.open_pbr(name=prim.path)
        a = opbr.attrs
        for key, idx, comps in OPEN_PBR_INPUTS:
            v = a.get(f"inputs:{key}")
            if v is not None:
                m.params[idx:idx + max(comps, 1)] = v
        lum = a.get("inputs:emission_luminance")
        col = a.get("inputs:emission_color")
        m.params[P_EMISSION:P_EMISSION + 3] = np.float32(0.0 if lum is None else lum) * np.asarray([1, 1, 1] if col is None else col, np.float32)
        m.klass = MAT_OPEN_PBR
    else:
        kw = {}
        if shader is not None:
            for key in ("diffuseColor", "emissiveColor", "specularColor"):
                if shader.attrs.get(f"inputs:{key}") is not None:
                    kw[key] = tuple(float(x) for x in shader.attrs[f"inputs:{key}"])
            for key in ("useSpecularWorkflow", "metallic", "roughness", "clearcoat", "clearcoatRoughness", "opacity",
                        "opacityThreshold", "ior"):
                if shader.attrs.get(f"inputs:{key}") is not None:
                    kw[key] = float(shader.attrs[f"inputs:{key}"])
        m = MaterialDesc.usd_preview_surface(name=prim.path, klass=klass, **kw)
    if shader is not None:
        for slot, (ups, opbr_name, _rtype, _vtype) in SLOT_INPUTS.items():
            if opbr is None and ups is None:
                continue
            conn = shader.attrs.get(f"inputs:{opbr_name if opbr is not None else ups}.connect")
            if isinstance(conn, tuple) and conn[0] == "path":
                reader = shaders.get(conn[1].split(".")[0].split("/")[-1])
                if reader is not None and str(reader.attrs.get("info:id", "")).startswith("UsdPrimvarReader"):
                    m.primvar_inputs[slot] = reader.attrs.get("inputs:varname")
    return m


def _quat_from_rows(m) -> tuple:
    """Rotation quaternion (x, y, z, w) of a row-vector rotation matrix (GfMatrix4d::ExtractRotationQuat)."""
    r = np.asarray(m, np.float64)[:3, :3].T  # column-vector form
    w = math.sqrt(max(0.0, 1.0 + r[0, 0] + r[1, 1] + r[2, 2])) * 0.5
    if w > 1e-6:
        return ((r[2, 1] - r[1, 2]) / (4 * w), (r[0, 2] - r[2, 0]) / (4 * w), (r[1, 0] - r[0, 1]) / (4 * w), w)
    k = int(np.argmax([r[0, 0], r[1, 1], r[2, 2]]))
    i, j = (k + 1) % 3, (k + 2) % 3
    q = [0.0, 0.0, 0.0, 0.0]
    q[k] = math.sqrt(max(0.0, 1.0 + r[k, k] - r[i, i] - r[j, j])) * 0.5
    q[i] = (r[i, k] + r[k, i]) / (4 * q[k]); q[j] = (r[j, k] + r[k, j]) / (4 * q[k]); q[3] = (r[j, i] - r[i, j]) / (4 * q[k])
    return tuple(q)


def _light_from_prim(scene: SceneDesc, prim: Prim, world: np.ndarray, base_dir: str):
    """hdGatling's light sync (light.cpp:58-94 base emission; :110-135 sphere, :150-180 distant, :216-262 rect, :279-320 disk,
    :336-400 dome), without colour temperature."""
    from .scene import DiskLight, DistantLight, DomeLight, RectLight, SphereLight
    a = prim.attrs
    w = world.astype(np.float32).astype(np.float64)
    tdir = lambda v: np.asarray(v, np.float64) @ w[:3, :3]
    origin = tuple(np.float32(w[3, :3]))
    intensity = float(a.get("inputs:intensity", 1.0) or 0.0) * 2.0 ** float(a.get("inputs:exposure", 0.0) or 0.0)
    color = np.asarray(a.get("inputs:color") or (1, 1, 1), np.float32)
    normalize = bool(float(a.get("inputs:normalize", 0) or 0))
    diffuse, specular = float(a.get("inputs:diffuse", 1.0)), float(a.get("inputs:specular", 1.0))
    em = lambda factor: tuple(color * np.float32(intensity / (factor if (normalize and factor > 0) else 1.0)))
    unit = lambda v: tuple(np.float32(v / np.linalg.norm(v)))
    if prim.type == "SphereLight":
        r = float(a.get("inputs:radius", 0.5))
        rx, ry, rz = tdir((r, 0, 0))[0], tdir((0, r, 0))[1], tdir((0, 0, r))[2]
        area = (((rx * ry) ** 1.6 + (rx * rz) ** 1.6 + (ry * rz) ** 1.6) / 3.0) ** (1 / 1.6) * 4.0 * math.pi
        scene.sphere_lights.append(SphereLight(origin, em(area), (float(rx), float(ry), float(rz)), diffuse, specular))
    elif prim.type == "DistantLight":
        angle = math.radians(float(a.get("inputs:angle", 0.53)))
        nm = np.linalg.inv(w[:3, :3]).T
        d = np.asarray((0.0, 0.0, -1.0)) @ nm
        s = math.sin(angle * 0.5)
        scene.distant_lights.append(DistantLight(unit(d), em(s * s * math.pi if s > 1e-6 else 0.0), angle, diffuse, specular))
    elif prim.type in ("RectLight", "DiskLight"):
        t0, t1 = unit(tdir((1, 0, 0))), unit(tdir((0, 1, 0)))
        if prim.type == "RectLight":
            wd = tdir((float(a.get("inputs:width", 1.0)), 0, 0))[0]
            ht = tdir((0, float(a.get("inputs:height", 1.0)), 0))[1]
            scene.rect_lights.append(RectLight(origin, t0, t1, em(wd * ht), float(wd), float(ht), diffuse, specular))
        else:
            r = float(a.get("inputs:radius", 0.5))
            rx, ry = tdir((r, 0, 0))[0], tdir((0, r, 0))[1]
            scene.disk_lights.append(DiskLight(origin, t0, t1, em(rx * ry * math.pi), float(rx), float(ry), diffuse, specular))
    elif prim.type == "DomeLight":
        tex = -1
        f = a.get("inputs:texture:file")
        if isinstance(f, tuple) and f[0] == "asset":
            from .imageio import read_hdr
            px = read_hdr(os.path.join(base_dir, f[1]))
            if px is not None:
                tex = len(scene.textures)
                scene.textures.append(px)
        q = _quat_from_rows(w)
        scene.dome_light = DomeLight(tex, (float(q[0]), float(q[1]), float(q[2]), float(-q[3])), em(0.0), diffuse, specular)


_PRIMVAR_TYPES = {"float": 0, "float2": 1, "float3": 2, "float4": 3, "color3f": 2, "texCoord2f": 1, "normal3f": 2, "vector3f": 2, "point3f": 2}
_PRIMVAR_INTERP = {"constant": 0, "uniform": 2, "vertex": 3, "varying": 3}


def load_usda(path: str, material_class: int = MAT_USD_PREVIEW_SURFACE) -> SceneDesc:
    """Builds the SceneDesc that hdGatling would feed through the gi boundary for this stage."""
    from .scene import Primvar
    with open(path, "r") as f:
        root = parse_usda(f.read())
    base_dir = os.path.dirname(os.path.abspath(path))
    scene = SceneDesc()
    materials: Dict[str, int] = {}
    pending = []          # (mesh prim, prototype-to-world, [instance transforms] or None)
    prototypes = {}       # class prim path -> [(mesh prim, local transform inside the prototype)]
    instances = {}        # class prim path -> [instance world transforms], in stage order
    order = []            # first-use order of prototypes among the pending meshes
    camera = [None]

    def collect_proto(prim: Prim, local: np.ndarray, out: list):
        m = local
        if prim.attrs.get("xformOp:transform") is not None:
            m = _matrix(prim.attrs["xformOp:transform"]) @ local
        if prim.type == "Mesh":
            out.append((prim, m))
        for c in prim.children:
            collect_proto(c, m, out)

    def walk(prim: Prim, world: np.ndarray):
        if prim.specifier == "class":
            lst = []
            for c in prim.children:
                collect_proto(c, np.eye(4), lst)
            prototypes[prim.path] = lst
            return
        local = world
        if "xformOp:transform" in prim.attrs and prim.attrs["xformOp:transform"] is not None:
            local = _matrix(prim.attrs["xformOp:transform"]) @ world  # row vectors: p * M_child * M_parent
        inh = prim.meta.get("inherits")
        if isinstance(inh, tuple) and inh[0] == "path":  # an instance of a class prototype (native instancing)
            if inh[1] not in instances:
                instances[inh[1]] = []
                order.append(("proto", inh[1]))
            instances[inh[1]].append(local)
            return
        if prim.type == "Material":
            materials[prim.path] = len(scene.materials)
            scene.materials.append(_material_from_prim(prim, material_class))
        elif prim.type == "Camera" and camera[0] is None:
            camera[0] = camera_from_prim(prim, local)
        elif prim.type == "Mesh":
            order.append(("mesh", len(pending)))
            pending.append((prim, local))
        elif prim.type in ("SphereLight", "DistantLight", "RectLight", "DiskLight", "DomeLight"):
            _light_from_prim(scene, prim, local, base_dir)
        for c in prim.children:
            walk(c, local)

    walk(root, np.eye(4))
    if camera[0] is not None:
        scene.camera = camera[0]
    todo = []
    for kind, key in order:
        if kind == "mesh":
            todo.append((pending[key][0], pending[key][1], None))
        else:
            for prim, local in prototypes.get(key, []):
                todo.append((prim, local, instances[key]))
    for mesh_id, (prim, world, inst) in enumerate(todo):
        a = prim.attrs
        nrm = a.get("normals")
        interp = prim.attr_meta.get("normals", {}).get("interpolation", "vertex")
        left = a.get("orientation", "rightHanded") == "leftHanded"
        st = a.get("primvars:st") if prim.attr_meta.get("primvars:st", {}).get("interpolation", "vertex") == "vertex" else None
        verts, faces = build_mesh_arrays(a["points"], a["faceVertexCounts"], a["faceVertexIndices"],
                                         normals=nrm, normals_interpolation=interp, left_handed=left, texcoords=st)
        binding = a.get("material:binding")
        mat = materials.get(binding[1], -1) if isinstance(binding, tuple) else -1
        if mat < 0:
            if "__default__" not in materials:
                materials["__default__"] = len(scene.materials)
                scene.materials.append(MaterialDesc.usd_preview_surface(name="__default__", klass=material_class))
            mat = materials["__default__"]
        primvars = []
        for key, val in a.items():
            if key.startswith("primvars:") and key != "primvars:st" and val is not None:
                meta = prim.attr_meta.get(key, {})
                data = np.asarray(val, np.float32)
                ptype = data.shape[1] - 1 if data.ndim == 2 else 0
                primvars.append(Primvar(key[len("primvars:"):], ptype, _PRIMVAR_INTERP.get(meta.get("interpolation", "constant"), 0), data))
        mesh = MeshDesc(name=prim.path, vertices=verts, faces=faces, material=mat, id=mesh_id,
                        double_sided=bool(float(a.get("doubleSided", 0) or 0)), left_handed=left,
                        visible=a.get("visibility", "inherited") != "invisible", transform=world.astype(np.float32), primvars=primvars)
        if inst is not None:
            mesh.instance_transforms = np.stack(inst).astype(np.float32)
            mesh.instance_ids = np.arange(len(inst), dtype=np.int32)
        scene.meshes.append(mesh)
    return scene
