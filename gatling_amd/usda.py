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
        self.next()  # def
        k, v = self.next()
        typ = ""
        if k == "id":
            typ = v
            k, v = self.next()
        name = v[1:-1]
        p = Prim(typ, name, f"{parent_path}/{name}")
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
    shader = next((c for c in prim.children if c.type == "Shader" and c.attrs.get("info:id") == "UsdPreviewSurface"), None)
    kw = {}
    if shader is not None:
        for key in ("diffuseColor", "emissiveColor", "specularColor"):
            if shader.attrs.get(f"inputs:{key}") is not None:
                kw[key] = tuple(float(x) for x in shader.attrs[f"inputs:{key}"])
        for key in ("useSpecularWorkflow", "metallic", "roughness", "clearcoat", "clearcoatRoughness", "opacity",
                    "opacityThreshold", "ior"):
            if shader.attrs.get(f"inputs:{key}") is not None:
                kw[key] = float(shader.attrs[f"inputs:{key}"])
    return MaterialDesc.usd_preview_surface(name=prim.path, klass=klass, **kw)


def load_usda(path: str, material_class: int = MAT_USD_PREVIEW_SURFACE) -> SceneDesc:
    """Builds the SceneDesc that hdGatling would feed through the gi boundary for this stage."""
    with open(path, "r") as f:
        root = parse_usda(f.read())
    scene = SceneDesc()
    materials: Dict[str, int] = {}
    pending = []
    camera = [None]

    def walk(prim: Prim, world: np.ndarray):
        local = world
        if "xformOp:transform" in prim.attrs and prim.attrs["xformOp:transform"] is not None:
            local = _matrix(prim.attrs["xformOp:transform"]) @ world  # row vectors: p * M_child * M_parent
        if prim.type == "Material":
            materials[prim.path] = len(scene.materials)
            scene.materials.append(_material_from_prim(prim, material_class))
        elif prim.type == "Camera" and camera[0] is None:
            camera[0] = camera_from_prim(prim, local)
        elif prim.type == "Mesh":
            pending.append((prim, local))
        for c in prim.children:
            walk(c, local)

    walk(root, np.eye(4))
    if camera[0] is not None:
        scene.camera = camera[0]
    for mesh_id, (prim, world) in enumerate(pending):
        a = prim.attrs
        nrm = a.get("normals")
        interp = prim.attr_meta.get("normals", {}).get("interpolation", "vertex")
        left = a.get("orientation", "rightHanded") == "leftHanded"
        verts, faces = build_mesh_arrays(a["points"], a["faceVertexCounts"], a["faceVertexIndices"],
                                         normals=nrm, normals_interpolation=interp, left_handed=left)
        binding = a.get("material:binding")
        mat = materials.get(binding[1], -1) if isinstance(binding, tuple) else -1
        if mat < 0:
            if "__default__" not in materials:
                materials["__default__"] = len(scene.materials)
                scene.materials.append(MaterialDesc.usd_preview_surface(name="__default__", klass=material_class))
            mat = materials["__default__"]
        scene.meshes.append(MeshDesc(name=prim.path, vertices=verts, faces=faces, material=mat, id=mesh_id,
                                     double_sided=bool(float(a.get("doubleSided", 0) or 0)), left_handed=left,
                                     transform=world.astype(np.float32)))
    return scene
