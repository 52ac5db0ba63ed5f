"""MaterialX documents for :class:`MaterialDesc` parameter blocks -- the inverse of the gtl shim's MaterialX reader
(``gatling_amd/csrc/gtl_shim.cpp: descFromMtlx``), used by the parity tests of that reader (tests/test_mtlx_parity.py): a document written here and
read back through ``gtl::giCreateMaterialFromMtlxStr`` must produce the parameter block it was written from, bit for bit, and therefore the same image.

Two spellings of the same material, both of which hdGatling can hand over (src/hdGatling/materialNetworkCompiler.cpp:667-686):

``direct``     ``<open_pbr_surface>`` / ``<UsdPreviewSurface>`` with constant ``<input name value>`` children;
``nodegraph``  what ``HdMtlxCreateMtlxDocumentFromHdNetwork`` emits for a network whose inputs hang on upstream nodes: every input is connected
               (``nodegraph=`` + ``output=``) to an ``<output>`` of a ``<nodegraph>``, which names the ``<constant>`` node carrying the value
               (hdGatling's colour / float mismatch patchers insert exactly such constants), followed by a ``<surfacematerial>``.

Floats are written with 9 significant digits, which round-trips every float32 through ``strtof``.
"""
from __future__ import annotations

import numpy as np

from . import scene as S


def _f(v) -> str:
    return "%.9g" % float(np.float32(v))


def _vals(p, idx, n):
    return ", ".join(_f(p[idx + k]) for k in range(n))


def _inputs(m: S.MaterialDesc):
    """(category, [(input name, MaterialX type, value string)]) for every input the shim reads."""
    p = np.asarray(m.params, np.float32)
    if m.klass == S.MAT_USD_PREVIEW_SURFACE:
        return "UsdPreviewSurface", [
            ("diffuseColor", "color3", _vals(p, S.P_BASE_COLOR, 3)), ("emissiveColor", "color3", _vals(p, S.P_EMISSION, 3)),
            ("useSpecularWorkflow", "integer", _f(p[S.P_USE_SPECULAR_WORKFLOW])), ("specularColor", "color3", _vals(p, S.P_SPECULAR_COLOR, 3)),
            ("metallic", "float", _f(p[S.P_METALLIC])), ("roughness", "float", _f(p[S.P_ROUGHNESS])), ("clearcoat", "float", _f(p[S.P_CLEARCOAT])),
            ("clearcoatRoughness", "float", _f(p[S.P_CLEARCOAT_ROUGHNESS])), ("opacity", "float", _f(p[S.P_OPACITY])),
            ("opacityThreshold", "float", _f(p[S.P_OPACITY_THRESHOLD])), ("ior", "float", _f(p[S.P_IOR]))]
    if m.klass != S.MAT_OPEN_PBR:
        raise ValueError("only UsdPreviewSurface and open_pbr_surface have a MaterialX spelling")
    em = p[S.P_EMISSION:S.P_EMISSION + 3]
    return "open_pbr_surface", [
        ("base_weight", "float", _f(p[S.P_BASE_WEIGHT])), ("base_color", "color3", _vals(p, S.P_BASE_COLOR, 3)),
        ("base_diffuse_roughness", "float", _f(p[S.P_DIFFUSE_ROUGHNESS])), ("base_metalness", "float", _f(p[S.P_METALLIC])),
        ("specular_weight", "float", _f(p[S.P_SPECULAR_WEIGHT])), ("specular_color", "color3", _vals(p, S.P_SPECULAR_COLOR, 3)),
        ("specular_roughness", "float", _f(p[S.P_ROUGHNESS])), ("specular_ior", "float", _f(p[S.P_IOR])),
        ("transmission_weight", "float", _f(p[S.P_TRANSMISSION_WEIGHT])), ("transmission_color", "color3", _vals(p, S.P_TRANSMISSION_COLOR, 3)),
        ("transmission_depth", "float", _f(p[S.P_TRANSMISSION_DEPTH])), ("transmission_scatter", "color3", _vals(p, S.P_TRANSMISSION_SCATTER, 3)),
        ("transmission_scatter_anisotropy", "float", _f(p[S.P_TRANSMISSION_SCATTER_ANISOTROPY])),
        ("coat_weight", "float", _f(p[S.P_CLEARCOAT])), ("coat_color", "color3", _vals(p, S.P_COAT_COLOR, 3)),
        ("coat_roughness", "float", _f(p[S.P_CLEARCOAT_ROUGHNESS])), ("coat_ior", "float", _f(p[S.P_COAT_IOR])), ("coat_darkening", "float", _f(p[S.P_COAT_DARKENING])),
        ("fuzz_weight", "float", _f(p[S.P_FUZZ_WEIGHT])), ("fuzz_color", "color3", _vals(p, S.P_FUZZ_COLOR, 3)), ("fuzz_roughness", "float", _f(p[S.P_FUZZ_ROUGHNESS])),
        ("geometry_thin_walled", "boolean", "true" if p[S.P_THIN_WALLED] != 0.0 else "false"),
        ("subsurface_weight", "float", _f(p[S.P_SUBSURFACE_WEIGHT])), ("subsurface_color", "color3", _vals(p, S.P_SUBSURFACE_COLOR, 3)),
        ("subsurface_scatter_anisotropy", "float", _f(p[S.P_SUBSURFACE_ANISOTROPY])),
        ("subsurface_radius", "float", _f(p[S.P_SUBSURFACE_RADIUS])), ("subsurface_radius_scale", "color3", _vals(p, S.P_SUBSURFACE_RADIUS_SCALE, 3)),
        ("specular_roughness_anisotropy", "float", _f(p[S.P_SPECULAR_ANISOTROPY])), ("coat_roughness_anisotropy", "float", _f(p[S.P_COAT_ANISOTROPY])),
        # [ext] the turn of the coat's tangent as a plain float on the surface node (round-trips every float32); material_to_mtlx(coat_tangent="rotate3d")
        # writes the spelling documents use for geometry_coat_tangent instead
        ("coat_rotation", "float", _f(p[S.P_COAT_ROTATION])), ("specular_rotation", "float", _f(p[S.P_SPECULAR_ROTATION])),
        ("thin_film_weight", "float", _f(p[S.P_THIN_FILM_WEIGHT])), ("thin_film_thickness", "float", _f(p[S.P_THIN_FILM_THICKNESS])), ("thin_film_ior", "float", _f(p[S.P_THIN_FILM_IOR])),
        # the parameter block keeps luminance x colour: luminance 1 and the product as the colour reproduce it exactly
        ("emission_luminance", "float", "1" if em.any() else "0"), ("emission_color", "color3", _vals(p, S.P_EMISSION, 3) if em.any() else "1, 1, 1"),
        ("geometry_opacity", "float", _f(p[S.P_OPACITY]))]


def material_to_mtlx(m: S.MaterialDesc, form: str = "direct", coat_tangent: str = "ext") -> str:
    """``coat_tangent="rotate3d"`` spells the coat tangent's turn the way documents feed ``geometry_coat_tangent`` (open_pbr_surface.mtlx:91, 561):
    ``<rotate3d in=<tangent> amount=degrees axis=<normal>>`` behind a ``<normalize>`` (the shim divides the degrees by 360, so only turns whose degrees are
    exact round-trip bit for bit), and ``geometry_tangent`` (:89) likewise; the default writes the [ext] ``coat_rotation`` / ``specular_rotation`` floats."""
    cat, inputs = _inputs(m)
    extra, tail = "", ""
    if coat_tangent == "rotate3d" and cat == "open_pbr_surface":
        inputs = [i for i in inputs if i[0] not in ("coat_rotation", "specular_rotation")]
        extra = (f'<tangent name="T_{m.name}" type="vector3"><input name="space" type="string" value="world" /></tangent>'
                 f'<normal name="N_{m.name}" type="vector3"><input name="space" type="string" value="world" /></normal>')
        for tag, idx, inp in (("Coat", S.P_COAT_ROTATION, "geometry_coat_tangent"), ("Spec", S.P_SPECULAR_ROTATION, "geometry_tangent")):
            deg = _f(np.float32(m.params[idx]) * np.float32(360.0))
            extra += (f'<rotate3d name="{tag}R_{m.name}" type="vector3"><input name="in" type="vector3" nodename="T_{m.name}" />'
                      f'<input name="amount" type="float" value="{deg}" /><input name="axis" type="vector3" nodename="N_{m.name}" /></rotate3d>'
                      f'<normalize name="{tag}T_{m.name}" type="vector3"><input name="in" type="vector3" nodename="{tag}R_{m.name}" /></normalize>')
            tail += f'<input name="{inp}" type="vector3" nodename="{tag}T_{m.name}" />'
    elif coat_tangent != "ext":
        raise ValueError(coat_tangent)
    if form == "direct":
        body = "".join(f'<input name="{n}" type="{t}" value="{v}" />' for n, t, v in inputs)
        return f'<?xml version="1.0"?><materialx version="1.38">{extra}<{cat} name="SR_{m.name}" type="surfaceshader">{body}{tail}</{cat}></materialx>'
    if form != "nodegraph":
        raise ValueError(form)
    ng, conn = [], []
    for i, (n, t, v) in enumerate(inputs):
        ct = "float" if t in ("integer", "boolean") else t  # (the patchers turn int / bool constants into floats; the shim parses true / false as well)
        ng.append(f'<constant name="c{i}" type="{ct}"><input name="value" type="{ct}" value="{v}" /></constant><output name="out_{n}" type="{ct}" nodename="c{i}" />')
        conn.append(f'<input name="{n}" type="{t}" nodegraph="NG_{m.name}" output="out_{n}" />')
    return (f'<?xml version="1.0"?><materialx version="1.38"><nodegraph name="NG_{m.name}">{"".join(ng)}</nodegraph>{extra}'
            f'<{cat} name="SR_{m.name}" type="surfaceshader">{"".join(conn)}{tail}</{cat}>'
            f'<surfacematerial name="{m.name}" type="material"><input name="surfaceshader" type="surfaceshader" nodename="SR_{m.name}" /></surfacematerial></materialx>')
