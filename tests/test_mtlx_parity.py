"""Parity of the MaterialX front end (VERDICT r02 "what's weak" #1: "a MaterialX document and the equivalent parameter block are never
shown to give the same image").  hdGatling hands every UsdPreviewSurface / MaterialX network over as a document (src/hdGatling/materialNetworkCompiler.cpp:667-686
-> giCreateMaterialFromMtlxDoc / ...Str); the shim's reader (gatling_amd/csrc/gtl_shim.cpp descFromMtlx) turns it into the closed-form parameter block.

CPU: for C4's 32 parameter sets (BASELINE.json configs[3]) and a set of edge cases, the document written by gatling_amd/mtlx_writer.py -- in the direct spelling and
in the nodegraph spelling HdMtlxCreateMtlxDocumentFromHdNetwork emits -- reads back to the SAME 64-float block, bit for bit; unknown shading models are refused
(nullptr -> hdGatling falls back to its default material, materialNetworkCompiler.cpp:619-633).
GPU: the scene whose materials were created from documents through gtl::giCreateMaterialFromMtlxStr renders the bit-identical image to the parameter-block scene and
to the oracle."""
import ctypes as C

import numpy as np
import pytest

from gatling_amd import capi
from gatling_amd import scene as S
from gatling_amd.mtlx_writer import material_to_mtlx
from gatling_amd.scenes import _parameter_sets, sphere_grid


def _edge_cases():
    M = S.MaterialDesc
    return [M.open_pbr(name="emit", emission_luminance=3.5, emission_color=(1.0, 0.45, 0.2), base_color=(0.1, 0.2, 0.3)),
            M.open_pbr(name="glass", transmission_weight=1.0, transmission_color=(0.9, 0.95, 0.7), transmission_depth=0.25, transmission_scatter=(0.1, 0.2, 0.3),
                       transmission_scatter_anisotropy=-0.35, specular_ior=1.33, specular_roughness=0.07),
            M.open_pbr(name="thin", geometry_thin_walled=True, transmission_weight=0.5, base_diffuse_roughness=0.6, specular_weight=0.4, specular_color=(0.9, 0.8, 0.7)),
            M.open_pbr(name="coat", coat_weight=0.8, coat_color=(0.9, 0.3, 0.2), coat_roughness=0.12, coat_ior=1.45, coat_darkening=0.35, base_weight=0.75, base_metalness=1.0),
            M.open_pbr(name="fuzz", fuzz_weight=0.5, fuzz_color=(0.2, 0.3, 0.4), fuzz_roughness=0.8),
            M.open_pbr(name="soap", transmission_weight=0.8, specular_roughness=0.05, thin_film_weight=1.0, thin_film_thickness=0.3, thin_film_ior=1.33),
            M.open_pbr(name="brushed", base_metalness=1.0, specular_roughness=0.35, specular_roughness_anisotropy=0.7, coat_weight=0.4, coat_roughness=0.2, coat_roughness_anisotropy=0.25),
            M.usd_preview_surface(name="spec", useSpecularWorkflow=1, specularColor=(0.3, 0.4, 0.5), diffuseColor=(0.6, 0.1, 0.05), roughness=0.23, ior=1.7),
            M.usd_preview_surface(name="cut", opacity=0.37, opacityThreshold=0.5, emissiveColor=(0.5, 1.5, 2.5), clearcoat=0.7, clearcoatRoughness=0.2, metallic=0.33)]


def _c4_sets():
    return _parameter_sets(np.random.default_rng(4321), 32)  # the generator call of scenes.sphere_grid (config C4)


def _desc_from_doc(L, xml):
    L.gtlMaterialDescFromMtlxStrC.restype = C.c_int
    L.gtlMaterialDescFromMtlxStrC.argtypes = [C.c_char_p, C.POINTER(capi.GiCMaterialDesc)]
    d = capi.GiCMaterialDesc()
    rc = L.gtlMaterialDescFromMtlxStrC(xml.encode(), C.byref(d))
    return (d if rc == capi.GI_C_OK else None)


@pytest.mark.parametrize("form", ["direct", "nodegraph"])
def test_documents_read_back_to_the_parameter_block(form):
    L = capi.load_library()
    sets = _c4_sets() + _edge_cases()
    assert {m.klass for m in sets} == {S.MAT_OPEN_PBR, S.MAT_USD_PREVIEW_SURFACE}
    for m in sets:
        d = _desc_from_doc(L, material_to_mtlx(m, form))
        assert d is not None, (m.name, form)
        assert d.klass == m.klass and d.flags == 0
        got, want = np.frombuffer(bytes(d.p), np.uint32), np.asarray(m.params, np.float32).view(np.uint32)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (m.name, form, bad.tolist(), np.frombuffer(bytes(d.p), np.float32)[bad].tolist(), np.asarray(m.params)[bad].tolist())


def test_defaults_and_refusals():
    L = capi.load_library()
    # an element without inputs takes the specification's defaults (open_pbr_surface.mtlx:11-92; UsdPreviewSurface spec)
    d = _desc_from_doc(L, '<materialx version="1.39"><open_pbr_surface name="m" type="surfaceshader"/></materialx>')
    assert np.array_equal(np.frombuffer(bytes(d.p), np.float32), S.MaterialDesc.open_pbr().params)
    d = _desc_from_doc(L, '<materialx version="1.38"><UsdPreviewSurface name="m" type="surfaceshader"></UsdPreviewSurface></materialx>')
    assert np.array_equal(np.frombuffer(bytes(d.p), np.float32), S.MaterialDesc.usd_preview_surface().params)
    # shading models the closed forms do not cover are refused: the caller (hdGatling) then uses its fallback material
    for xml in ('<materialx version="1.38"><standard_surface name="s" type="surfaceshader"><input name="base" type="float" value="1"/></standard_surface></materialx>',
                '<materialx version="1.38"><gltf_pbr name="g" type="surfaceshader"/></materialx>', "<materialx/>", "", "not xml at all",
                '<materialx><UsdPreviewSurfaceX name="near_miss"/></materialx>'):
        assert _desc_from_doc(L, xml) is None, xml
    assert L.gtlMaterialDescFromMtlxStrC(None, None) != capi.GI_C_OK


@pytest.mark.gpu
def test_document_materials_render_the_parameter_block_image(gi):
    """C4's 32 parameter sets on a 6 x 6 grid of instanced icospheres (the scene generator of config C4 at test size)."""
    from oracle import orc
    desc = sphere_grid(6, 2, 32)
    assert [m.params.tobytes() for m in desc.materials] == [m.params.tobytes() for m in _c4_sets()]
    rs = S.RenderSettings(spp=4, max_bounces=6)
    w, h = 96, 54
    ref, cnt = orc.render(desc, rs, w, h, threads=8)
    imgs = {}
    for form in (None, "direct", "nodegraph"):
        docs = {i: material_to_mtlx(m, form) for i, m in enumerate(desc.materials)} if form else None
        sc = capi.Scene(desc, mtlx_materials=docs)
        try:
            imgs[form] = sc.render(rs, w, h).copy()
            assert sc.stats()["segments"] == cnt["segments"]
        finally:
            sc.close()
    for form, img in imgs.items():
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"materials from {form or 'parameter blocks'}: image differs from the oracle"


@pytest.mark.gpu
def test_document_edge_cases_render_the_parameter_block_image(gi):
    """Emission, transmission with a medium, thin walls, coat darkening, the specular workflow, cutout opacity: one sphere per edge case."""
    desc = sphere_grid(3, 2, 7)
    desc.materials = _edge_cases()
    rs = S.RenderSettings(spp=3, max_bounces=6, medium_stack_size=2)
    a = capi.Scene(desc); b = capi.Scene(desc, mtlx_materials={i: material_to_mtlx(m, "nodegraph" if i % 2 else "direct") for i, m in enumerate(desc.materials)})
    try:
        ia, ib = a.render(rs, 64, 36).copy(), b.render(rs, 64, 36).copy()
    finally:
        a.close(); b.close()
    assert np.array_equal(ia.view(np.uint32), ib.view(np.uint32))
    assert np.isfinite(ia).all() and ia[..., :3].max() > 0.0
