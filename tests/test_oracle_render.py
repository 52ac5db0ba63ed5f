"""Oracle-level tests (CPU): golden fixtures + size-independent properties of the restated render loop."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden import CASES, build_case  # noqa: E402

from gatling_amd.scene import MAT_DIFFUSE, MaterialDesc, MeshDesc, RenderSettings, SceneDesc, CameraDesc, RectLight
from gatling_amd.meshprep import build_mesh_arrays
from gatling_amd.scenes import cornell_box

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(orc, name):
    desc, rs, w, h = build_case(name)
    img, cnt = orc.render(desc, rs, w, h, threads=2)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert cnt["segments"] == int(g["segments"]) and cnt["shadow_rays"] == int(g["shadow_rays"])
    assert np.array_equal(img.view(np.uint32), g["color"].view(np.uint32))  # bit-exact regression anchor


def test_row_split_is_bit_identical(orc):
    """Sharding by pixel rows must not change a single bit (global pixel index seeds the RNG)."""
    desc = cornell_box()
    rs = RenderSettings(spp=4, max_bounces=6)
    full, _ = orc.render(desc, rs, 40, 24)
    a, _ = orc.render(desc, rs, 40, 24, rows=(0, 11))
    b, _ = orc.render(desc, rs, 40, 24, rows=(11, 24))
    assert np.array_equal(np.concatenate([a, b]).view(np.uint32), full.view(np.uint32))


def test_thread_count_is_irrelevant(orc):
    desc = cornell_box()
    rs = RenderSettings(spp=3, max_bounces=5)
    a, ca = orc.render(desc, rs, 32, 18, threads=1)
    b, cb = orc.render(desc, rs, 32, 18, threads=5)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and ca == cb


def test_progressive_accumulation(orc):
    """rp_main.rgen:506-515: (prev*offset + new*spp) / (offset+spp); two calls of 4 spp ~ one call of 8 spp."""
    desc = cornell_box(MAT_DIFFUSE)
    rs4 = RenderSettings(spp=4, max_bounces=4)
    first, _ = orc.render(desc, rs4, 32, 18, sample_offset=0)
    second, _ = orc.render(desc, rs4, 32, 18, sample_offset=4, prev_color=first)
    full, _ = orc.render(desc, RenderSettings(spp=8, max_bounces=4), 32, 18)
    assert not np.array_equal(first, second)
    np.testing.assert_allclose(second, full, rtol=2e-6, atol=2e-6)  # same samples, different float association


def test_empty_scene_is_background(orc):
    """No geometry: every primary ray misses and returns the fallback dome = colour clear value quantised to RGBA8."""
    desc = SceneDesc(materials=[MaterialDesc.usd_preview_surface()], camera=CameraDesc(position=(0, 0, 5)))
    rs = RenderSettings(spp=2, max_bounces=3, clear_color=(0.5, 0.25, 1.0, 1.0))
    img, cnt = orc.render(desc, rs, 8, 4)
    exp = np.float32([int(0.5 * 255) / 255.0, int(0.25 * 255) / 255.0, 1.0, 1.0])
    assert cnt["segments"] == 8 * 4 * 2 and cnt["hits"] == 0
    np.testing.assert_allclose(img, np.broadcast_to(exp, img.shape), rtol=1e-6)


def test_max_bounces_bounds_segments(orc):
    desc = cornell_box()
    for b in (1, 2, 5):
        _, cnt = orc.render(desc, RenderSettings(spp=2, max_bounces=b), 24, 14)
        assert cnt["samples"] == 24 * 14 * 2
        assert cnt["samples"] <= cnt["segments"] <= cnt["samples"] * b
        assert sum(cnt["bounce_histogram"]) == cnt["segments"] and all(x == 0 for x in cnt["bounce_histogram"][b:])


def _furnace(albedo):
    """Closed diffuse box seen from inside, lit only by its own uniform emission."""
    pts = [(-1, -1, -1), (-1, -1, 1), (-1, 1, -1), (-1, 1, 1), (1, -1, -1), (1, -1, 1), (1, 1, -1), (1, 1, 1)]
    counts = [4] * 6
    idx = [0, 1, 3, 2, 2, 3, 7, 6, 6, 7, 5, 4, 4, 5, 1, 0, 2, 6, 4, 0, 7, 3, 1, 5]
    nrm = [(-1, 0, 0)] * 4 + [(0, 1, 0)] * 4 + [(1, 0, 0)] * 4 + [(0, -1, 0)] * 4 + [(0, 0, -1)] * 4 + [(0, 0, 1)] * 4
    v, f = build_mesh_arrays(pts, counts, idx, normals=nrm, normals_interpolation="faceVarying")  # flat shading
    mat = MaterialDesc.usd_preview_surface(diffuseColor=(albedo,) * 3, emissiveColor=(1, 1, 1), klass=MAT_DIFFUSE)
    return SceneDesc(meshes=[MeshDesc("box", v, f, 0)], materials=[mat], camera=CameraDesc(position=(0, 0, 0), forward=(0, 1, 0), up=(0, 0, 1)))


def test_furnace_energy(orc):
    """Inside a closed emitting Lambertian box, radiance = sum_k albedo^k over the traced bounces (no RR, no clamp)."""
    albedo, bounces = 0.5, 6
    rs = RenderSettings(spp=8, max_bounces=bounces, rr_bounce_offset=100, max_sample_value=1e9)
    img, cnt = orc.render(_furnace(albedo), rs, 16, 16)
    expected = sum(albedo ** k for k in range(bounces))
    assert cnt["hits"] == cnt["segments"]
    np.testing.assert_allclose(img[..., :3], expected, rtol=1e-5)


def test_nee_rect_light_direct_illumination(orc):
    """One-bounce NEE on a diffuse floor under a small rect light matches the analytic form-factor integral."""
    v, f = build_mesh_arrays([(-50, -50, 0), (50, -50, 0), (50, 50, 0), (-50, 50, 0)], [4], [0, 1, 2, 3])
    mat = MaterialDesc.usd_preview_surface(diffuseColor=(0.6, 0.6, 0.6), klass=MAT_DIFFUSE)
    desc = SceneDesc(meshes=[MeshDesc("floor", v, f, 0, double_sided=True)], materials=[mat],
                     camera=CameraDesc(position=(0, 0, 3), forward=(0, 0, -1), up=(0, 1, 0), vfov=0.02))
    # light faces -z when t0 x t1... lightNormal = cross(t1, t0): t0=(1,0,0), t1=(0,1,0) -> (0,0,-1)
    desc.rect_lights.append(RectLight(origin=(0, 0, 2), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(5, 5, 5), width=0.2, height=0.2))
    rs = RenderSettings(spp=256, max_bounces=1, next_event_estimation=True, clear_color=(0, 0, 0, 0), max_sample_value=1e9)
    img, cnt = orc.render(desc, rs, 4, 4, threads=4)
    # small-light limit at the point below the light: E = L * A * cos*cos / d^2 ; Lo = albedo/pi * E
    # Reference quirk kept on purpose: rp_main.chit:386 multiplies throughput by bsdf_over_pdf (= albedo) BEFORE the NEE
    # weight `throughput * neeRadiance` (:433) is formed, so direct light carries one extra albedo factor.
    expected = 0.6 * (0.6 / np.pi * 5.0 * (0.2 * 0.2) / (2.0 ** 2))
    assert cnt["shadow_rays"] > 0
    np.testing.assert_allclose(img[..., :3].mean(), expected, rtol=0.02)


def _frames(n, rng, cos_k1):
    nrm = np.tile(np.float32([0, 0, 1]), (n, 1)); tu = np.tile(np.float32([1, 0, 0]), (n, 1)); tv = np.tile(np.float32([0, 1, 0]), (n, 1))
    s = np.sqrt(1 - cos_k1 ** 2)
    k1 = np.tile(np.float32([s, 0, cos_k1]), (n, 1))
    k2 = rng.normal(size=(n, 3)); k2[:, 2] = np.abs(k2[:, 2]); k2 /= np.linalg.norm(k2, axis=1, keepdims=True)
    xi = rng.uniform(size=(n, 4)); xi[:, 3] = 0.0  # xi.w < 0.5 -> front face in the debug hook
    return np.concatenate([nrm, tu, tv, nrm, k1, k2, xi], axis=1).astype(np.float32)


def _closed_form_materials():
    from gatling_amd.scene import MAT_DIFFUSE
    M = MaterialDesc
    return [M.usd_preview_surface(diffuseColor=(1, 1, 1), klass=MAT_DIFFUSE),
            M.usd_preview_surface(diffuseColor=(1, 1, 1), roughness=0.5),
            M.usd_preview_surface(diffuseColor=(1, 1, 1), roughness=0.2, metallic=1.0),
            M.usd_preview_surface(diffuseColor=(1, 1, 1), roughness=0.6, clearcoat=1.0, clearcoatRoughness=0.1),
            M.open_pbr(base_color=(1, 1, 1)),
            M.open_pbr(base_color=(1, 1, 1), base_metalness=1.0, specular_roughness=0.4),
            M.open_pbr(base_color=(1, 1, 1), coat_weight=1.0, coat_roughness=0.1),
            M.open_pbr(base_color=(1, 1, 1), transmission_weight=1.0, specular_roughness=0.2),
            M.open_pbr(base_color=(1, 1, 1), base_diffuse_roughness=1.0, specular_weight=0.0),      # energy-preserving Oren-Nayar alone
            M.open_pbr(base_color=(1, 1, 1), base_diffuse_roughness=0.6, coat_weight=1.0, coat_roughness=0.4, coat_darkening=1.0),
            M.open_pbr(base_color=(1, 1, 1), transmission_weight=1.0, specular_roughness=0.3, geometry_thin_walled=True),
            M.open_pbr(base_color=(1, 1, 1), geometry_thin_walled=True, subsurface_weight=1.0, subsurface_color=(1, 1, 1), specular_weight=0.0),   # thin-walled subsurface alone
            M.open_pbr(base_color=(1, 1, 1), geometry_thin_walled=True, subsurface_weight=0.6, subsurface_color=(1, 1, 1), subsurface_scatter_anisotropy=0.4,
                       base_diffuse_roughness=0.5, coat_weight=0.5, coat_roughness=0.2),
            M.open_pbr(base_color=(0, 0, 0), specular_weight=0.0, fuzz_weight=1.0, fuzz_color=(1, 1, 1), fuzz_roughness=0.5),                      # the fuzz lobe alone
            M.open_pbr(base_color=(1, 1, 1), coat_weight=1.0, coat_roughness=0.1, fuzz_weight=1.0, fuzz_color=(1, 1, 1), fuzz_roughness=0.07),     # smooth fuzz (E > 1 at grazing views) over a coat
            M.open_pbr(base_color=(1, 1, 1), base_metalness=1.0, specular_roughness=0.4, fuzz_weight=0.6, fuzz_color=(1, 1, 1), fuzz_roughness=1.0),
            M.open_pbr(base_color=(1, 1, 1), base_metalness=1.0, specular_roughness=0.5, specular_roughness_anisotropy=0.8),                         # anisotropic metal
            M.open_pbr(base_color=(1, 1, 1), specular_roughness=0.4, specular_roughness_anisotropy=0.5, coat_weight=1.0, coat_roughness=0.3, coat_roughness_anisotropy=0.9),
            M.open_pbr(base_color=(1, 1, 1), specular_roughness=0.3, thin_film_weight=1.0, thin_film_thickness=0.3, thin_film_ior=1.9),           # film on a dielectric: reflection up, base down
            M.open_pbr(base_color=(1, 1, 1), base_metalness=1.0, specular_roughness=0.4, thin_film_weight=0.7, thin_film_thickness=0.55, thin_film_ior=1.33)]


def test_closed_form_bsdfs_conserve_energy(orc):
    """E[bsdf_over_pdf] over the sampling distribution is the directional albedo: <= 1 for white parameters."""
    rng = np.random.default_rng(21)
    for m in _closed_form_materials():
        for c in (0.95, 0.5, 0.15):
            out = orc.bsdf_debug(m, _frames(40000, rng, c))
            albedo = out[:, 3:6].mean(axis=0)
            assert np.all(albedo <= 1.03), (m.klass, c, albedo)
            assert np.all(albedo >= 0.0) and np.isfinite(out).all()


def test_closed_form_evaluate_matches_sampling(orc):
    """For reflection lobes: integrating evaluate() over the hemisphere (uniform directions) reproduces the mean
    sampled weight of the reflected events, and the evaluate pdf integrates to the probability of reflecting."""
    rng = np.random.default_rng(22)
    mats = _closed_form_materials()
    for m in mats[:7] + mats[8:10] + mats[11:]:  # the refraction lobes have no evaluate counterpart on the reflection side
        items = _frames(200000, rng, 0.7)
        out = orc.bsdf_debug(m, items)
        refl = (out[:, 7].astype(int) & 8) != 0
        sampled = (out[:, 3:6] * refl[:, None]).mean(axis=0)
        integ = (out[:, 8:11] + out[:, 11:14]).mean(axis=0) * (2 * np.pi)  # uniform hemisphere pdf = 1/(2 pi); bsdf*cos is returned
        np.testing.assert_allclose(integ, sampled, rtol=0.06, atol=0.01)
        np.testing.assert_allclose(out[:, 14].mean() * 2 * np.pi, refl.mean(), rtol=0.06, atol=0.01)


def test_fuzz_layer(orc):
    """open_pbr_surface.mtlx:569-581: sheen_bsdf(fuzz_weight, fuzz_color, fuzz_roughness) layered over the coat.  Our closed form (oracle/gi_oracle.cpp "fuzz
    (sheen) lobe"): Charlie distribution x Ashikhmin/Neubelt visibility, chosen with P = w * min(E, 1) where E is the tabulated directional albedo + 0.01.
    Checked against an INDEPENDENT numerical integration of the published formula (tools/gen_fuzz_albedo.py): the lobe's probability, its albedo, the
    reciprocity of its value, and that fuzz_weight = 0 is the material without fuzz, bit for bit."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("gen_fuzz_albedo", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_fuzz_albedo.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    rng = np.random.default_rng(29)
    tint = np.float32([0.9, 0.5, 0.25])
    for w, r, c in ((1.0, 0.5, 0.8), (0.7, 1.0, 0.3), (1.0, 0.2, 0.6), (0.5, 0.07, 0.1)):
        m = MaterialDesc.open_pbr(base_color=(0, 0, 0), specular_weight=0.0, fuzz_weight=w, fuzz_color=tuple(tint), fuzz_roughness=r)
        out = orc.bsdf_debug(m, _frames(200000, rng, c))
        chosen = out[:, 3] > 0.0                                  # black base, no specular: every event that carries light is the fuzz lobe
        E = gen.albedo(c, max(r, 0.07), 300)                       # the true directional albedo of D * V
        Pf = chosen.mean()
        assert w * min(E, 1.0) - 0.01 <= Pf <= w * min(E + 0.025, 1.0) + 0.01, (w, r, c, E, Pf)   # P = w * min(E_table + 0.01, 1), E_table within [E - 0.0093, E + ...]
        albedo = out[:, 3:6].mean(axis=0)                          # = tint * P * E / E_fit: never above tint * w, and the table's slack costs at most a few per cent
        assert np.all(albedo <= tint * w * min(E, 1.0) * 1.02 + 0.003), (albedo, E)
        assert np.all(albedo >= tint * w * min(E, 1.0) * (E / (E + 0.03)) * 0.97 - 0.003), (albedo, E)
        assert np.all(out[chosen, 5] <= out[chosen, 3] + 1e-6)     # tinted by fuzz_color
    # reciprocity of the lobe's value where E <= 1 on both sides: evaluate() returns f * cos(k2)
    m = MaterialDesc.open_pbr(base_color=(0, 0, 0), specular_weight=0.0, fuzz_weight=1.0, fuzz_color=(1, 1, 1), fuzz_roughness=0.8)
    n = 4000
    a = rng.normal(size=(n, 3)); a[:, 2] = np.abs(a[:, 2]) + 0.3; a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.normal(size=(n, 3)); b[:, 2] = np.abs(b[:, 2]) + 0.3; b /= np.linalg.norm(b, axis=1, keepdims=True)
    base = _frames(n, rng, 0.5)
    ab = base.copy(); ab[:, 12:15] = a; ab[:, 15:18] = b
    ba = base.copy(); ba[:, 12:15] = b; ba[:, 15:18] = a
    fab = orc.bsdf_debug(m, ab)[:, 11] / b[:, 2]; fba = orc.bsdf_debug(m, ba)[:, 11] / a[:, 2]
    np.testing.assert_allclose(fab, fba, rtol=2e-5, atol=1e-7)
    assert fab.min() > 0.0
    # fuzz_weight = 0: colour and roughness of the fuzz change nothing, bit for bit
    items = _frames(20000, rng, 0.4)
    plain = orc.bsdf_debug(MaterialDesc.open_pbr(base_color=(0.7, 0.6, 0.5), coat_weight=0.5), items)
    zero = orc.bsdf_debug(MaterialDesc.open_pbr(base_color=(0.7, 0.6, 0.5), coat_weight=0.5, fuzz_weight=0.0, fuzz_color=(0.2, 0.9, 0.1), fuzz_roughness=0.2), items)
    assert np.array_equal(plain.view(np.uint32), zero.view(np.uint32))
    # the layers beneath keep 1 - P of the light: a white diffuse base under a black fuzz loses exactly the fuzz's share
    m0 = MaterialDesc.open_pbr(base_color=(1, 1, 1), specular_weight=0.0)
    m1 = MaterialDesc.open_pbr(base_color=(1, 1, 1), specular_weight=0.0, fuzz_weight=1.0, fuzz_color=(0, 0, 0), fuzz_roughness=0.5)
    items = _frames(200000, rng, 0.6)
    a0 = orc.bsdf_debug(m0, items)[:, 3].mean(); a1 = orc.bsdf_debug(m1, items)[:, 3].mean()
    E = gen.albedo(0.6, 0.5, 300)
    assert abs(a1 - a0 * (1.0 - E)) < 0.02, (a0, a1, E)


def test_specular_anisotropy(orc):
    """open_pbr_surface.mtlx:27, 65, 133-136, 552-555 (open_pbr_anisotropy): alpha_t = r^2 sqrt(2 / (1 + (1 - a)^2)) along the tangent, alpha_b = (1 - a) alpha_t.
    The micro-normals of a GGX lobe with (alpha_t, alpha_b) have slope variances in that ratio; the lobe's value is reciprocal; anisotropy 0 is the isotropic lobe
    bit for bit; a tangent frame rotated by 90 degrees swaps the stretch."""
    rng = np.random.default_rng(31)
    r, a = 0.3, 0.75                                          # (small enough that almost no sampled micro-normal reflects below the horizon: the medians below are untruncated)
    at = r * r * np.sqrt(2.0 / (1.0 + (1.0 - a) ** 2)); ab = (1.0 - a) * at
    m = MaterialDesc.open_pbr(base_color=(1, 1, 1), base_metalness=1.0, specular_roughness=r, specular_roughness_anisotropy=a)
    items = _frames(200000, rng, 1.0)                        # normal incidence: the sampled half vector is the micro-normal distribution itself
    out = orc.bsdf_debug(m, items)
    ok = out[:, 7] != 0
    k2 = out[ok, 0:3]
    h = k2 + np.float32([0, 0, 1]); h /= np.linalg.norm(h, axis=1, keepdims=True)
    sx, sy = np.abs(h[:, 0] / h[:, 2]), np.abs(h[:, 1] / h[:, 2])        # slopes: GGX slopes are heavy-tailed, compare medians (median |slope_x| = alpha_x * const)
    np.testing.assert_allclose(np.median(sx) / np.median(sy), at / ab, rtol=0.05)
    np.testing.assert_allclose(np.median(sx), at * 0.5774, rtol=0.06)  # marginal of the GGX slope distribution: P(|s| < a / sqrt(3)) = 1/2
    # rotated frame: stretch follows the tangent
    rot = items.copy(); rot[:, 3:6] = items[:, 6:9]; rot[:, 6:9] = -items[:, 3:6]
    o2 = orc.bsdf_debug(m, rot); ok2 = o2[:, 7] != 0
    h2 = o2[ok2, 0:3] + np.float32([0, 0, 1]); h2 /= np.linalg.norm(h2, axis=1, keepdims=True)
    np.testing.assert_allclose(np.median(np.abs(h2[:, 1] / h2[:, 2])) / np.median(np.abs(h2[:, 0] / h2[:, 2])), at / ab, rtol=0.05)
    # reciprocity of evaluate / cos
    n = 4000
    u = rng.normal(size=(n, 3)); u[:, 2] = np.abs(u[:, 2]) + 0.2; u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = rng.normal(size=(n, 3)); v[:, 2] = np.abs(v[:, 2]) + 0.2; v /= np.linalg.norm(v, axis=1, keepdims=True)
    base = _frames(n, rng, 0.5)
    uv = base.copy(); uv[:, 12:15] = u; uv[:, 15:18] = v
    vu = base.copy(); vu[:, 12:15] = v; vu[:, 15:18] = u
    fuv = orc.bsdf_debug(m, uv)[:, 11] / v[:, 2]; fvu = orc.bsdf_debug(m, vu)[:, 11] / u[:, 2]
    np.testing.assert_allclose(fuv, fvu, rtol=3e-4, atol=1e-6)  # (the F82 Fresnel depends on k.h only: symmetric)
    # anisotropy 0 == no anisotropy input at all, bit for bit
    items = _frames(20000, rng, 0.4)
    a0 = orc.bsdf_debug(MaterialDesc.open_pbr(base_color=(0.7, 0.6, 0.5), coat_weight=0.5, coat_roughness=0.2), items)
    a1 = orc.bsdf_debug(MaterialDesc.open_pbr(base_color=(0.7, 0.6, 0.5), coat_weight=0.5, coat_roughness=0.2, specular_roughness_anisotropy=0.0, coat_roughness_anisotropy=0.0), items)
    assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32))


def test_coat_tangent(orc):
    """open_pbr_surface.mtlx:91, 561 (geometry_coat_tangent feeds the coat's dielectric_bsdf a tangent of its own), in the form documents bind it:
    the geometry tangent turned by coat_rotation turns towards the bitangent.  The stretch of an anisotropic coat's micro-normals follows the turned axes; a
    quarter turn is the frame with tangent and bitangent exchanged; half a turn is no turn (GGX is even); without an anisotropic coat the input is not read
    (bit for bit); evaluate stays reciprocal and consistent with sampling; nothing beneath the coat moves."""
    rng = np.random.default_rng(61)
    r, a = 0.3, 0.75
    at = r * r * np.sqrt(2.0 / (1.0 + (1.0 - a) ** 2)); ab = (1.0 - a) * at
    # only the coat reflects glossily: no specular reflection beneath it (specular_weight 0), a dense coat so that a quarter of the samples take it
    def mat(rot, **kw):
        d = dict(base_color=(0.5, 0.5, 0.5), specular_weight=0.0, coat_weight=1.0, coat_ior=3.0, coat_roughness=r, coat_roughness_anisotropy=a, coat_rotation=rot)
        d.update(kw)
        return MaterialDesc.open_pbr(**d)
    items = _frames(400000, rng, 1.0)
    for turns in (0.0, 0.125, 0.3):
        out = orc.bsdf_debug(mat(turns), items)
        coat = out[:, 7].astype(int) == (2 | 8)             # EV_GLOSSY | EV_REFLECTION
        assert 0.15 < coat.mean() < 0.35
        h = out[coat, 0:3] + np.float32([0, 0, 1]); h /= np.linalg.norm(h, axis=1, keepdims=True)
        c, s_ = np.cos(2 * np.pi * turns), np.sin(2 * np.pi * turns)
        sx = np.abs((h[:, 0] * c + h[:, 1] * s_) / h[:, 2]); sy = np.abs((h[:, 1] * c - h[:, 0] * s_) / h[:, 2])
        np.testing.assert_allclose(np.median(sx) / np.median(sy), at / ab, rtol=0.06)
        np.testing.assert_allclose(np.median(sx), at * 0.5774, rtol=0.07)
    # an eighth of a turn: the stretch lies on the diagonal, the geometry axes see the same spread
    h = out = None
    o8 = orc.bsdf_debug(mat(0.125), items); c8 = o8[:, 7].astype(int) == 10
    h8 = o8[c8, 0:3] + np.float32([0, 0, 1]); h8 /= np.linalg.norm(h8, axis=1, keepdims=True)
    np.testing.assert_allclose(np.median(np.abs(h8[:, 0])) / np.median(np.abs(h8[:, 1])), 1.0, rtol=0.05)
    # a quarter turn == the frame (tangentV, -tangentU) without a turn; half a turn == no turn (to rounding: cos / sin of the fp32 angle)
    ev = _frames(20000, rng, 0.45)
    swapped = ev.copy(); swapped[:, 3:6] = ev[:, 6:9]; swapped[:, 6:9] = -ev[:, 3:6]
    q, f0 = orc.bsdf_debug(mat(0.25), ev), orc.bsdf_debug(mat(0.0), swapped)
    np.testing.assert_allclose(q[:, 8:15], f0[:, 8:15], rtol=2e-4, atol=2e-6)
    same = (q[:, 7] == f0[:, 7]) & (q[:, 7] == 10)          # the coat's own samples (the diffuse base beneath follows the exchanged frame, as it must)
    assert same.sum() > 2000 and (q[:, 7] == f0[:, 7]).mean() > 0.999
    np.testing.assert_allclose(q[same, 0:7], f0[same, 0:7], rtol=2e-3, atol=2e-5)
    hlf, z = orc.bsdf_debug(mat(0.5), ev), orc.bsdf_debug(mat(0.0), ev)
    np.testing.assert_allclose(hlf[:, 8:15], z[:, 8:15], rtol=2e-4, atol=2e-6)
    assert not np.allclose(orc.bsdf_debug(mat(0.125), ev)[:, 11:14], z[:, 11:14], rtol=1e-2)      # ... and an eighth is a different material
    # not read without an anisotropic coat: bit for bit the material without the input
    for kw in (dict(coat_roughness_anisotropy=0.0), dict(coat_weight=0.0), dict(coat_weight=0.0, specular_weight=1.0, specular_roughness_anisotropy=0.6)):
        a0, a1 = orc.bsdf_debug(mat(0.0, **kw), ev), orc.bsdf_debug(mat(0.37, **kw), ev)
        assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32)), kw
    # the lobes beneath keep the surface's frame: an anisotropic metal under a turned, isotropic-looking... (the base's glossy samples do not move)
    under = dict(base_metalness=1.0, specular_weight=1.0, specular_roughness=0.35, specular_roughness_anisotropy=0.7, coat_ior=1.5)
    b0, b1 = orc.bsdf_debug(mat(0.0, **under), ev), orc.bsdf_debug(mat(0.2, **under), ev)
    metal = (b0[:, 7] == b1[:, 7]) & (ev[:, 20] >= 0.3)      # xi.z above the coat's share at this incidence: the metal lobe on both sides
    assert metal.sum() > 5000 and np.array_equal(b0[metal, 0:3].view(np.uint32), b1[metal, 0:3].view(np.uint32))
    # reciprocity of the turned coat's evaluate / cos, and evaluate == sampling
    n = 4000
    u = rng.normal(size=(n, 3)); u[:, 2] = np.abs(u[:, 2]) + 0.2; u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = rng.normal(size=(n, 3)); v[:, 2] = np.abs(v[:, 2]) + 0.2; v /= np.linalg.norm(v, axis=1, keepdims=True)
    base = _frames(n, rng, 0.5)
    uv = base.copy(); uv[:, 12:15] = u; uv[:, 15:18] = v
    vu = base.copy(); vu[:, 12:15] = v; vu[:, 15:18] = u
    black = mat(0.2, base_color=(0, 0, 0))                   # the coat alone (its Fresnel factor depends on k.h only: symmetric)
    fuv = orc.bsdf_debug(black, uv)[:, 11] / v[:, 2]; fvu = orc.bsdf_debug(black, vu)[:, 11] / u[:, 2]
    np.testing.assert_allclose(fuv, fvu, rtol=3e-4, atol=1e-6)
    big = _frames(300000, rng, 0.7)
    o = orc.bsdf_debug(mat(0.2), big)
    refl = (o[:, 7].astype(int) & 8) != 0
    np.testing.assert_allclose((o[:, 8:11] + o[:, 11:14]).mean(axis=0) * (2 * np.pi), (o[:, 3:6] * refl[:, None]).mean(axis=0), rtol=0.06, atol=0.01)
    np.testing.assert_allclose(o[:, 14].mean() * 2 * np.pi, refl.mean(), rtol=0.06, atol=0.01)


def test_base_tangent(orc):
    """open_pbr_surface.mtlx:89 (geometry_tangent: the tangent of the dielectric and conductor lobes, :385, 402, 410, 449, 457) in the form documents bind it -- the
    geometry tangent turned by specular_rotation turns.  The stretch of an anisotropic metal's micro-normals follows the turned axes; a quarter turn is the frame with
    tangent and bitangent exchanged (every lobe); half a turn evaluates like no turn; without anisotropic base lobes the input is not read (bit for bit); the coat keeps
    the geometry tangent (and its own turn) whatever the base does; evaluate stays reciprocal and consistent with sampling."""
    rng = np.random.default_rng(62)
    r, a = 0.3, 0.75
    at = r * r * np.sqrt(2.0 / (1.0 + (1.0 - a) ** 2)); ab = (1.0 - a) * at
    def metal(rot, **kw):
        d = dict(base_color=(1, 1, 1), base_metalness=1.0, specular_roughness=r, specular_roughness_anisotropy=a, specular_rotation=rot)
        d.update(kw)
        return MaterialDesc.open_pbr(**d)
    items = _frames(200000, rng, 1.0)
    for turns in (0.125, 0.3, -0.9):
        out = orc.bsdf_debug(metal(turns), items)
        ok = out[:, 7] != 0
        h = out[ok, 0:3] + np.float32([0, 0, 1]); h /= np.linalg.norm(h, axis=1, keepdims=True)
        c, s_ = np.cos(2 * np.pi * turns), np.sin(2 * np.pi * turns)
        sx = np.abs((h[:, 0] * c + h[:, 1] * s_) / h[:, 2]); sy = np.abs((h[:, 1] * c - h[:, 0] * s_) / h[:, 2])
        np.testing.assert_allclose(np.median(sx) / np.median(sy), at / ab, rtol=0.05)
        np.testing.assert_allclose(np.median(sx), at * 0.5774, rtol=0.06)
    # a quarter turn == the frame (tangentV, -tangentU), for a material with every lobe beneath the coat; half a turn evaluates like none
    ev = _frames(20000, rng, 0.45)
    swapped = ev.copy(); swapped[:, 3:6] = ev[:, 6:9]; swapped[:, 6:9] = -ev[:, 3:6]
    def mixed(rot, **kw):
        d = dict(base_color=(0.6, 0.5, 0.4), base_metalness=0.4, specular_roughness=0.35, specular_roughness_anisotropy=0.6, transmission_weight=0.3, base_diffuse_roughness=0.5,
                 specular_rotation=rot)
        d.update(kw)
        return MaterialDesc.open_pbr(**d)
    q, f0 = orc.bsdf_debug(mixed(0.25), ev), orc.bsdf_debug(mixed(0.0), swapped)
    same = q[:, 7] == f0[:, 7]
    assert same.mean() > 0.999
    np.testing.assert_allclose(q[:, 8:15], f0[:, 8:15], rtol=3e-4, atol=3e-6)
    np.testing.assert_allclose(q[same, 0:7], f0[same, 0:7], rtol=3e-3, atol=3e-5)
    hlf, z = orc.bsdf_debug(mixed(0.5), ev), orc.bsdf_debug(mixed(0.0), ev)
    np.testing.assert_allclose(hlf[:, 8:15], z[:, 8:15], rtol=3e-4, atol=3e-6)
    assert not np.allclose(orc.bsdf_debug(mixed(0.125), ev)[:, 11:14], z[:, 11:14], rtol=1e-2)
    # not read without anisotropic base lobes: bit for bit the material without the input (an anisotropic COAT does not count)
    for kw in (dict(specular_roughness_anisotropy=0.0), dict(specular_roughness_anisotropy=0.0, coat_weight=0.6, coat_roughness=0.3, coat_roughness_anisotropy=0.7, coat_rotation=0.1)):
        a0, a1 = orc.bsdf_debug(mixed(0.0, **kw), ev), orc.bsdf_debug(mixed(0.37, **kw), ev)
        assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32)), kw
    # the coat stays on the geometry tangent (+ its own turn) when the base turns: the coat's own samples do not move (to the rounding of the relative turn)
    for coat_turn in (0.0, 0.1):
        coated = dict(specular_weight=0.0, base_metalness=0.0, transmission_weight=0.0, coat_weight=1.0, coat_ior=3.0, coat_roughness=0.3, coat_roughness_anisotropy=0.75, coat_rotation=coat_turn)
        c0, c1 = orc.bsdf_debug(mixed(0.0, **coated), ev), orc.bsdf_debug(mixed(0.3, **coated), ev)
        coat = (c0[:, 7] == 10) & (c1[:, 7] == 10)
        assert coat.sum() > 2000
        np.testing.assert_allclose(c1[coat, 0:7], c0[coat, 0:7], rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(c1[:, 11:14], c0[:, 11:14], rtol=3e-4, atol=3e-6)            # specular_weight 0: the glossy part IS the coat
    # reciprocity (metal: its Fresnel factor depends on k.h only) and evaluate == sampling, both turns at once
    n = 4000
    u = rng.normal(size=(n, 3)); u[:, 2] = np.abs(u[:, 2]) + 0.2; u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = rng.normal(size=(n, 3)); v[:, 2] = np.abs(v[:, 2]) + 0.2; v /= np.linalg.norm(v, axis=1, keepdims=True)
    base = _frames(n, rng, 0.5)
    uv = base.copy(); uv[:, 12:15] = u; uv[:, 15:18] = v
    vu = base.copy(); vu[:, 12:15] = v; vu[:, 15:18] = u
    fuv = orc.bsdf_debug(metal(0.2), uv)[:, 11] / v[:, 2]; fvu = orc.bsdf_debug(metal(0.2), vu)[:, 11] / u[:, 2]
    np.testing.assert_allclose(fuv, fvu, rtol=3e-4, atol=1e-6)
    big = _frames(300000, rng, 0.7)
    o = orc.bsdf_debug(mixed(0.2, transmission_weight=0.0, coat_weight=0.5, coat_roughness=0.25, coat_roughness_anisotropy=0.6, coat_rotation=-0.15), big)
    refl = (o[:, 7].astype(int) & 8) != 0
    np.testing.assert_allclose((o[:, 8:11] + o[:, 11:14]).mean(axis=0) * (2 * np.pi), (o[:, 3:6] * refl[:, None]).mean(axis=0), rtol=0.06, atol=0.01)
    np.testing.assert_allclose(o[:, 14].mean() * 2 * np.pi, refl.mean(), rtol=0.06, atol=0.01)


def test_thin_film(orc):
    """open_pbr_surface.mtlx:300-304, 404-431, 450-464: thin_film_weight mixes a film's interference into the Fresnel factor of the dielectric and metal lobes.
    Our closed form is the Airy summation at three wavelengths (oracle/gi_oracle.cpp "thin film").  Known answers of the reflectance: a film of zero or
    half-wave thickness is invisible, a quarter-wave film of index sqrt(n) is the textbook anti-reflection coating; then the lobes: weight 0 changes nothing
    bit for bit, a film raises the reflected share and lowers what passes by the same amount, colours appear."""
    L = orc.lib()
    n3, lam = 1.5, 550.0
    nf = float(np.sqrt(n3))
    for c in (1.0, 0.8, 0.45, 0.1):
        assert abs(L.orc_film_reflectance(c, 1.7, n3, 0.0, lam) - L.orc_fresnel_dielectric(c, n3)) < 2e-6
    assert abs(L.orc_film_reflectance(1.0, nf, n3, float(lam / (2 * nf)), lam) - L.orc_fresnel_dielectric(1.0, n3)) < 1e-5   # half-wave: absent
    assert L.orc_film_reflectance(1.0, nf, n3, float(lam / (4 * nf)), lam) < 1e-6                                             # quarter-wave at sqrt(n): no reflection
    assert L.orc_film_reflectance(1.0, 2.2, n3, float(lam / (4 * 2.2)), lam) > 4 * L.orc_fresnel_dielectric(1.0, n3)           # a dense quarter-wave film: a mirror coating
    assert L.orc_film_reflectance(0.2, 1.4, 1.0 / 1.5, 300.0, lam) == 1.0                                                     # from inside, beyond the critical angle
    rng = np.random.default_rng(33)
    items = _frames(200000, rng, 0.8)
    plain = MaterialDesc.open_pbr(base_color=(1, 1, 1), specular_roughness=0.25)
    zero = MaterialDesc.open_pbr(base_color=(1, 1, 1), specular_roughness=0.25, thin_film_weight=0.0, thin_film_thickness=0.9, thin_film_ior=2.0)
    film = MaterialDesc.open_pbr(base_color=(1, 1, 1), specular_roughness=0.25, thin_film_weight=1.0, thin_film_thickness=0.12, thin_film_ior=2.2)
    a, z, f = orc.bsdf_debug(plain, items), orc.bsdf_debug(zero, items), orc.bsdf_debug(film, items)
    assert np.array_equal(a.view(np.uint32), z.view(np.uint32))
    glossy = (a[:, 7].astype(int) & 2) != 0                   # the same lobes are chosen (selection keeps the plain Fresnel term) ...
    assert np.array_equal(a[:, 7], f[:, 7]) and np.array_equal(a[:, 0:3], f[:, 0:3])
    ra, rf = (a[:, 3:6] * glossy[:, None]).mean(axis=0), (f[:, 3:6] * glossy[:, None]).mean(axis=0)
    da, df = (a[:, 3:6] * ~glossy[:, None]).mean(axis=0), (f[:, 3:6] * ~glossy[:, None]).mean(axis=0)
    assert np.all(rf > 1.2 * ra) and rf.max() > 2.0 * ra.max() and np.all(df < da)   # ... the film reflects more and lets less through,
    np.testing.assert_allclose(rf + df, ra + da, rtol=0.03)   # what one side gains the other loses (white base: nothing is absorbed)
    assert rf.max() / rf.min() > 1.05                         # and it is coloured
    # metal: the film tints the reflection, energy stays below 1
    m = MaterialDesc.open_pbr(base_color=(0.9, 0.9, 0.9), base_metalness=1.0, specular_roughness=0.3, thin_film_weight=1.0, thin_film_thickness=0.4, thin_film_ior=1.5)
    o = orc.bsdf_debug(m, items)[:, 3:6].mean(axis=0)
    assert np.all(o <= 1.0) and o.max() / o.min() > 1.03
    # from inside, beyond the critical angle (back face, relative eta 1 / 1.5, cos 0.2): total internal reflection -- the Fresnel term and the film's reflectance
    # are both exactly 1, so "what lies beneath the interface" is 0 and not (1 - 1) * 1 / (1 - 1) = NaN (NEE added NaN radiance there)
    inside = _frames(4000, rng, 0.2); inside[:, 21] = 1.0
    for mat in (film, MaterialDesc.open_pbr(base_color=(0.7, 0.7, 0.7), transmission_weight=0.6, thin_film_weight=0.5, thin_film_thickness=0.35, thin_film_ior=1.8,
                                           geometry_thin_walled=False, subsurface_weight=0.0)):
        t = orc.bsdf_debug(mat, inside)
        assert np.isfinite(t).all()
        assert np.all(t[:, 8:11] == 0.0) and np.all(t[:, 14] > 0.0)     # nothing diffuse beneath a totally reflecting interface; the pdf stays that of the glossy lobes


def test_thin_walled_subsurface_lobes(orc):
    """open_pbr_surface.mtlx:140-196, 207-218: opaque_base = mix(diffuse, subsurface_thin_walled, subsurface_weight) with
    subsurface_thin_walled = 1/2 oren_nayar(max(c, 0)) * c (1 - g) + 1/2 translucent(max(c, 0)) * c (1 + g).  Known answers of the sampling routine for a
    thin-walled sheet without specular reflection (F = 0) -- and: the volumetric form of a non-thin-walled material is not modelled (weight ignored)."""
    rng = np.random.default_rng(5)
    c, g, w = np.float32([0.9, 0.6, 0.3]), 0.25, 0.7
    base = np.float32([0.2, 0.4, 0.8])
    m = MaterialDesc.open_pbr(base_color=tuple(base), geometry_thin_walled=True, subsurface_weight=w, subsurface_color=tuple(c), subsurface_scatter_anisotropy=g,
                              specular_weight=0.0)
    out = orc.bsdf_debug(m, _frames(400000, rng, 0.8))
    ev = out[:, 7].astype(int)
    trans, refl = (ev & 16) != 0, (ev & 8) != 0
    # lobe probabilities: transmitted half of the subsurface share, everything else reflected (cosine sampling never fails on a flat frame)
    np.testing.assert_allclose(trans.mean(), 0.5 * w, atol=0.004)
    np.testing.assert_allclose(refl.mean(), 1.0 - 0.5 * w, atol=0.004)
    assert np.all(out[trans, 2] < 0.0) and np.all(out[refl, 2] > 0.0)                        # k2 below / above the surface
    np.testing.assert_allclose(out[trans, 3:6], np.tile(c * c * np.float32(1 + g), (trans.sum(), 1)), rtol=2e-6)   # translucent: colour^2 (1 + g)
    w_refl = np.unique(np.round(out[refl, 3:6], 5), axis=0)
    expect = {tuple(np.round(base, 5)), tuple(np.round(c * c * np.float32(1 - g), 5))}        # base Lambert | subsurface reflection colour^2 (1 - g)
    assert {tuple(r) for r in w_refl} == expect, w_refl
    # evaluate() on the reflection side integrates to the mean sampled reflection weight; its pdf to the reflection probability
    integ = (out[:, 8:11] + out[:, 11:14]).mean(axis=0) * (2 * np.pi)
    np.testing.assert_allclose(integ, (out[:, 3:6] * refl[:, None]).mean(axis=0), rtol=0.02)
    np.testing.assert_allclose(out[:, 14].mean() * 2 * np.pi, refl.mean(), rtol=0.02)
    # not thin-walled: subsurface_weight changes nothing (volumetric subsurface_bsdf is not modelled)
    a = MaterialDesc.open_pbr(base_color=tuple(base), subsurface_weight=w, subsurface_color=tuple(c))
    b = MaterialDesc.open_pbr(base_color=tuple(base))
    items = _frames(2000, rng, 0.6)
    assert np.array_equal(orc.bsdf_debug(a, items), orc.bsdf_debug(b, items))


def test_expf_polynomial(orc):
    L = orc.lib()
    for x in np.concatenate([-np.float32(10.0) ** np.linspace(-6, 1.9, 300), np.float32([0.0, -0.5, -1.0, -87.5, -100.0])]).astype(np.float32):
        got = L.orc_expf(float(x))
        assert got == pytest.approx(np.exp(np.float64(x)), rel=3e-7, abs=1e-38)


def test_cutout_opacity(orc):
    """rp_main.ahit: opacity 1 == opaque; opacity 0 == the mesh is not there (the any-hit draw does not advance the path rng in
    our order-independent restatement); opacityThreshold turns opacity into a binary mask; 0.5 lets about half the rays pass."""
    rs = RenderSettings(spp=4, max_bounces=4)
    base = cornell_box(MAT_DIFFUSE)
    ref, _ = orc.render(base, rs, 48, 27)
    gone = cornell_box(MAT_DIFFUSE); gone.meshes[6].visible = False; gone.meshes[7].visible = False
    ref_gone, _ = orc.render(gone, rs, 48, 27)
    from gatling_amd.scene import P_OPACITY, P_OPACITY_THRESHOLD
    def with_opacity(op, th=0.0):
        d = cornell_box(MAT_DIFFUSE)
        m = MaterialDesc.usd_preview_surface(name="cut", diffuseColor=(0.8, 0.8, 0.8), klass=MAT_DIFFUSE, opacity=op, opacityThreshold=th)
        d.materials.append(m); d.meshes[6].material = d.meshes[7].material = len(d.materials) - 1
        return d
    img1, _ = orc.render(with_opacity(1.0), rs, 48, 27)
    assert np.array_equal(img1, ref)
    # triangle ids are unchanged (the boxes are the last meshes), so "fully transparent" must equal "invisible" up to the
    # 2^-23 chance per candidate that the hash returns exactly 0
    img0, _ = orc.render(with_opacity(0.0), rs, 48, 27)
    assert (img0 != ref_gone).any(axis=-1).sum() <= 2
    imgt, _ = orc.render(with_opacity(0.3, th=0.5), rs, 48, 27)   # below the threshold -> masked out
    assert (imgt != ref_gone).any(axis=-1).sum() <= 2
    imgo, _ = orc.render(with_opacity(0.7, th=0.5), rs, 48, 27)   # above the threshold -> opaque
    assert np.array_equal(imgo, ref)
    half, _ = orc.render(with_opacity(0.5), rs, 48, 27)
    d_full, d_none = np.abs(half - ref).mean(), np.abs(half - ref_gone).mean()
    assert d_full > 1e-4 and d_none > 1e-4  # neither opaque nor absent


def test_dome_light_lookup(orc):
    """rp_main.miss:38-86: no geometry, every primary ray returns texel(direction) * emission; camera-invisible domes show the
    fallback colour times the emission multiplier; an image-less dome light is ignored (Gi.cpp:2221-2230)."""
    from gatling_amd.scene import DomeLight, SceneDesc, CameraDesc
    s = SceneDesc()
    s.camera = CameraDesc(position=(0, 0, 0), forward=(0, 0, -1), up=(0, 1, 0), vfov=1.0)
    env = np.zeros((8, 16, 4), np.float32)
    env[..., 0] = np.linspace(0.0, 1.0, 8)[:, None]   # r grows with v (row): up is bright
    env[..., 1] = 0.25; env[..., 3] = 1.0
    s.textures = [env]
    s.dome_light = DomeLight(texture=0, base_emission=(2.0, 1.0, 1.0))
    rs = RenderSettings(spp=1, max_bounces=2, jittered_sampling=False, clear_color=(0.2, 0.4, 0.6, 1.0))
    img, _ = orc.render(s, rs, 32, 16)
    assert np.allclose(img[..., 1], 0.25) and np.allclose(img[..., 2], 0.0)
    assert img[-1, 16, 0] > img[0, 16, 0]          # row 0 = bottom of the image = looking down = small v
    centre = img[8, 16, 0]                          # looking along -z, horizon: v = 0.5 -> r interpolates to ~0.5, times emission 2
    assert centre == pytest.approx(1.0, abs=0.15)
    rs.dome_light_camera_visible = False
    hidden, _ = orc.render(s, rs, 32, 16)
    q = np.float32(np.floor(np.float32([0.2, 0.4, 0.6]) * 255.0)) / np.float32(255.0)
    assert np.allclose(hidden[..., :3], q * np.float32([2.0, 1.0, 1.0]), atol=1e-6)
    s.dome_light = DomeLight(texture=-1, base_emission=(2.0, 1.0, 1.0))
    rs.dome_light_camera_visible = True
    ignored, _ = orc.render(s, rs, 32, 16)
    assert np.allclose(ignored[..., :3], q, atol=1e-6)
    # rotating the dome by 180 degrees about y == rolling the image by half its width (up to filtering round-off)
    s.dome_light = DomeLight(texture=0, rotation=(0.0, 1.0, 0.0, 0.0))
    env2 = env.copy(); env2[..., 2] = np.linspace(0, 1, 16)[None, :]
    s.textures = [env2]
    rot, _ = orc.render(s, rs, 32, 16)
    s.textures = [np.roll(env2, 8, axis=1)]; s.dome_light = DomeLight(texture=0)
    rolled, _ = orc.render(s, rs, 32, 16)
    assert np.allclose(rot, rolled, atol=2e-5)


def test_textured_inputs(orc):
    """Per-hit material inputs (UsdUVTexture semantics): a constant texture equals the constant parameter up to filter
    round-off; scale/bias act per channel; the normal map changes shading but not geometry AOVs."""
    from gatling_amd.scene import TextureBinding, TEX_BASE_COLOR, TEX_NORMAL
    from gatling_amd.scenes import textured_scene
    rs = RenderSettings(spp=4, max_bounces=4, next_event_estimation=True)
    s = textured_scene(dome=False)
    base, _ = orc.render(s, rs, 48, 27)
    assert np.isfinite(base).all() and base[..., :3].mean() > 0.01
    # ground base colour as a constant texture of the same value
    s2 = textured_scene(dome=False)
    const = np.zeros((2, 2, 4), np.float32); const[..., :3] = (0.5, 0.25, 0.125); const[..., 3] = 1
    s2.textures.append(const)
    s2.materials[0].textures[TEX_BASE_COLOR] = TextureBinding(texture=len(s2.textures) - 1)
    a, _ = orc.render(s2, rs, 48, 27)
    s3 = textured_scene(dome=False)
    del s3.materials[0].textures[TEX_BASE_COLOR]
    s3.materials[0].params[0:3] = (0.5, 0.25, 0.125)
    b, _ = orc.render(s3, rs, 48, 27)
    assert np.allclose(a, b, rtol=1e-4, atol=1e-4)
    # scale 0.5 on a texture of twice the value: same again
    s4 = textured_scene(dome=False)
    s4.textures.append(const * np.float32([2, 2, 2, 1]))
    s4.materials[0].textures[TEX_BASE_COLOR] = TextureBinding(texture=len(s4.textures) - 1, scale=(0.5, 0.5, 0.5, 1.0))
    c, _ = orc.render(s4, rs, 48, 27)
    assert np.allclose(a, c, rtol=1e-4, atol=1e-4)
    # without the normal map the image changes, the Normal AOV (geometry state) does not
    s5 = textured_scene(dome=False)
    del s5.materials[0].textures[TEX_NORMAL]
    d, _ = orc.render(s5, rs, 48, 27)
    assert not np.allclose(base, d, atol=1e-3)
    n1 = orc.render_aovs(s, rs, 48, 27, ["normal"])["normal"]
    n2 = orc.render_aovs(s5, rs, 48, 27, ["normal"])["normal"]
    assert np.array_equal(n1, n2)


def test_volume_medium_stack(orc):
    """rp_main.rgen:48-97, 317-346, 462-477; rp_main.miss:16-34; rp_main.chit:160-186, 447-480 with MEDIUM_STACK_SIZE > 0."""
    from gatling_amd.scenes import volume_scene
    rs = lambda stack, **kw: RenderSettings(spp=6, max_bounces=12, next_event_estimation=True, medium_stack_size=stack, **kw)
    # one absorbing, non-scattering, un-nested medium: the stack top holds what the 1-bit toggle reads off the hit material
    plain = volume_scene(scatter=(0, 0, 0), nested=False)
    a, ca = orc.render(plain, rs(0), 64, 36, threads=4)
    b, cb = orc.render(plain, rs(1), 64, 36, threads=4)
    assert np.array_equal(a, b) and ca["segments"] == cb["segments"]
    # scattering: the walk changes the image, stays finite, and a deeper stack than the nesting depth changes nothing
    murky = volume_scene()
    i0, _ = orc.render(murky, rs(0), 64, 36, threads=4)
    i1, c1 = orc.render(murky, rs(1), 64, 36, threads=4)
    i2, c2 = orc.render(murky, rs(2), 64, 36, threads=4)
    i4, c4 = orc.render(murky, rs(4), 64, 36, threads=4)
    assert np.isfinite(i1).all() and np.isfinite(i2).all()
    assert not np.array_equal(i0, i1) and not np.array_equal(i1, i2)
    assert np.array_equal(i2, i4) and c2["segments"] == c4["segments"]
    # in-medium scattering events are segments that end without a surface hit, yet the path goes on
    assert c2["segments"] > c2["hits"] + c2["samples"] * 0  # misses exist ...
    assert sum(c2["bounce_histogram"][1:]) > sum(orc.render(plain, rs(2), 64, 36, threads=4)[1]["bounce_histogram"][1:]) * 0.5
    # metersPerSceneUnit scales the optical depth: a 100x smaller unit makes the medium nearly transparent
    thin, _ = orc.render(murky, rs(2, meters_per_scene_unit=0.01), 64, 36, threads=4)
    assert np.isfinite(thin).all() and not np.array_equal(thin, i2)


def test_textured_cutout_opacity(orc):
    """Opacity from a texture, evaluated at each candidate hit's st (rp_main.ahit:51-60): an all-one mask is the opaque card, an
    all-zero mask removes the card, and a half mask through opacityThreshold is (up to edge pixels) the card cut in half."""
    from gatling_amd.meshprep import bake_vertices
    from gatling_amd.scene import MeshDesc, TEX_OPACITY, TEX_WRAP_CLAMP, TextureBinding
    rs = RenderSettings(spp=4, max_bounces=4, next_event_estimation=True)

    def scene(mask, x1=0.6, threshold=0.5):
        d = cornell_box(MAT_DIFFUSE)
        d.rect_lights = [RectLight(origin=(0, 0, 0.9), t0=(1, 0, 0), t1=(0, -1, 0), base_emission=(10, 10, 10), width=0.7, height=0.5)]
        if mask is not None:
            tex = np.ones((1, len(mask), 4), np.float32); tex[0, :, 3] = mask
            d.textures = [tex]
        m = MaterialDesc.usd_preview_surface(name="card", diffuseColor=(0.1, 0.7, 0.2), klass=MAT_DIFFUSE, opacityThreshold=threshold)
        if mask is not None:
            m.textures = {TEX_OPACITY: TextureBinding(texture=0, wrap_s=TEX_WRAP_CLAMP, wrap_t=TEX_WRAP_CLAMP, channel=3)}
        d.materials.append(m)
        p = np.array([[-0.6, -0.3, -0.5], [x1, -0.3, -0.5], [x1, -0.3, 0.4], [-0.6, -0.3, -0.5], [x1, -0.3, 0.4], [-0.6, -0.3, 0.4]], np.float32)
        uv = np.array([[0, 0], [1, 0], [1, 1], [0, 0], [1, 1], [0, 1]], np.float32)
        d.meshes.append(MeshDesc(name="/Card", vertices=bake_vertices(p, np.tile([0, -1, 0], (6, 1)), uv), faces=np.arange(6, dtype=np.uint32).reshape(-1, 3),
                                 material=len(d.materials) - 1, id=99, double_sided=True))
        return d
    opaque, _ = orc.render(scene(None), rs, 64, 36)
    ones, _ = orc.render(scene([1.0, 1.0]), rs, 64, 36)
    assert np.array_equal(ones, opaque)
    absent = scene(None); absent.meshes[-1].visible = False
    ref_absent, _ = orc.render(absent, rs, 64, 36)
    zeros, _ = orc.render(scene([0.0, 0.0]), rs, 64, 36)
    assert (zeros != ref_absent).any(axis=-1).sum() <= 2   # the card is the last mesh: triangle ids of the rest are unchanged
    # [1, 0] with clamp + bilinear: alpha falls from 1 at u = 0.25 to 0 at u = 0.75, the threshold 0.5 cuts at u = 0.5
    half_mask, _ = orc.render(scene([1.0, 0.0]), rs, 64, 36)
    half_geom, _ = orc.render(scene(None, x1=0.0), rs, 64, 36)
    differing = (np.abs(half_mask - half_geom).max(axis=-1) > 0.25).mean()
    assert differing < 0.03 and np.abs(half_mask - opaque).mean() > 1e-3 and np.abs(half_mask - ref_absent).mean() > 1e-3
    # without a threshold the same ramp is used stochastically: between the two
    soft, _ = orc.render(scene([1.0, 0.0], threshold=0.0), rs, 64, 36)
    assert np.abs(soft - half_mask).mean() > 1e-4


def test_nee_aov_is_the_bounce0_shadow_test(orc):
    """rp_main.rgen:431-435: the NEE AOV is written at bounce 0 only, for every sample; a shadow ray that cannot contribute is
    dispatched with tMin = tMax = 0 (:406-410), misses, and rp_main_shadow.miss clears `shadowed` -> green.  So with NEE compiled in
    every pixel is red or green (primary misses included) and the result does not depend on later bounces; with NEE off the block is
    compiled out and the AOV keeps its clear value."""
    desc = cornell_box()
    desc.rect_lights = [RectLight(origin=(0, 0, 0.9), t0=(1, 0, 0), t1=(0, -1, 0), base_emission=(10, 10, 10), width=0.7, height=0.5)]
    clear = {"nee": (0.25, 0.5, 0.75, 0.0)}
    a = orc.render_aovs(desc, RenderSettings(spp=3, max_bounces=1, next_event_estimation=True), 48, 27, ["nee"], clear_values=clear)["nee"][..., :3]
    b = orc.render_aovs(desc, RenderSettings(spp=3, max_bounces=7, next_event_estimation=True), 48, 27, ["nee"], clear_values=clear)["nee"][..., :3]
    assert np.array_equal(a, b)  # bounces > 0 never touch the AOV
    kinds = {tuple(v) for v in np.unique(a.reshape(-1, 3), axis=0).tolist()}
    assert kinds == {(1.0, 0.0, 0.0), (0.0, 1.0, 0.0)}
    off = orc.render_aovs(desc, RenderSettings(spp=3, max_bounces=7), 48, 27, ["nee"], clear_values=clear)["nee"]
    assert np.allclose(off, clear["nee"])
    zero = orc.render_aovs(desc, RenderSettings(spp=3, max_bounces=0, next_event_estimation=True), 48, 27, ["nee"], clear_values=clear)["nee"]
    assert np.allclose(zero, clear["nee"])  # the loop body never runs


# ---- OpenPBR graph pieces (src/gi/mtlx/open_pbr_surface.mtlx): known answers recomputed here in numpy float32 from the graph's own nodes ----
def _f32(x):
    return np.float32(x)


def test_openpbr_coat_roughening_and_darkening_known_answers(orc):
    """effective_specular_roughness (mtlx:101-131) and modulated_base_darkening (:470-541) enter the closed form exactly as the graph
    composes them: checked through the sampled weights of a smooth-ish coated dielectric against numpy float32 restatements."""
    r, cr, coat, cior, bc, sw, metal, cd = _f32(0.3), _f32(0.5), _f32(1.0), _f32(1.6), np.float32([0.8, 0.4, 0.2]), _f32(1.0), _f32(0.0), _f32(1.0)
    # :101-131
    ra = np.sqrt(np.sqrt(np.minimum(_f32(1.0), _f32(2.0) * (cr * cr) * (cr * cr) + (r * r) * (r * r))))
    r_eff = ra * coat + r * (_f32(1.0) - coat)
    assert r_eff == pytest.approx(0.6040, abs=2e-4)  # (2 * 0.5^4 + 0.3^4)^(1/4)
    # :470-541
    f0c = ((cior - 1) / (cior + 1)) ** 2
    K = _f32(1.0) - (_f32(1.0) - f0c) / (cior * cior)
    Eb = (bc * sw) * metal + bc * (_f32(1.0) - metal)
    bd = (_f32(1.0) - K) / (_f32(1.0) - Eb * K)
    mod = bd * (coat * cd) + (_f32(1.0) - coat * cd)
    m = MaterialDesc.open_pbr(base_color=tuple(bc), specular_roughness=float(r), coat_weight=float(coat), coat_roughness=float(cr), coat_ior=float(cior))
    m0 = MaterialDesc.open_pbr(base_color=tuple(bc), specular_roughness=float(r), coat_weight=float(coat), coat_roughness=float(cr), coat_ior=float(cior), coat_darkening=0.0)
    rng = np.random.default_rng(5)
    items = _frames(4000, rng, 0.8)
    a, b = orc.bsdf_debug(m, items), orc.bsdf_debug(m0, items)
    diffuse = (a[:, 7].astype(int) & 1) != 0  # EV_DIFFUSE
    assert diffuse.sum() > 500 and np.array_equal(a[:, 7], b[:, 7])
    # the diffuse weight is base_color * coat_attenuation * darkening: the ratio with / without darkening is `mod`
    np.testing.assert_allclose(a[diffuse, 3:6] / b[diffuse, 3:6], np.tile(mod, (diffuse.sum(), 1)), rtol=2e-6)
    assert np.all(mod < 1.0) and np.all(mod > 0.3)
    # roughening: with coat_ior = 1 the coat lobe has F0 = 0 (never sampled at normal incidence), Kcoat = 0 (no darkening) and eta_s =
    # specular_ior, so the coated material must behave exactly like an uncoated one whose specular_roughness is r_eff
    coated = MaterialDesc.open_pbr(base_color=tuple(bc), specular_roughness=float(r), coat_weight=1.0, coat_roughness=float(cr), coat_ior=1.0)
    plain = MaterialDesc.open_pbr(base_color=tuple(bc), specular_roughness=float(r_eff))
    top = _frames(4000, rng, 1.0)
    np.testing.assert_array_equal(orc.bsdf_debug(coated, top)[:, :8], orc.bsdf_debug(plain, top)[:, :8])


def test_openpbr_emission_through_coat(orc):
    """emission_edf (mtlx:590-619): mix(uncoated, coat_color * (1 - coat_F0) * (1 - (1 - cos)^5), coat_weight) at normal incidence."""
    from gatling_amd.scenes import cornell_box  # noqa: F401
    v, f = build_mesh_arrays([(-50, -50, 0), (50, -50, 0), (50, 50, 0), (-50, 50, 0)], [4], [0, 1, 2, 3])
    out = {}
    for coat in (0.0, 1.0):
        mat = MaterialDesc.open_pbr(base_color=(0, 0, 0), specular_weight=0.0, emission_luminance=2.0, emission_color=(1.0, 0.5, 0.25),
                                    coat_weight=coat, coat_color=(0.5, 0.8, 1.0), coat_ior=1.6)
        desc = SceneDesc(meshes=[MeshDesc("floor", v, f, 0, double_sided=True)], materials=[mat],
                         camera=CameraDesc(position=(0, 0, 3), forward=(0, 0, -1), up=(0, 1, 0), vfov=0.02))
        rs = RenderSettings(spp=4, max_bounces=1, clear_color=(0, 0, 0, 0), max_sample_value=1e9)
        out[coat] = orc.render(desc, rs, 4, 4)[0][..., :3].mean(axis=(0, 1))
    f0c = ((1.6 - 1) / (1.6 + 1)) ** 2
    np.testing.assert_allclose(out[0.0], [2.0, 1.0, 0.5], rtol=1e-5)
    np.testing.assert_allclose(out[1.0], np.array([2.0, 1.0, 0.5]) * np.array([0.5, 0.8, 1.0]) * (1 - f0c), rtol=1e-4)


def test_eon_diffuse_is_reciprocal_energy_preserving_and_meets_lambert(orc):
    """oren_nayar_diffuse_bsdf with energy_compensation (mtlx:200-206), energy-preserving Oren-Nayar: white albedo loses no energy
    at any roughness, evaluate() is symmetric in (k1, k2), and roughness -> 0 meets the Lambert lobe."""
    rng = np.random.default_rng(9)
    for rough in (0.25, 1.0):
        m = MaterialDesc.open_pbr(base_color=(1, 1, 1), base_diffuse_roughness=rough, specular_weight=0.0)
        for c in (0.95, 0.5, 0.15):
            out = orc.bsdf_debug(m, _frames(60000, rng, c))
            albedo = out[:, 3:6].mean(axis=0)
            np.testing.assert_allclose(albedo, 1.0, atol=0.02)  # E_ss + E_ms = 1 for rho = 1
        items = _frames(2000, rng, 0.6)
        swapped = items.copy(); swapped[:, 12:15], swapped[:, 15:18] = items[:, 15:18], items[:, 12:15]
        a, b = orc.bsdf_debug(m, items), orc.bsdf_debug(m, swapped)
        # evaluate returns bsdf * cos(k2): divide the cosines out before comparing
        ok = items[:, 17] > 0.01  # (k1's cosine is clamped to 1e-4 inside the closed forms: leave grazing directions out)
        fa, fb = a[ok, 8] / items[ok, 17], b[ok, 8] / swapped[ok, 17]
        np.testing.assert_allclose(fa, fb, rtol=2e-5)
    lam = orc.bsdf_debug(MaterialDesc.open_pbr(base_color=(0.7, 0.5, 0.3), specular_weight=0.0), _frames(500, np.random.default_rng(1), 0.7))
    eon = orc.bsdf_debug(MaterialDesc.open_pbr(base_color=(0.7, 0.5, 0.3), specular_weight=0.0, base_diffuse_roughness=1e-4), _frames(500, np.random.default_rng(1), 0.7))
    np.testing.assert_allclose(eon[:, 8:11], lam[:, 8:11], rtol=2e-4, atol=1e-7)


def test_thin_walled_semantics(orc):
    """mdl_thin_walled (rp_main.chit:155-157, 188-189, 218-220, 447): transmission leaves the inside/outside state alone (no Beer-Lambert
    absorption behind a thin-walled pane, unlike a solid one), does not bend the ray, and the ThinWalled AOV turns red."""
    pane_v, pane_f = build_mesh_arrays([(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)], [4], [0, 1, 2, 3])
    back_v, back_f = build_mesh_arrays([(-5, -5, -2), (5, -5, -2), (5, 5, -2), (-5, 5, -2)], [4], [0, 1, 2, 3])
    imgs = {}
    for thin in (False, True):
        glass = MaterialDesc.open_pbr(base_color=(1, 1, 1), transmission_weight=1.0, transmission_color=(0.2, 0.6, 0.9), transmission_depth=0.5,
                                      specular_roughness=0.0, specular_weight=0.0, geometry_thin_walled=thin)
        wall = MaterialDesc.open_pbr(base_color=(0, 0, 0), specular_weight=0.0, emission_luminance=1.0)
        desc = SceneDesc(meshes=[MeshDesc("pane", pane_v, pane_f, 0, double_sided=True), MeshDesc("wall", back_v, back_f, 1, double_sided=True)],
                         materials=[glass, wall], camera=CameraDesc(position=(0.3, 0, 3), forward=(-0.1, 0, -1), up=(0, 1, 0), vfov=0.05))
        rs = RenderSettings(spp=16, max_bounces=4, clear_color=(0, 0, 0, 0), max_sample_value=1e9, rr_bounce_offset=100, medium_stack_size=2)
        imgs[thin] = orc.render(desc, rs, 4, 4)[0][..., :3].mean(axis=(0, 1))
        aov = orc.render_aovs(desc, rs, 4, 4, ["thinWalled"])["thinWalled"][..., :3].mean(axis=(0, 1))
        np.testing.assert_allclose(aov, [1, 0, 0] if thin else [0, 1, 0], atol=1e-6)
    # solid pane: the transmission pushes the glass medium, the wall hit is attenuated by exp(-sigma_t d); thin-walled: nothing is pushed
    assert np.allclose(imgs[True], imgs[True][0], rtol=1e-4) and imgs[True][0] > 0.9   # specular_weight 0 -> F = 0: everything passes, untinted (depth > 0)
    assert imgs[False][0] < 0.1 * imgs[True][0] and imgs[False][2] > imgs[False][0]   # Beer-Lambert with the blue-ish transmission colour


def test_scene_data_int_nearest_and_named_values(orc):
    """scene_data_lookup_int (mdl_interface.glsl:426-457): the value of the vertex with the largest barycentric weight; CAMERA_POSITION /
    FRAME (:329-334, :390-395) come from the camera and the render settings."""
    from gatling_amd.scene import INTERP_VERTEX, PRIMVAR_INT, Primvar, TEX_EMISSION, TEX_ROUGHNESS
    v, f = build_mesh_arrays([(-1, -1, 0), (1, -1, 0), (0, 1, 0)], [3], [0, 1, 2])
    mat = MaterialDesc.open_pbr(base_color=(0, 0, 0), specular_weight=0.0, emission_luminance=1.0)
    mat.primvar_inputs = {TEX_EMISSION: "level"}
    mesh = MeshDesc("tri", v, f, 0, double_sided=True)
    mesh.primvars = [Primvar("level", PRIMVAR_INT, INTERP_VERTEX, np.int32([1, 2, 4]))]
    cam = CameraDesc(position=(0, -0.2, 4), forward=(0, 0, -1), up=(0, 1, 0), vfov=0.6)
    desc = SceneDesc(meshes=[mesh], materials=[mat], camera=cam)
    rs = RenderSettings(spp=1, max_bounces=1, clear_color=(0, 0, 0, 0), jittered_sampling=False, max_sample_value=1e9)
    img = orc.render(desc, rs, 64, 64)[0][..., 0]
    vals = set(np.unique(img).tolist())
    assert vals == {0.0, 1.0, 2.0, 4.0}  # background + exactly the three vertex values: nothing in between
    # camera position as an emission colour, frame as a roughness (visible through the emission only for the former)
    mat.primvar_inputs = {TEX_EMISSION: "CAMERA_POSITION", TEX_ROUGHNESS: "FRAME"}
    cam2 = CameraDesc(position=(0.5, 0.25, 4), forward=(0, 0, -1), up=(0, 1, 0), vfov=0.3)
    img = orc.render(SceneDesc(meshes=[mesh], materials=[mat], camera=cam2), RenderSettings(spp=1, max_bounces=1, clear_color=(0, 0, 0, 0), jittered_sampling=False,
                                                                                         max_sample_value=1e9, frame=0.5), 16, 16)[0]
    hit = img[..., 2] > 0
    assert hit.any()
    np.testing.assert_allclose(img[hit][:, :3], np.tile(np.float32([0.5, 0.25, 4.0]), (hit.sum(), 1)), rtol=1e-6)


def _emissive_quad(transform, wrap=(1, 1)):
    """A quad facing the camera whose only light is its own emission texture (no lights, black background): with one bounce a pixel shows the texel its st maps to."""
    from gatling_amd.meshprep import bake_vertices
    from gatling_amd.scene import TEX_EMISSION, CameraDesc, MeshDesc, SceneDesc, TextureBinding
    rng = np.random.default_rng(11)
    s = SceneDesc()
    img = np.zeros((12, 20, 4), np.float32); img[..., :3] = rng.uniform(0.1, 2.0, (12, 20, 3)); img[..., 3] = 1.0
    s.textures.append(img)
    m = MaterialDesc.usd_preview_surface(name="screen", diffuseColor=(0.0, 0.0, 0.0), roughness=1.0)
    m.textures = {TEX_EMISSION: TextureBinding(texture=0, wrap_s=wrap[0], wrap_t=wrap[1], scale=(1.5, 0.5, 2.0, 1.0), bias=(0.01, 0.02, 0.03, 0.0), transform=transform)}
    s.materials = [m]
    p = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    uv = np.array([[0, 0], [1.7, 0], [1.7, 1.3], [0, 0], [1.7, 1.3], [0, 1.3]], np.float32)
    s.meshes = [MeshDesc(name="/Q", vertices=bake_vertices(p, np.tile([0, -1, 0], (6, 1)), uv), faces=np.arange(6, dtype=np.uint32).reshape(-1, 3), material=0, double_sided=True)]
    s.camera = CameraDesc(position=(0.0, -3.0, 0.0), forward=(0.0, 1.0, 0.0), up=(0.0, 0.0, 1.0), vfov=0.9)
    return s


def test_texture_coordinate_transform(orc):
    """UsdTransform2d upstream of a UsdUVTexture's st (VERDICT r03 missing #4), folded into six floats per binding: st' = (xf0 s + xf1 t) + xf2, (xf3 s + xf4 t) + xf5.
    Known answers: each pixel of a self-lit quad equals the texel at the TRANSFORMED texture coordinate of its hit (coordinates from the Texcoords AOV, texel from the
    oracle's own sampler), for a rotation + non-uniform scale + translation under every wrap mode; the identity transform is the untransformed material bit for bit."""
    from gatling_amd.scene import usd_transform_2d
    rs = RenderSettings(spp=1, max_bounces=1, jittered_sampling=False, clear_color=(0.0, 0.0, 0.0, 0.0), max_sample_value=1e9)
    w, h = 40, 30
    plain, _ = orc.render(_emissive_quad(None), rs, w, h)
    ident, _ = orc.render(_emissive_quad((1.0, 0.0, 0.0, 0.0, 1.0, 0.0)), rs, w, h)
    assert np.array_equal(plain.view(np.uint32), ident.view(np.uint32))
    xf = usd_transform_2d(rotation_deg=33.0, scale=(1.8, 0.6), translation=(0.25, -0.4))
    np.testing.assert_allclose(xf, (np.cos(np.radians(33)) * 1.8, -np.sin(np.radians(33)) * 0.6, 0.25, np.sin(np.radians(33)) * 1.8, np.cos(np.radians(33)) * 0.6, -0.4), rtol=1e-6)
    for wrap in ((1, 1), (0, 2), (3, 1)):     # repeat / clamp + mirrored repeat / clip
        desc = _emissive_quad(xf, wrap)
        img, _ = orc.render(desc, rs, w, h)
        st = orc.render_aovs(desc, rs, w, h, ["texcoords"])["texcoords"]
        assert not np.array_equal(img, orc.render(_emissive_quad(None, wrap), rs, w, h)[0])
        f = np.float32
        xf32 = [f(x) for x in xf]
        b = desc.materials[0].textures[1]
        hit = 0
        for y in range(h):
            for x in range(w):
                s_, t_ = f(st[y, x, 0]), f(st[y, x, 1])
                if (img[y, x, :3] == 0).all() and s_ == 0 and t_ == 0:
                    continue                    # the ray missed the quad
                u = f(f(xf32[0] * s_) + f(xf32[1] * t_)) + xf32[2]
                v = f(f(xf32[3] * s_) + f(xf32[4] * t_)) + xf32[5]
                texel = np.float32(orc.tex_lookup(desc.textures[0], float(u), float(v), wrap[0], wrap[1]))
                want = texel[:3] * np.float32(b.scale[:3]) + np.float32(b.bias[:3])
                np.testing.assert_array_equal(img[y, x, :3], want.astype(np.float32), err_msg=f"pixel {x},{y} wrap {wrap}")
                hit += 1
        assert hit > w * h // 4


def _sss_ball(material, bounces=200):
    """A closed icosphere in a uniform white environment (the colour clear value lights the scene): a furnace for transmissive / scattering materials."""
    from gatling_amd.meshprep import bake_vertices
    from gatling_amd.scene import CameraDesc, MeshDesc, SceneDesc
    from gatling_amd.scenes import icosphere
    pts, faces = icosphere(2)
    s = SceneDesc()
    s.materials = [material]
    s.meshes = [MeshDesc(name="/Ball", vertices=bake_vertices(pts, pts), faces=faces, material=0)]
    s.camera = CameraDesc(position=(0.0, -3.2, 0.0), forward=(0.0, 1.0, 0.0), up=(0.0, 0.0, 1.0), vfov=0.7)
    rs = RenderSettings(spp=24, max_bounces=bounces, rr_bounce_offset=4000, max_sample_value=1e9, clear_color=(1.0, 1.0, 1.0, 1.0), medium_stack_size=2)
    return s, rs


def test_volumetric_subsurface(orc):
    """open_pbr_surface.mtlx:182-192, 207-218: subsurface_bsdf(color, radius * radius_scale, anisotropy) for materials that are not thin-walled, rendered through the
    medium stack (oracle/gi_oracle.cpp opbr_params "Volumetric subsurface": diffuse transmission at the boundary, extinction 1 / radius, van de Hoek albedo inversion).
    White furnace: a white subsurface colour inverts to single-scattering albedo 1, so a ball in a white environment is invisible (energy in = energy out); colours
    come out in the order of subsurface_color; weight 0 and renders without a medium stack are the material without the lobe, bit for bit."""
    w = h = 28
    white = MaterialDesc.open_pbr(base_color=(1, 1, 1), specular_weight=0.0, subsurface_weight=1.0, subsurface_color=(1, 1, 1), subsurface_radius=0.4,
                                  subsurface_radius_scale=(1.0, 1.0, 1.0))
    # (the reference's loop multiplies the throughput by exp(-sigma_t d) at every surface hit inside a medium -- rp_main.chit:174-184 -- although the collision
    # distance was SAMPLED, rp_main.rgen:317-346: a dense white medium therefore darkens.  Restated literally, so the furnace holds in the thin limit only.)
    thin = MaterialDesc.open_pbr(base_color=(1, 1, 1), specular_weight=0.0, subsurface_weight=1.0, subsurface_color=(1, 1, 1), subsurface_radius=4000.0,
                                 subsurface_radius_scale=(1.0, 1.0, 1.0))
    desc, rs = _sss_ball(thin)
    img, cnt = orc.render(desc, rs, w, h, threads=8)
    assert np.isfinite(img).all()
    np.testing.assert_allclose(img[..., :3].mean(axis=(0, 1)), 1.0, atol=0.01)     # mean free path >> the ball: enters and leaves by the two cosine lobes (the few samples below the faceted surface are absorbed)
    desc, rs = _sss_ball(white)
    img, cnt = orc.render(desc, rs, w, h, threads=8)
    assert np.isfinite(img).all()
    m = float(img[..., :3].mean())
    assert 0.5 < m < 0.99, m
    assert cnt["segments"] > 3 * cnt["samples"] * 0.2                               # the walk really happened (many segments per path that entered)
    tinted = MaterialDesc.open_pbr(base_color=(1, 1, 1), specular_weight=0.0, subsurface_weight=1.0, subsurface_color=(0.8, 0.4, 0.1), subsurface_radius=0.3,
                                   subsurface_scatter_anisotropy=0.3)
    desc, rs = _sss_ball(tinted)
    img, _ = orc.render(desc, rs, w, h, threads=8)
    ball = img[h // 2 - 4:h // 2 + 4, w // 2 - 4:w // 2 + 4, :3].mean(axis=(0, 1))
    assert 1.0 > ball[0] > ball[1] > ball[2] > 0.0, ball
    # half weight: half of the opaque base stays a diffuse reflector of base_color
    half = MaterialDesc.open_pbr(base_color=(0.2, 0.2, 0.9), specular_weight=0.0, subsurface_weight=0.5, subsurface_color=(0.8, 0.4, 0.1), subsurface_radius=0.3)
    desc, rs = _sss_ball(half)
    mixed = orc.render(desc, rs, w, h, threads=8)[0][h // 2 - 4:h // 2 + 4, w // 2 - 4:w // 2 + 4, :3].mean(axis=(0, 1))
    assert ball[0] > mixed[0] > 0.2 and mixed[2] > ball[2]
    # off switches: weight 0, and no medium stack
    plain = MaterialDesc.open_pbr(base_color=(0.2, 0.2, 0.9), specular_weight=0.0)
    zero = MaterialDesc.open_pbr(base_color=(0.2, 0.2, 0.9), specular_weight=0.0, subsurface_weight=0.0, subsurface_color=(0.8, 0.4, 0.1), subsurface_radius=0.3)
    a, _ = orc.render(*_sss_ball(plain), w, h, threads=8); b, _ = orc.render(*_sss_ball(zero), w, h, threads=8)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    desc, rs = _sss_ball(half); rs.medium_stack_size = 0
    c, _ = orc.render(desc, rs, w, h, threads=8)
    desc2, rs2 = _sss_ball(plain); rs2.medium_stack_size = 0
    d, _ = orc.render(desc2, rs2, w, h, threads=8)
    assert np.array_equal(c.view(np.uint32), d.view(np.uint32))


def _coated_ball(coat_map, coat_weight=1.0, scale=(2.0, 2.0, 2.0, 1.0), bias=(-1.0, -1.0, -1.0, 0.0)):
    from gatling_amd.meshprep import bake_vertices
    from gatling_amd.scene import TEX_COAT_NORMAL, CameraDesc, MeshDesc, SceneDesc, TextureBinding
    from gatling_amd.scenes import icosphere
    pts, faces = icosphere(3)
    uv = np.stack([np.arctan2(pts[:, 1], pts[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(pts[:, 2], -1, 1)) / np.pi], axis=1).astype(np.float32)
    s = SceneDesc()
    m = MaterialDesc.open_pbr(base_color=(0.05, 0.05, 0.05), specular_weight=0.0, coat_weight=coat_weight, coat_roughness=0.15, coat_ior=1.6)
    if coat_map is not None:
        s.textures.append(coat_map)
        m.textures = {TEX_COAT_NORMAL: TextureBinding(texture=0, scale=scale, bias=bias)}
    s.materials = [m]
    s.meshes = [MeshDesc(name="/Ball", vertices=bake_vertices(pts, pts, uv), faces=faces, material=0)]
    s.rect_lights = [RectLight(origin=(0.0, -2.0, 3.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(30, 30, 30), width=0.6, height=0.6)]
    s.camera = CameraDesc(position=(0.0, -3.2, 0.0), forward=(0.0, 1.0, 0.0), up=(0.0, 0.0, 1.0), vfov=0.7)
    return s


def test_coat_normal(orc):
    """OpenPBR geometry_coat_normal (open_pbr_surface.mtlx:87, 560): a tangent-space normal map gives the coat lobe a shading frame of its own; the base keeps the
    surface's.  A flat map is (to rounding) the unmapped material; a bumpy map breaks the coat's highlight up while a material without coat does not see it at all
    (bit for bit); UsdPreviewSurface ignores the slot."""
    rs = RenderSettings(spp=16, max_bounces=3, next_event_estimation=True, clear_color=(0.0, 0.0, 0.0, 0.0), max_sample_value=1e9)
    w = h = 40
    flat = np.zeros((4, 4, 4), np.float32); flat[..., :3] = (0.5, 0.5, 1.0); flat[..., 3] = 1.0
    yy, xx = np.mgrid[0:32, 0:64]
    bumpy = np.zeros((32, 64, 4), np.float32)
    bumpy[..., 0] = 0.5 + 0.35 * np.sin(xx * 1.7); bumpy[..., 1] = 0.5 + 0.35 * np.cos(yy * 2.3); bumpy[..., 2] = 0.85; bumpy[..., 3] = 1.0
    plain, _ = orc.render(_coated_ball(None), rs, w, h, threads=8)
    a, _ = orc.render(_coated_ball(flat), rs, w, h, threads=8)
    b, _ = orc.render(_coated_ball(bumpy), rs, w, h, threads=8)
    assert np.isfinite(b).all()
    np.testing.assert_allclose(a[..., :3].mean(), plain[..., :3].mean(), rtol=0.03)      # a flat map is the surface normal again (renormalised, Iray bend: not bit-identical)
    assert np.abs(a - plain).mean() < 0.2 * np.abs(b - plain).mean()                      # ... while bumps really move the highlight
    assert np.abs(b - plain).max() > 0.05
    c0, _ = orc.render(_coated_ball(None, coat_weight=0.0), rs, w, h, threads=8)
    c1, _ = orc.render(_coated_ball(bumpy, coat_weight=0.0), rs, w, h, threads=8)
    assert np.array_equal(c0.view(np.uint32), c1.view(np.uint32))                          # no coat, no effect


def _glass_ball(weight=None, colour=None, tw=0.0, tc=(1.0, 1.0, 1.0), depth=0.0, primvar=False):
    """A uv-mapped ball over a lit back wall; `weight` / `colour`: textures for transmission_weight / transmission_color (None: the constants `tw` / `tc`)."""
    from gatling_amd.meshprep import bake_vertices
    from gatling_amd.scene import TEX_TRANSMISSION_COLOR, TEX_TRANSMISSION_WEIGHT, CameraDesc, MeshDesc, Primvar, SceneDesc, TextureBinding
    from gatling_amd.scenes import icosphere
    pts, faces = icosphere(3)
    uv = np.stack([np.arctan2(pts[:, 1], pts[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(pts[:, 2], -1, 1)) / np.pi], axis=1).astype(np.float32)
    s = SceneDesc()
    m = MaterialDesc.open_pbr(base_color=(0.8, 0.2, 0.1), specular_roughness=0.1, transmission_weight=tw, transmission_color=tc, transmission_depth=depth)
    m.textures = {}
    if weight is not None:
        weight, wb = weight if isinstance(weight, tuple) else (weight, (0.0, 0.0, 0.0, 0.0))
        s.textures.append(weight); m.textures[TEX_TRANSMISSION_WEIGHT] = TextureBinding(texture=len(s.textures) - 1, bias=wb)
    if colour is not None:
        colour, cb = colour if isinstance(colour, tuple) else (colour, (0.0, 0.0, 0.0, 0.0))
        s.textures.append(colour); m.textures[TEX_TRANSMISSION_COLOR] = TextureBinding(texture=len(s.textures) - 1, bias=cb)
    wall = MaterialDesc.open_pbr(base_color=(0.2, 0.6, 0.9))
    s.materials = [m, wall]
    ball = MeshDesc(name="/Ball", vertices=bake_vertices(pts, pts, uv), faces=faces, material=0)
    if primvar:  # the same inputs from scene data: a per-vertex weight and a constant colour
        ball.primvars = [Primvar("tw", 0, 3, (pts[:, 2] > 0).astype(np.float32)), Primvar("tcol", 2, 0, np.float32([0.3, 0.9, 0.5]))]
        m.primvar_inputs = {TEX_TRANSMISSION_WEIGHT: "tw", TEX_TRANSMISSION_COLOR: "tcol"}
    q = np.float32([[-3, 2.5, -3], [3, 2.5, -3], [3, 2.5, 3], [-3, 2.5, 3]])
    s.meshes = [ball, MeshDesc(name="/Wall", vertices=bake_vertices(q, np.float32([[0, -1, 0]] * 4)), faces=np.uint32([[0, 1, 2], [0, 2, 3]]), material=1, double_sided=True)]
    s.rect_lights = [RectLight(origin=(0.0, -2.0, 3.0), t0=(1, 0, 0), t1=(0, 1, 0), base_emission=(30, 30, 30), width=1.0, height=1.0)]
    s.camera = CameraDesc(position=(0.0, -3.2, 0.0), forward=(0.0, 1.0, 0.0), up=(0.0, 0.0, 1.0), vfov=0.7)
    return s


def _const_tex(*rgba):
    """(texture, bias) whose lookup is exactly `rgba` everywhere: black texels (a bilinear blend of equal non-zero texels need not return them bit for bit) + the binding's bias"""
    return np.zeros((2, 2, 4), np.float32), tuple(float(x) for x in rgba)


def test_textured_transmission_inputs(orc):
    """OpenPBR transmission_weight / transmission_color as textured inputs (open_pbr_surface.mtlx:29, 31; VERDICT r03 missing #5).  A texture that holds one value IS
    that constant, bit for bit (weight and colour, alone and together); a weight map that is 1 on the upper half of a ball and 0 on the lower makes the upper half
    glass and leaves the lower half the opaque material; with a transmission_depth the colour belongs to the MEDIUM, which keeps the material's constant (a colour
    texture then changes nothing), while the weight map still selects where the path may enter it."""
    rs = RenderSettings(spp=8, max_bounces=6, next_event_estimation=True, progressive_accumulation=False)
    w = h = 36
    ref, _ = orc.render(_glass_ball(tw=0.7, tc=(0.3, 0.9, 0.5)), rs, w, h, threads=8)
    a, _ = orc.render(_glass_ball(weight=_const_tex(0.7, 0.7, 0.7, 1.0), tc=(0.3, 0.9, 0.5)), rs, w, h, threads=8)
    b, _ = orc.render(_glass_ball(colour=_const_tex(0.3, 0.9, 0.5, 1.0), tw=0.7), rs, w, h, threads=8)
    c, _ = orc.render(_glass_ball(weight=_const_tex(0.7, 0.7, 0.7, 1.0), colour=_const_tex(0.3, 0.9, 0.5, 1.0), tw=0.1, tc=(1, 0, 0)), rs, w, h, threads=8)
    for img in (a, b, c):
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    # half glass / half opaque
    half = np.zeros((8, 4, 4), np.float32); half[:4] = 1.0      # v < 0.5 (the upper hemisphere, v = acos(z) / pi) transmits
    opaque, _ = orc.render(_glass_ball(tw=0.0), rs, w, h, threads=8)
    glass, _ = orc.render(_glass_ball(tw=1.0), rs, w, h, threads=8)
    mixed, _ = orc.render(_glass_ball(weight=half), rs, w, h, threads=8)
    top, bottom = slice(h // 2 - 9, h // 2 - 3), slice(h // 2 + 3, h // 2 + 9)
    cols = slice(w // 2 - 4, w // 2 + 4)
    d = lambda x, y, rows: float(np.abs(x[rows, cols] - y[rows, cols]).mean())
    span = min(d(glass, opaque, top), d(glass, opaque, bottom))
    glass_rows, opaque_rows = (top, bottom) if d(mixed, glass, top) < d(mixed, opaque, top) else (bottom, top)   # (whichever way the image's rows run)
    assert d(mixed, opaque, opaque_rows) < 0.25 * span and d(mixed, glass, glass_rows) < 0.5 * span, (d(mixed, opaque, opaque_rows), d(mixed, glass, glass_rows), span)
    # scene data drives the same inputs
    pv, _ = orc.render(_glass_ball(primvar=True), rs, w, h, threads=8)
    assert np.isfinite(pv).all() and d(pv, opaque, glass_rows) > 0.25 * span and d(pv, opaque, opaque_rows) < 0.25 * span   # per-vertex weight: z > 0 transmits
    # with a depth the colour is the medium's: constant per material (medium stack on and off)
    for stack in (0, 2):
        rs2 = RenderSettings(spp=4, max_bounces=6, next_event_estimation=True, progressive_accumulation=False, medium_stack_size=stack)
        d0, _ = orc.render(_glass_ball(tw=1.0, tc=(0.3, 0.9, 0.5), depth=0.5), rs2, w, h, threads=8)
        d1, _ = orc.render(_glass_ball(tw=1.0, tc=(0.3, 0.9, 0.5), depth=0.5, colour=_const_tex(0.9, 0.1, 0.1, 1.0)), rs2, w, h, threads=8)
        assert np.array_equal(d0.view(np.uint32), d1.view(np.uint32)), stack
