"""The .usda emitter for the generated configurations against the harness's .usda reader: geometry,
materials, primvar bindings, instancing, cameras and all light types survive the trip; a scene without composed transforms
renders bit-identically (oracle) after it."""
import dataclasses

import numpy as np
import pytest

from gatling_amd.scene import RenderSettings
from gatling_amd.scenes import cornell_box, interior_scene, random_triangle_soup, sphere_grid, textured_scene, volume_scene
from gatling_amd.usda import load_usda, parse_usda
from gatling_amd.usda_writer import write_usda

SCENES = {
    "cornell": lambda: cornell_box(),
    "soup": lambda: random_triangle_soup(400, seed=3),
    "instances": lambda: sphere_grid(grid=3, subdivisions=1, material_count=4),
    "interior": lambda: interior_scene(clutter_instances=12, subdivisions=1, prototypes=3, material_count=6),
    "textured+dome": lambda: textured_scene(dome=True),
    "volume": lambda: volume_scene(),
}


def _close(a, b, tol=1e-5):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() <= tol


def _world(m):
    return np.einsum("ij,kjl->kil", np.asarray(m.transform, np.float64), np.asarray(m.instance_transforms, np.float64).reshape(-1, 4, 4))


@pytest.mark.parametrize("name", sorted(SCENES))
def test_usda_round_trip(tmp_path, name):
    a = SCENES[name]()
    write_usda(tmp_path / "s.usda", a)
    b = load_usda(str(tmp_path / "s.usda"))
    assert len(a.meshes) == len(b.meshes) and len(a.materials) == len(b.materials)
    for m, n in zip(a.meshes, b.meshes):
        for f in ("pos", "norm", "u", "v", "tangent", "bitangentSign"):  # %.9g round-trips float32 exactly
            assert np.array_equal(m.vertices[f], n.vertices[f]), (m.name, f)
        assert np.array_equal(np.asarray(m.faces, np.uint32), n.faces)
        assert _world(m).shape == _world(n).shape and _close(_world(m), _world(n)), m.name  # prim * instance, composed in the file
        assert (m.material, m.double_sided, m.left_handed, m.visible) == (n.material, n.double_sided, n.left_handed, n.visible)
        assert [(p.name, p.type, p.interpolation) for p in m.primvars if p.interpolation != 1] == [(p.name, p.type, p.interpolation) for p in n.primvars]
        for p, q in zip([p for p in m.primvars if p.interpolation != 1], n.primvars):
            assert np.array_equal(np.asarray(p.data, np.float32).reshape(-1), q.data.reshape(-1))
    for m, n in zip(a.materials, b.materials):
        assert m.klass == n.klass and _close(m.params, n.params, 1e-6), m.name
        assert m.primvar_inputs == n.primvar_inputs
    for kind in ("sphere_lights", "distant_lights", "rect_lights", "disk_lights"):
        la, lb = getattr(a, kind), getattr(b, kind)
        assert len(la) == len(lb)
        for x, y in zip(la, lb):
            for f in dataclasses.fields(x):
                assert _close(getattr(x, f.name), getattr(y, f.name)), (kind, f.name)
    for f in dataclasses.fields(a.camera):
        assert _close(getattr(a.camera, f.name), getattr(b.camera, f.name)), f.name
    assert (a.dome_light is None) == (b.dome_light is None)
    if a.dome_light is not None:
        qa, qb = np.asarray(a.dome_light.rotation, np.float64), np.asarray(b.dome_light.rotation, np.float64)
        assert _close(qa, qb) or _close(qa, -qb)  # q and -q are the same rotation
        assert _close(a.dome_light.base_emission, b.dome_light.base_emission)
        src, got = a.textures[a.dome_light.texture], b.textures[b.dome_light.texture]
        assert got.shape == src.shape and np.all(np.abs(got[..., :3] - src[..., :3]) <= src[..., :3].max(axis=2, keepdims=True) / 128.0 + 1e-6)  # RGBE mantissas


def test_usda_structure(tmp_path):
    """What a USD reader needs to find: header, defaultPrim, native instancing through a class prototype, bindings."""
    desc = sphere_grid(grid=2, subdivisions=1, material_count=2)
    write_usda(tmp_path / "s.usda", desc)
    text = open(tmp_path / "s.usda").read()
    assert text.startswith("#usda 1.0\n(") and 'defaultPrim = "Root"' in text
    root = parse_usda(text)
    top = root.children[0]
    assert top.type == "Xform" and top.name == "Root"
    protos = [c for c in top.children if c.specifier == "class"]
    insts = [g for c in top.children for g in c.children if g.meta.get("instanceable") == "true"]
    assert len(protos) == len(desc.meshes) and len(insts) == sum(len(m.instance_transforms) for m in desc.meshes)
    assert all(i.meta["inherits"][1] in {p.path for p in protos} for i in insts)
    mats = next(c for c in top.children if c.name == "Materials").children
    assert [m.type for m in mats] == ["Material"] * len(desc.materials)
    assert all(any(k.endswith("surface.connect") for k in m.attrs) for m in mats)


def test_round_tripped_scene_renders_identically(orc, tmp_path):
    """No composed transforms in the soup scene, so every float survives and the oracle image is the same bit for bit."""
    a = random_triangle_soup(300, seed=9)
    write_usda(tmp_path / "s.usda", a)
    b = load_usda(str(tmp_path / "s.usda"))
    rs = RenderSettings(spp=2, max_bounces=4, next_event_estimation=True)
    ia, _ = orc.render(a, rs, 48, 27)
    ib, _ = orc.render(b, rs, 48, 27)
    assert np.array_equal(ia, ib)


def test_usda_keeps_the_coat_tangent_turn(tmp_path):
    """geometry_coat_tangent as the [ext] coat_rotation float of the flat OpenPBR vocabulary (usda_writer OPEN_PBR_INPUTS, gtl_shim.cpp kOpbr): the block comes back bit for bit."""
    from gatling_amd.scene import P_COAT_ROTATION, MaterialDesc
    a = SCENES[sorted(SCENES)[0]]()
    a.materials[0] = MaterialDesc.open_pbr(name=a.materials[0].name, coat_weight=0.6, coat_roughness=0.3, coat_roughness_anisotropy=0.4, coat_rotation=0.3137,
                                           specular_roughness_anisotropy=0.2, specular_rotation=-0.41)
    write_usda(tmp_path / "s.usda", a)
    b = load_usda(str(tmp_path / "s.usda"))
    assert b.materials[0].params[P_COAT_ROTATION] == np.float32(0.3137)
    assert np.array_equal(np.asarray(a.materials[0].params, np.float32).view(np.uint32), np.asarray(b.materials[0].params, np.float32).view(np.uint32))
