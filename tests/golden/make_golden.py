"""Generates the committed golden fixtures in this directory.

The reference cannot run here or on the GPU box (Vulkan-RT + OpenUSD + MDL SDK) and its own
reference images are git-LFS stubs, so these goldens are outputs of the CPU oracle (oracle/gi_oracle.cpp), i.e.
regression anchors for the oracle AND parity targets for the HIP path -- "parity unpinned" against the Vulkan path.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gatling_amd.scene import MAT_DIFFUSE, MAT_USD_PREVIEW_SURFACE, RectLight, RenderSettings, SphereLight  # noqa: E402
from gatling_amd.scenes import cornell_box, interior_scene, leaf_card_scene, sphere_grid, textured_scene, volume_scene  # noqa: E402
from oracle import orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (material class, width, height, settings kwargs, lights)
    "cornell_diffuse_48x27_spp8_b4": (MAT_DIFFUSE, 48, 27, dict(spp=8, max_bounces=4), None),
    "cornell_ups_48x27_spp8_b8": (MAT_USD_PREVIEW_SURFACE, 48, 27, dict(spp=8, max_bounces=8), None),
    "cornell_ups_nee_48x27_spp4_b5": (MAT_USD_PREVIEW_SURFACE, 48, 27, dict(spp=4, max_bounces=5, next_event_estimation=True), "rect+sphere"),
    # scene generators instead of cornell: (generator name, width, height, settings kwargs, generator kwargs)
    "textured_dome_64x36_spp4_b6": ("textured", 64, 36, dict(spp=4, max_bounces=6, next_event_estimation=True), dict()),
    "volume_stack2_64x36_spp4_b12": ("volume", 64, 36, dict(spp=4, max_bounces=12, next_event_estimation=True, medium_stack_size=2), dict()),
    "instances_openpbr_64x36_spp4_b6": ("spheres", 64, 36, dict(spp=4, max_bounces=6), dict(grid=4, subdivisions=1, material_count=8)),
    "leaf_cards_64x36_spp4_b5": ("leaves", 64, 36, dict(spp=4, max_bounces=5, next_event_estimation=True, rr_bounce_offset=0), dict()),
    "interior_64x36_spp3_b6": ("interior", 64, 36, dict(spp=3, max_bounces=6, next_event_estimation=True), dict(clutter_instances=40, subdivisions=1, prototypes=4, material_count=10)),
}
GENERATORS = {"textured": textured_scene, "volume": volume_scene, "spheres": sphere_grid, "interior": interior_scene, "leaves": leaf_card_scene}


def build_case(name):
    klass, w, h, kw, lights = CASES[name]
    if isinstance(klass, str):
        return GENERATORS[klass](**lights), RenderSettings(**kw), w, h
    desc = cornell_box(klass)
    if lights == "rect+sphere":
        desc.rect_lights.append(RectLight(origin=(0.0, 0.0, 0.9), t0=(1, 0, 0), t1=(0, -1, 0), base_emission=(12, 12, 10), width=0.6, height=0.6))
        desc.sphere_lights.append(SphereLight(pos=(-0.5, -0.5, 0.2), base_emission=(4, 2, 1), radius=(0.1, 0.1, 0.1)))
    return desc, RenderSettings(**kw), w, h


if __name__ == "__main__":
    for name in CASES:
        desc, rs, w, h = build_case(name)
        img, cnt = orc.render(desc, rs, w, h, threads=4)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), color=img, segments=np.uint64(cnt["segments"]),
                            shadow_rays=np.uint64(cnt["shadow_rays"]))
        print(name, img.shape, img[..., :3].mean(), cnt["segments"], cnt["shadow_rays"])
