"""tools/compare_reference.py -- the SURVEY 8d(ii) comparator against a reference-path image -- validated on the CPU: the oracle stands in for both sides.
Two independent renders of the same scene (disjoint sample offsets) pass against a third; a "reference" whose diffuse weights are 3 % off fails."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import compare_reference as cr  # noqa: E402
from gatling_amd.scene import RenderSettings  # noqa: E402
from gatling_amd.scenes import cornell_box  # noqa: E402
from oracle import orc  # noqa: E402

W, H, SPP = 48, 27, 1536


def _render(desc, offset):
    rs = RenderSettings(spp=SPP, max_bounces=8)
    rs.progressive_accumulation = False
    img, _ = orc.render(desc, rs, W, H, sample_offset=offset, threads=os.cpu_count() or 4)
    return img


def test_comparator_accepts_an_independent_render_and_rejects_a_biased_one(tmp_path):
    desc = cornell_box()
    a, b, ref = _render(desc, 0), _render(desc, SPP), _render(desc, 2 * SPP)
    ok = cr.compare(a, b, ref, False)
    assert ok["pass"] and max(ok["rmse_over_standard_error"]) < 1.7, ok       # two unbiased estimates: sqrt(2) standard errors apart
    assert ok["mean_luminance_rel_error"] < 0.005
    # the reference's own file format: clipped, gamma-encoded, quantised to 8 bits (hdGatling/main.cpp:463-487)
    ok8 = cr.compare(a, b, cr.to_srgb8(ref[..., :3]), True)
    assert ok8["pass"], ok8
    assert ok8["srgb8_differing_bytes"] > 0  # Monte-Carlo noise: the byte-equality criterion of the reference's tests cannot hold between two renderers
    # a renderer whose diffuse weights are 3 % low is not the same renderer
    biased = copy.deepcopy(desc)
    for m in biased.materials:
        m.params = np.array(m.params, np.float32, copy=True)
        m.params[0:3] *= np.float32(0.97)  # diffuseColor (GI_C_P_DIFFUSE_COLOR)
    bad = cr.compare(a, b, _render(biased, 2 * SPP), False)
    assert not bad["pass"], bad
    assert bad["mean_luminance_rel_error"] > 0.005
    # identical inputs: zero error whatever the noise
    same = cr.compare(a, b, a, False)
    assert same["pass"] and max(same["rmse"]) == 0.0 and same["srgb8_differing_bytes"] == 0
