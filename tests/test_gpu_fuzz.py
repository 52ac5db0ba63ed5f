"""A fixed sample of the differential campaign (tests/fuzz_parity.py) in the GPU suite: random renders -- geometry, materials with every optional lobe, textures,
all light types, cameras, render settings drawn from one seed each (tests/fuzz_scenes.py) -- through the C ABI == the oracle, bit for bit.  $GATLING_FUZZ_SEEDS
("first:last") widens the sample; the campaigns that were run are logged under profiles/ (r06*_fuzz_parity.log)."""
import os

import pytest

from fuzz_parity import bsdf_case, run_case

_FIRST, _LAST = (int(x) for x in os.environ.get("GATLING_FUZZ_SEEDS", "0:96").split(":"))


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(_FIRST, _LAST, 16))
def test_random_renders_match_the_oracle(gi, orc, block):
    failures = []
    for seed in range(block, min(block + 16, _LAST)):
        r = run_case(gi, orc, seed, threads=min(32, os.cpu_count() or 8))
        if r["status"] != "same":
            failures.append(f"seed {seed}: {r['status']}: {r['detail']}")
    assert not failures, "\n".join(failures)



@pytest.mark.gpu
def test_random_materials_on_random_frames_match_the_oracle(gi, orc):
    """The BSDF half of the campaign (`tests/fuzz_parity.py --bsdf`): 48 random materials x 2 048 frames / directions / random numbers each, grazing and below-horizon
    directions and random numbers at 0 and 1 - 2^-24 included: giCDebugEvalBsdf == the oracle's entry points, all 15 outputs, bit for bit."""
    failures = [f"seed {r['seed']}: {r['detail'][:600]}" for r in (bsdf_case(gi, orc, seed) for seed in range(48)) if r["status"] != "same"]
    assert not failures, "\n".join(failures)
