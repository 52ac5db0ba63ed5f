"""One process, one scene build, many option variants of the render loop (gi_options.h: $GATLING_OPTIONS is read per giCRender).

  python tools/gpu_variants.py c3 16 - trace_dyn=0 trace_dyn=32,trace_dyn_spill8=1        ("-" = defaults)

Prints per variant: Msamples/s, wall ms, per-stage ms (HIP events, every 4th iteration), and whether the image is
bit-identical to the first variant's (it must be: the knobs change scheduling, never arithmetic)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload  # noqa: E402
from gatling_amd import capi  # noqa: E402


def main():
    workload, spp = sys.argv[1], int(sys.argv[2])
    variants = [a if a != "-" else "" for a in sys.argv[3:]] or [""]
    desc, rs, w, h, label = make_workload(workload, spp)
    rs.progressive_accumulation = False
    t0 = time.perf_counter()
    scene = capi.Scene(desc)
    scene.set_option(capi.OPTION_KERNEL_TIMERS, 4)
    ref = None
    for v in variants:
        os.environ["GATLING_OPTIONS"] = v
        img = scene.render(rs, w, h)  # warm-up (first variant: includes scene build + upload)
        if ref is None:
            print(f"# {label}: first render (with build) {time.perf_counter() - t0:.1f} s", flush=True)
        t1 = time.perf_counter()
        img = scene.render(rs, w, h)
        dt = time.perf_counter() - t1
        st = scene.stats()
        same = True if ref is None else bool(np.array_equal(ref, img))
        if ref is None:
            ref = img.copy()
        print(v, f"value {w * h * rs.spp / dt / 1e6:.1f} ms {dt * 1e3:.1f}",
              {k: round(st[k], 1) for k in ("raygenMs", "traceMs", "shadeMs", "shadowMs")}, "iters", st["iterations"], "identical", same, flush=True)
    scene.set_option(capi.OPTION_COUNT_TRAVERSAL, 1)
    scene.render(rs, w, h)
    c = scene.stats()
    print("# segments", c["segments"], "nodes/ray", c["nodesVisited"] / max(1, c["segments"]), "tris/ray", c["trisTested"] / max(1, c["segments"]),
          "shadow rays", c["shadowRays"], "shadow nodes/ray", c["shadowNodesVisited"] / max(1, c["shadowRays"]),
          "shadow tris/ray", c["shadowTrisTested"] / max(1, c["shadowRays"]), "nodes", c["nodeCount"], "tris", c["triangleCount"])
    scene.close()


if __name__ == "__main__":
    main()
