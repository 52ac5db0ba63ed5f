"""Per-iteration stage times and queue sizes of the wavefront pipeline (GATLING_ITER_LOG=1; gi_c.cpp prints one line per iteration to stderr).

  python tools/exp_iter_log.py c4 256      # workload, spp
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GATLING_ITER_LOG"] = "1"
from bench import make_workload  # noqa: E402
from gatling_amd import capi  # noqa: E402

workload, spp = sys.argv[1], int(sys.argv[2])
desc, rs, w, h, label = make_workload(workload, spp)
rs.progressive_accumulation = False
sc = capi.Scene(desc)
sc.render(rs, w, h)                      # build + warm-up (no timers: no log)
sc.set_option(capi.OPTION_KERNEL_TIMERS, 1)
print("#", label, flush=True)
sc.render(rs, w, h)
print("#", {k: round(v, 2) if isinstance(v, float) else v for k, v in sc.stats().items() if k.endswith("Ms") or k in ("iterations", "segments")})
sc.close()
