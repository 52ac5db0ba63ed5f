"""Per-iteration stage times and queue sizes of the wavefront pipeline (GATLING_ITER_LOG=1; gi_render.cpp prints one line per iteration to stderr).

  python tools/exp_iter_log.py c4 256               # workload, spp (the workload's own settings)
  python tools/exp_iter_log.py c4 1 delegate        # hdGatling's defaults instead: 13 bounces, progressive accumulation (the spp-1 viewport frame)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GATLING_ITER_LOG"] = "1"
from bench import make_workload  # noqa: E402
from gatling_amd import capi  # noqa: E402
from gatling_amd.scene import RenderSettings  # noqa: E402

workload, spp = sys.argv[1], int(sys.argv[2])
desc, rs, w, h, label = make_workload(workload, spp)
if len(sys.argv) > 3 and sys.argv[3] == "delegate":
    rs = RenderSettings(spp=spp, next_event_estimation=rs.next_event_estimation)
else:
    rs.progressive_accumulation = False
sc = capi.Scene(desc)
for _ in range(3):
    sc.render(rs, w, h)                  # build + warm-up (no timers: no log)
sc.set_option(capi.OPTION_KERNEL_TIMERS, 1)
print("#", label, "spp", spp, flush=True)
sc.render(rs, w, h)
print("#", {k: round(v, 2) if isinstance(v, float) else v for k, v in sc.stats().items() if k.endswith("Ms") or k in ("iterations", "segments")})
sc.close()
