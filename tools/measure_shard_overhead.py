#!/usr/bin/env python3
"""How well does one rank's share of the C2 frame scale down?  Renders rows 0::N of the 1920x1080 frame (what rank 0 of an N-GPU run
does, device-only, no gather) for N = 1, 2, 4, 8 on ONE GPU and prints time per frame and the implied strong-scaling efficiency
t(1) / (N * t(N)) -- the part of the 1 -> 8 curve that does not depend on RCCL."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatling_amd import capi  # noqa: E402
from gatling_amd.scene import RenderSettings  # noqa: E402
from gatling_amd.scenes import cornell_box  # noqa: E402

w, h = 1920, 1080
rs = RenderSettings(spp=int(os.environ.get("SPP", "1024")), max_bounces=8)
rs.progressive_accumulation = False
sc = capi.Scene(cornell_box())
t1 = None
for n in (1, 2, 4, 8):
    sc.render(rs, w, h, rows=(0, h), device_only=True, row_stride=n)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); sc.render(rs, w, h, rows=(0, h), device_only=True, row_stride=n); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    t1 = t1 or t
    st = sc.stats()
    print(f"N={n}: {t * 1e3:8.2f} ms per frame share, kernel {st['traceMs']:.2f} ms (0 without timers), efficiency {t1 / (n * t):.3f}", flush=True)
sc.close()
