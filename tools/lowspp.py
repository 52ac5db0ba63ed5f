#!/usr/bin/env python3
"""The delegate's own default workload: one giRender per displayed frame at spp 1 (hdGatling defaults: spp 1, max bounces 13, Russian roulette from bounce 3,
progressive accumulation -- /root/reference/src/hdGatling/renderDelegate.cpp:93-110), 1920x1080.  Prints, per workload and spp, the milliseconds of one
giCRender call INCLUDING the D2H of the colour AOV (mean and minimum over the timed calls), the bounce-loop iterations of a call, the library's own stage timers
and Msamples/s.

  python tools/lowspp.py [c2,c3,c4] [1,4,16] [calls]        one JSON object per (workload, spp) on stdout"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload  # noqa: E402
from gatling_amd import capi  # noqa: E402
from gatling_amd.scene import RenderSettings  # noqa: E402


def measure(workload, spps, calls, quiet=False):
    desc, rs0, w, h, label = make_workload(workload)
    scene = capi.Scene(desc)
    out = []
    for spp in spps:
        rs = RenderSettings(spp=spp, next_event_estimation=rs0.next_event_estimation)  # everything else: the delegate's defaults (13 bounces, progressive accumulation)
        scene.set_option(capi.OPTION_KERNEL_TIMERS, 0)
        for _ in range(3):  # (the first call builds the scene / sizes the pool)
            scene.render(rs, w, h, copy=False)
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            scene.render(rs, w, h, copy=False)  # blocks; the colour AOV is in host memory on return
            ts.append((time.perf_counter() - t0) * 1e3)
        st = scene.stats()
        scene.set_option(capi.OPTION_KERNEL_TIMERS, 1)
        scene.render(rs, w, h, copy=False)
        tm = scene.stats()
        row = {"workload": workload, "spp": spp, "max_bounces": rs.max_bounces, "progressive": True, "width": w, "height": h, "calls": calls,
               "ms_per_call_mean": round(sum(ts) / len(ts), 3), "ms_per_call_min": round(min(ts), 3), "ms_per_call_max": round(max(ts), 3),
               "Msamples_per_s": round(w * h * spp / (sum(ts) / len(ts)) / 1e3, 1), "iterations": st["iterations"], "renderMs": round(st["renderMs"], 3),
               "batches": st.get("batches"), "poolSlots": st.get("poolSlots"),
               "stage_ms": {k: round(tm[k], 3) for k in ("raygenMs", "traceMs", "shadeMs", "shadowMs", "renderMs")}}
        if not quiet:
            print(json.dumps(row), flush=True)
        out.append(row)
    scene.close()
    return out


def main():
    wls = (sys.argv[1] if len(sys.argv) > 1 else "c3,c4").split(",")
    spps = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,4,16").split(",")]
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    for wl in wls:
        measure(wl, spps, calls)


if __name__ == "__main__":
    main()
