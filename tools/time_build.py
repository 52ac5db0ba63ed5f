"""Host-side scene build time (flatten + BVH8) of C5 / C3 on this box: GATLING_BUILD_TIMING=1 python tools/time_build.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatling_amd import capi
from gatling_amd.scene import RenderSettings
from gatling_amd.scenes import interior_scene, random_triangle_soup
for name, make in (("c3 soup 1M", lambda: random_triangle_soup()), ("c5 interior 10.24M", lambda: interior_scene())):
    t0 = time.perf_counter(); desc = make(); t1 = time.perf_counter()
    sc = capi.Scene(desc); t2 = time.perf_counter()
    sc.render(RenderSettings(spp=1, max_bounces=1), 64, 36); t3 = time.perf_counter()
    print(f"{name}: python scene description {t1-t0:.1f} s, upload {t2-t1:.1f} s, first render incl. BVH build {t3-t2:.1f} s, cores {os.cpu_count()}", flush=True)
    sc.close()
