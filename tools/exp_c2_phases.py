"""C2 cost split: time per step against max_bounces (segments per sample) -> per-sample (regeneration, camera ray, finish) and per-segment cost.

Measured (r02, 1 x MI355X, spp 256): max_bounces 1 / 2 / 3 / 4 / 6 / 8 -> 1.000 / 1.377 / 1.663 / 1.896 / 2.141 / 2.204 segments per sample,
43.3 / 63.9 / 83.0 / 98.7 / 117.1 / 121.3 ps per sample: the first segment (coherent camera ray, regeneration and finish included) costs 43 ps, every
further one 65 ps -- secondary rays, not regeneration, are where k_path_bw's time goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatling_amd import capi
from gatling_amd.scene import RenderSettings
from gatling_amd.scenes import cornell_box
w, h, spp = 1920, 1080, 256
sc = capi.Scene(cornell_box())
for mb in (1, 2, 3, 4, 6, 8):
    rs = RenderSettings(spp=spp, max_bounces=mb, progressive_accumulation=False)
    sc.render(rs, w, h)
    t0 = time.perf_counter(); sc.render(rs, w, h); dt = time.perf_counter() - t0
    st = sc.stats()
    print(f"max_bounces {mb}: {dt*1e3:8.2f} ms  segments/sample {st['segments']/st['samples']:.3f}  ps/sample {dt/ (w*h*spp) * 1e12:7.1f}", flush=True)
sc.close()
