"""First GPU bring-up: parity vs oracle at a few sizes + quick timing.  Writes gpurun_out/first.log"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gatling_amd import capi
from gatling_amd.scene import RenderSettings, MAT_DIFFUSE, MAT_USD_PREVIEW_SURFACE
from gatling_amd.scenes import cornell_box
from oracle import orc

def compare(desc, rs, w, h, tag):
    sc = capi.Scene(desc)
    sc.set_option(capi.OPTION_COUNT_TRAVERSAL, 1)
    t = time.time(); img = sc.render(rs, w, h); dt = time.time() - t
    st = sc.stats(); sc.close()
    ref, cnt = orc.render(desc, rs, w, h, threads=os.cpu_count())
    diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    ad = np.abs(img - ref)
    print(tag, f"{w}x{h} spp={rs.spp} b={rs.max_bounces}: bitdiff_pixels={int(diff.sum())}/{w*h} maxabs={ad.max():.3e} mean_gpu={img[...,:3].mean():.6f} mean_ref={ref[...,:3].mean():.6f}"
          f" seg gpu={st['segments']} ref={cnt['segments']} shadow gpu={st['shadowRays']} ref={cnt['shadow_rays']} nodes/ray={st['nodesVisited']/max(1,st['segments']):.2f} tris/ray={st['trisTested']/max(1,st['segments']):.2f}"
          f" iters={st['iterations']} render_ms={st['renderMs']:.2f} wall={dt:.3f}", flush=True)
    return img, ref

if __name__ == "__main__":
    for klass, name in ((MAT_DIFFUSE, "diffuse"), (MAT_USD_PREVIEW_SURFACE, "ups")):
        d = cornell_box(klass)
        compare(d, RenderSettings(spp=1, max_bounces=1), 32, 18, name)
        compare(d, RenderSettings(spp=4, max_bounces=4), 64, 36, name)
        compare(d, RenderSettings(spp=16, max_bounces=8), 256, 144, name)
    # timing at scale
    d = cornell_box(MAT_USD_PREVIEW_SURFACE)
    sc = capi.Scene(d)
    for spp in (4, 16, 64):
        rs = RenderSettings(spp=spp, max_bounces=8)
        sc.render(rs, 1920, 1080, device_only=True)
        st = sc.stats()
        print(f"1080p spp={spp}: render_ms={st['renderMs']:.1f} Msamples/s={st['samples']/st['renderMs']/1e3:.1f} seg/sample={st['segments']/st['samples']:.3f} iters={st['iterations']}", flush=True)
    sc.set_option(capi.OPTION_KERNEL_TIMERS, 1)
    rs = RenderSettings(spp=64, max_bounces=8)
    sc.render(rs, 1920, 1080, device_only=True)
    st = sc.stats(); print("timers:", json.dumps(st), flush=True)
    sc.close()
