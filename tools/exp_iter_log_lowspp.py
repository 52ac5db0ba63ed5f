import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["GATLING_ITER_LOG"] = "1"
from bench import make_workload
from gatling_amd import capi
from gatling_amd.scene import RenderSettings
workload, spp = sys.argv[1], int(sys.argv[2])
desc, rs0, w, h, label = make_workload(workload)
rs = RenderSettings(spp=spp, next_event_estimation=rs0.next_event_estimation)
sc = capi.Scene(desc)
for _ in range(3): sc.render(rs, w, h)
sc.set_option(capi.OPTION_KERNEL_TIMERS, 1)
print("#", label, "spp", spp, flush=True)
sc.render(rs, w, h)
print("#", {k: round(v, 2) if isinstance(v, float) else v for k, v in sc.stats().items() if k.endswith("Ms") or k in ("iterations", "segments")})
sc.close()
