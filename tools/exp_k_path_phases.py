import os, sys
sys.path.insert(0, '/root/repo')
os.environ["GATLING_OPTIONS"] = "phase_stats=1"
from gatling_amd import capi
from gatling_amd.scene import RenderSettings
from gatling_amd.scenes import cornell_box
sc = capi.Scene(cornell_box())
rs = RenderSettings(spp=64, max_bounces=8, progressive_accumulation=False)
sc.render(rs, 1920, 1080)
sc.set_option(capi.OPTION_COUNT_TRAVERSAL, 1)
sc.render(rs, 1920, 1080)
print(sc.stats())
sc.close()
