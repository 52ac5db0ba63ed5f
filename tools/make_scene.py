"""Emits a BASELINE.json configuration (c1..c5, bench.py's generators and seeds) as the flat binary the C harness loads and/or
as .usda for the reference's own CLI:

  python tools/make_scene.py c3 out/c3.gscn out/c3.usda [--spp N]
  tools/gi_render out/c3.gscn out/c3.pfm --stats                                   # this repo, plain C over the C ABI
  gatling out/c3.usda out/c3.png --image-width 1920 --image-height 1080 --spp 256 --max-bounces 8   # the reference, where it exists
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload  # noqa: E402
from gatling_amd.scenefile import save_scene  # noqa: E402
from gatling_amd.usda_writer import write_usda  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else None
    if spp is not None:
        args.remove(str(spp))
    if len(args) < 2:
        raise SystemExit(__doc__)
    desc, rs, w, h, label = make_workload(args[0], spp)
    print(label)
    for out in args[1:]:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        if out.endswith(".usda"):
            write_usda(out, desc, aspect=w / h)
        else:
            save_scene(out, desc, rs, w, h)
        print(f"wrote {out} ({os.path.getsize(out) / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
