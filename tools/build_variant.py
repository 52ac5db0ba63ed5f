#!/usr/bin/env python3
"""Builds a VARIANT of the product library for A/B experiments on the GPU box: the kernel translation units recompiled with extra
flags / macros, linked with the host objects of the regular build, into gatling_amd/variants/libgatling_gi_<name>.so
(git-ignored, travels with gpurun).  Select it with GATLING_GI_LIB=<path>.

  python tools/build_variant.py noslp -fno-slp-vectorize
  python tools/build_variant.py w4 -DGI_DYN_WAVES=4
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gatling_amd import build as B  # noqa: E402

KERNEL_TUS = ["gi_kernels.hip", "gi_trace.hip", "gi_shade.hip", "gi_aov.hip", "gi_path.hip", "gi_path_bw.hip"]


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    B.build()  # host objects (and the regular library) are current
    out_dir = os.path.join(ROOT, "gatling_amd", "variants")
    obj_dir = os.path.join(B.OBJDIR, "variant_" + name)
    os.makedirs(out_dir, exist_ok=True); os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def compile_one(src):
        base = [f for f in B.KERNEL_FLAGS if not ("-fslp-vectorize" in extra and f == "-fno-slp-vectorize")]
        subprocess.check_call([hipcc] + B.FLAGS + base + extra + ["-c", src, "-o", os.path.join(obj_dir, src + ".o")], cwd=B.CSRC)

    with ThreadPoolExecutor(max_workers=6) as pool:
        list(pool.map(compile_one, KERNEL_TUS))
    objs = [os.path.join(obj_dir, s + ".o") if s in KERNEL_TUS else B._obj(s) for s in B.SOURCES]
    lib = os.path.join(out_dir, f"libgatling_gi_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lz"], cwd=B.CSRC)
    print(lib)


if __name__ == "__main__":
    main()
