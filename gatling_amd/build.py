"""Builds the product library ``gatling_amd/libgatling_gi.so`` for gfx950 with hipcc (in-tree, so it travels to
the GPU box with the repo snapshot)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libgatling_gi.so")
SOURCES = ["gi_c.cpp", "gi_image.cpp", "bvh8.cpp", "gi_kernels.hip", "gtl_shim.cpp"]
HEADERS = ["gi_types.h", "gi_kernels.h", "gi_device_math.h", "gi_queues.h", "gi_traversal.h", "gi_shading.h", "gi_image.h", "bvh8.h", os.path.join("..", "..", "include", "gi_c.h"),
           os.path.join("..", "..", "include", "gtl", "gi", "Gi.h")]
# -ffp-contract=off: arithmetic contract.  No fast-math: IEEE divide/sqrt are part of it.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc] + FLAGS + ["-o", LIB] + SOURCES + ["-lz"]  # zlib: PNG inflate (gi_image.cpp)
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
