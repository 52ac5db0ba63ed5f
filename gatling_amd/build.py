"""Builds the product library ``gatling_amd/libgatling_gi.so`` for gfx950 with hipcc (in-tree, so it travels to
the GPU box with the repo snapshot).  Every translation unit is compiled to an object on its own (in parallel, and only
when it or a header changed), then linked."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJDIR = os.path.join(CSRC, "build")
LIB = os.path.join(_HERE, "libgatling_gi.so")
# the host side of the C ABI (gi_c.cpp until round 6): built with -fvisibility=hidden, the API keeps default visibility through gi_host.h
HOST_SOURCES = ["gi_c.cpp", "gi_scene.cpp", "gi_textures.cpp", "gi_lights.cpp", "gi_build.cpp", "gi_render.cpp", "gi_debug.cpp"]
SOURCES = HOST_SOURCES + ["gi_image.cpp", "bvh8.cpp", "gi_kernels.hip", "gi_trace.hip", "gi_shade.hip", "gi_aov.hip", "gi_path.hip", "gi_path_bw.hip", "gtl_shim.cpp"]
HEADERS = ["gi_host.h", "gi_types.h", "gi_kernels.h", "gi_device_math.h", "gi_queues.h", "gi_traversal.h", "gi_texture.h", "gi_shading.h", "gi_stages.h", "gi_image.h", "gi_options.h", "bvh8.h",
           os.path.join("..", "..", "include", "gi_c.h"), os.path.join("..", "..", "include", "gtl", "gi", "Gi.h"),
           os.path.join("..", "..", "include", "gtl", "gb", "ParamTypes.h")]
# -ffp-contract=off: arithmetic contract.  No fast-math: IEEE divide/sqrt are part of it.
INCLUDE = os.path.join(_HERE, "..", "include")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-I", INCLUDE]
# Kernel translation units: no SLP vectorisation.  The vectoriser packs pairs of independent fp32 operations into v_pk_mul / v_pk_add / v_pk_fma, which on
# gfx950 issue at the rate of the scalar forms (tools/valu_calib.hip) but need their operands in aligned register pairs -- the packing costs v_mov's (k_path_bw:
# 313 -> 153 static moves, k_trace_dyn 238 -> 161) and registers.  Measured (r03c): C2 7 034 -> 7 506 Msamples/s at spp 128, C3 / C4 unchanged.  Results are
# bit-identical (packed and scalar fp32 operations round the same way); the hand-written packed fma of the box test stays.
KERNEL_FLAGS = ["-fno-slp-vectorize"]


def materialx_include():
    """Directory holding <MaterialXFormat/XmlIo.h>, or None.  gtl_shim_mtlx.cpp (giCreateMaterialFromMtlxDoc) is built only when it exists:
    $MATERIALX_ROOT/include, an OpenUSD install ($PXR_USD_LOCATION / $USD_ROOT) or the system include path."""
    roots = [os.environ.get(k) for k in ("MATERIALX_ROOT", "PXR_USD_LOCATION", "USD_ROOT")] + ["/usr", "/usr/local", "/opt/local"]
    for r in roots:
        if r and os.path.exists(os.path.join(r, "include", "MaterialXFormat", "XmlIo.h")):
            return os.path.join(r, "include")
    return None


def _mtime(path: str) -> float:
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _headers_mtime() -> float:
    return max([_mtime(os.path.join(CSRC, h)) for h in HEADERS] + [_mtime(os.path.abspath(__file__))])


def _obj(src: str) -> str:
    return os.path.join(OBJDIR, src + ".o")


def _stale(src: str, force: bool) -> bool:
    return force or _mtime(_obj(src)) < max(_mtime(os.path.join(CSRC, src)), _headers_mtime())


def needs_build() -> bool:
    return _mtime(LIB) == 0.0 or any(_stale(s, False) or _mtime(_obj(s)) > _mtime(LIB) for s in SOURCES)


def build(force: bool = False, verbose: bool = False) -> str:
    if not (force or needs_build()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJDIR, exist_ok=True)

    def compile_one(src: str) -> None:
        cmd = [hipcc] + FLAGS + (KERNEL_FLAGS if src.endswith(".hip") else []) + (["-fvisibility=hidden"] if src in HOST_SOURCES else []) + (["-I", mtlx] if mtlx and src == "gtl_shim_mtlx.cpp" else []) + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)

    sources = list(SOURCES)
    mtlx = materialx_include()
    if mtlx:
        sources.append("gtl_shim_mtlx.cpp")
    todo = [s for s in sources if _stale(s, force)]
    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 2) or 1) as pool:
        list(pool.map(compile_one, todo))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(s) for s in sources] + ["-lz"]  # zlib: PNG inflate (gi_image.cpp)
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
