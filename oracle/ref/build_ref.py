#!/usr/bin/env python3
"""Builds oracle/_ref/libgi_ref.so: the PURE functions of the reference's own shader sources, compiled as C++ where they lie under
/root/reference, so that tests/test_oracle_ref.py can hold the oracle's restatements against the reference's code itself.

GLSL is not C++: the recipe (a) writes lightly rewritten copies of the needed shader files / functions into oracle/_ref/gen/ -- GENERATED,
git-ignored, never committed -- where the only edits are the ones the language difference forces (parameter qualifiers `in` / `out` /
`inout` become by-value / reference parameters, `#extension` lines go), and (b) compiles oracle/ref/ref_shim.cpp, which includes them
under oracle/ref/glsl_compat.h (`float` -> a strict-fp32 class, vec types, the built-ins these functions use).  Whole files: common.glsl,
aovs.glsl, colormap.glsl, rp_main_payload.glsl, mdl_shading_state.glsl, interface/rp_main.h; interface/gtl.h is replaced by the shim's macros.  Single functions, cut
out by name: fisGauss, russian_roulette, sampleDistance, sampleHenyeyGreensteinCos, sampleVolumeScatteringDirection (rp_main.rgen),
quatRotateDir (rp_main.miss), sampleLight (rp_main.chit), apply_wrap_and_crop, mdl_adapt_normal (mdl_interface.glsl).

Nothing here runs on the GPU box (no /root/reference there): the prebuilt .so travels with the repo snapshot.
    python oracle/ref/build_ref.py [--reference /root/reference]
"""
import argparse
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "_ref")
GEN = os.path.join(OUT, "gen")
LIB = os.path.join(OUT, "libgi_ref.so")

WHOLE = ["common.glsl", "aovs.glsl", "colormap.glsl", "rp_main_payload.glsl", "interface/rp_main.h", "mdl_shading_state.glsl"]
FUNCTIONS = {"rp_main.rgen": ["sampleDistance", "sampleHenyeyGreensteinCos", "sampleVolumeScatteringDirection", "russian_roulette", "fisGauss"],
             "rp_main.miss": ["quatRotateDir"], "rp_main.chit": ["sampleLight"], "mdl_interface.glsl": ["apply_wrap_and_crop", "mdl_adapt_normal"]}


def to_cpp(text: str) -> str:
    text = re.sub(r"^\s*#\s*extension[^\n]*\n", "\n", text, flags=re.M)
    text = re.sub(r"\b(?:inout|out)\s+(\w+)\s+(\w+)\s*([,)])", r"\1& \2\3", text)  # out / inout parameters -> references
    text = re.sub(r"\bin\s+(\w+)\s+(\w+)\s*([,)])", r"\1 \2\3", text)              # in parameters -> by value
    text = re.sub(r"(?<=[\w\)\]])\.(xy|yx|zw|xyz|r|g|b)\b(?!\s*\()", r".\1()", text)      # swizzle reads -> member functions (glsl_compat.h)
    return text


def cut_function(text: str, name: str) -> str:
    m = re.search(r"^[A-Za-z_]\w*\s+" + re.escape(name) + r"\s*\(", text, flags=re.M)
    if not m:
        raise SystemExit(f"function {name} not found")
    i = text.index("{", m.end())
    depth, j = 0, i
    while True:
        depth += {"{": 1, "}": -1}.get(text[j], 0)
        j += 1
        if depth == 0:
            break
    return text[m.start():j] + "\n"


def build(reference="/root/reference", verbose=False) -> str:
    shaders = os.path.join(reference, "src", "gi", "shaders")
    if not os.path.isdir(shaders):
        raise FileNotFoundError(shaders)
    os.makedirs(os.path.join(GEN, "interface"), exist_ok=True)
    open(os.path.join(GEN, "interface", "gtl.h"), "w").write("/* stub: ref_shim.cpp defines the GLSL side of interface/gtl.h (its C++ side needs glm) */\n")
    for rel in WHOLE:
        open(os.path.join(GEN, rel), "w").write(to_cpp(open(os.path.join(shaders, rel)).read()))
    for rel, names in FUNCTIONS.items():
        src = open(os.path.join(shaders, rel)).read()
        for n in names:
            open(os.path.join(GEN, f"fn_{n}.h"), "w").write(f"// {rel}: {n}\n" + to_cpp(cut_function(src, n)))
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused", "-I", GEN, "-I", HERE,
           os.path.join(HERE, "ref_shim.cpp"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    a = ap.parse_args()
    print(build(a.reference, verbose=True))
