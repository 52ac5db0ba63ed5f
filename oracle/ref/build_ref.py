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
             "rp_main.miss": ["quatRotateDir"], "rp_main.chit": ["sampleLight"], "mdl_interface.glsl": ["apply_wrap_and_crop", "mdl_adapt_normal", "tex_lookup_float4_2d", "tex_texel_float4_2d", "tex_resolution_2d", "tex_lookup_float4_3d",
                                    "tex_texel_float4_3d"]}


def to_cpp(text: str) -> str:
    text = re.sub(r"^\s*#\s*extension[^\n]*\n", "\n", text, flags=re.M)
    text = re.sub(r"\b(?:inout|out)\s+(\w+)\s+(\w+)\s*([,)])", r"\1& \2\3", text)  # out / inout parameters -> references
    text = re.sub(r"\bin\s+(\w+)\s+(\w+)\s*([,)])", r"\1 \2\3", text)              # in parameters -> by value
    text = re.sub(r"(?<=[\w\)\]])\.(xy|yx|zw|xyz|rgb|r|g|b)\b(?!\s*\()", r".\1()", text)      # swizzle reads -> member functions (glsl_compat.h)
    # GLSL evaluates constructor arguments left to right; C++ does so only in a braced list (the draws of rng1d_next2f / next4f, common.glsl:106-119)
    text = re.sub(r"return (vec[24])\(((?:\s*rng1d_next1f\(rng_state\)\s*,?)+)\s*\);", r"return \1{\2};", text)
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


LOOP_WHOLE = ["common.glsl", "aovs.glsl", "colormap.glsl", "rp_main_payload.glsl", "interface/rp_main.h", "mdl_types.glsl", "mdl_shading_state.glsl", "mdl_renderer_state.glsl",
              "rp_main_descriptors.glsl", "rp_main.rgen", "rp_main.chit", "rp_main.miss", "rp_main_shadow.miss"]
# (name, -D flags): the feature macros GlslShaderGen.cpp derives from the render settings (src/gi/impl/GlslShaderGen.cpp:196-300)
LOOP_VARIANTS = [("default", ["JITTERED_SAMPLING", "FILTER_IMPORTANCE_SAMPLING", "PROGRESSIVE_ACCUMULATION", "DOME_LIGHT_CAMERA_VISIBLE", "MEDIUM_STACK_SIZE=0"]),
                 ("nee", ["JITTERED_SAMPLING", "FILTER_IMPORTANCE_SAMPLING", "PROGRESSIVE_ACCUMULATION", "DOME_LIGHT_CAMERA_VISIBLE", "MEDIUM_STACK_SIZE=0", "NEXT_EVENT_ESTIMATION"]),
                 ("nee_stack2", ["JITTERED_SAMPLING", "FILTER_IMPORTANCE_SAMPLING", "PROGRESSIVE_ACCUMULATION", "DOME_LIGHT_CAMERA_VISIBLE", "MEDIUM_STACK_SIZE=2", "NEXT_EVENT_ESTIMATION"]),
                 ("dof_clip_box", ["JITTERED_SAMPLING", "PROGRESSIVE_ACCUMULATION", "DOME_LIGHT_CAMERA_VISIBLE", "MEDIUM_STACK_SIZE=0", "DEPTH_OF_FIELD", "CLIPPING_PLANES"]),
                 ("nojitter", ["DOME_LIGHT_CAMERA_VISIBLE", "MEDIUM_STACK_SIZE=0"]),
                 # every AOV but ClockCycles (bit 6: clockARB)
                 # the MDL renderer runtime's scene data (primvar) readers, mdl_interface.glsl:258-474, with the two named ids of Frontend.cpp:251-252
                 ("scenedata", ["JITTERED_SAMPLING", "FILTER_IMPORTANCE_SAMPLING", "PROGRESSIVE_ACCUMULATION", "DOME_LIGHT_CAMERA_VISIBLE", "MEDIUM_STACK_SIZE=0", "SCENE_DATA_COUNT=6",
                                "RENDERER_STATE_TYPE=mdl_renderer_state", "CAMERA_POSITION_SCENE_DATA_INDEX=7", "FRAME_SCENE_DATA_INDEX=8"]),
                 # a dome light the camera does not see (no DOME_LIGHT_CAMERA_VISIBLE: primary rays take the fallback texel, rp_main.miss:76-82)
                 ("nee_domehidden", ["JITTERED_SAMPLING", "FILTER_IMPORTANCE_SAMPLING", "PROGRESSIVE_ACCUMULATION", "MEDIUM_STACK_SIZE=0", "NEXT_EVENT_ESTIMATION"]),
                 ("aovs", ["JITTERED_SAMPLING", "FILTER_IMPORTANCE_SAMPLING", "PROGRESSIVE_ACCUMULATION", "DOME_LIGHT_CAMERA_VISIBLE", "MEDIUM_STACK_SIZE=0", "AOV_MASK=0x1ffbf"])]


def descriptors_to_cpp(text: str) -> str:
    """rp_main_descriptors.glsl: `layout(...) [readonly] buffer Block { T name[]; };` -> `static T* name;`, `... uniform Block { T name; };` -> `static T name;`;
    opaque resources (sampler, acceleration structure, texture arrays) and buffer-reference blocks are dropped (ref_loop.cpp declares what it uses)."""
    out, lines, i = [], text.split("\n"), 0
    one = re.compile(r"^\s*layout\([^)]*\)\s*(?:readonly\s+|writeonly\s+)*(?:uniform|buffer)\s+\w+\s*\{\s*(\w+)\s+(\w+)(\[\])?;\s*\};")
    while i < len(lines):
        ln = lines[i]
        m = one.match(ln)
        if m:
            out.append(f"static {m.group(1)}{'*' if m.group(3) else ''} {m.group(2)};")
        elif ln.lstrip().startswith("layout("):
            depth = 0
            while True:  # skip the whole statement
                depth += lines[i].count("{") - lines[i].count("}")
                if depth <= 0 and lines[i].rstrip().endswith(";"):
                    break
                i += 1
        else:
            out.append(ln)
        i += 1
    return "\n".join(out)


def remove_function(text: str, name: str) -> str:
    return text.replace(cut_function(text, name).rstrip("\n"), f"/* {name}: provided by ref_loop.cpp */")


def loop_to_cpp(rel: str, text: str) -> str:
    if rel == "rp_main_descriptors.glsl":
        text = descriptors_to_cpp(text)
    stem = rel.replace("rp_main", "").strip("._") or "rgen"
    if rel == "rp_main.rgen":
        text = re.sub(r"layout\([^)]*\)\s*rayPayloadEXT\s+(\w+)\s+(\w+);", r"static \1 \2;", text)
    elif rel == "rp_main_shadow.miss":
        text = re.sub(r"layout\([^)]*\)\s*rayPayloadInEXT\s+\w+\s+\w+;", "static ShadowRayPayload& rayPayload = shadowRayPayload;", text)
    else:
        text = re.sub(r"layout\([^)]*\)\s*rayPayloadInEXT\s+\w+\s+\w+;", "", text)
    text = re.sub(r"^\s*hitAttributeEXT[^\n]*\n", "\n", text, flags=re.M)
    text = re.sub(r"^\s*#\s*pragma\s+mdl_generated_code[^\n]*\n", "\n", text, flags=re.M)
    text = re.sub(r'^\s*#\s*include\s+"mdl_interface.glsl"[^\n]*\n', "\n", text, flags=re.M)
    names = {"rp_main.rgen": "rgen_main", "rp_main.chit": "chit_main", "rp_main.miss": "miss_main", "rp_main_shadow.miss": "shadow_miss_main"}
    if rel in names:
        text = re.sub(r"\bvoid\s+main\s*\(\s*\)", f"void {names[rel]}()", text)
    return "#pragma once\n" + to_cpp(text)


def build_loop(shaders: str, verbose=False):
    """The whole-shader build: rp_main.rgen / .chit / .miss / _shadow.miss as C++ functions, one object per feature-macro variant."""
    gen = os.path.join(OUT, "gen_loop")
    os.makedirs(os.path.join(gen, "interface"), exist_ok=True)
    open(os.path.join(gen, "interface", "gtl.h"), "w").write("/* stub: ref_loop.cpp defines the GLSL side of interface/gtl.h */\n")
    for rel in LOOP_WHOLE:
        open(os.path.join(gen, rel), "w").write(loop_to_cpp(rel, open(os.path.join(shaders, rel)).read()))
    # the scene-data readers of mdl_interface.glsl (from scene_data_isvalid up to, not including, the float4x4 stub); the rest of that file is
    # the texture runtime over hardware samplers (deviation D5)
    mi = open(os.path.join(shaders, "mdl_interface.glsl")).read()
    a, b = mi.index("bool scene_data_isvalid"), mi.index("mat4 scene_data_lookup_float4x4")
    open(os.path.join(gen, "fn_scene_data.h"), "w").write("// mdl_interface.glsl: scene data readers\n" + to_cpp(mi[a:b]))
    objs = []
    for name, defs in LOOP_VARIANTS:
        obj = os.path.join(OUT, f"ref_loop_{name}.o")
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-c", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused", "-Wno-attributes", "-Wno-unknown-pragmas", "-I", gen, "-I", HERE,
               f"-DREF_VARIANT={name}"] + [f"-D{d}" for d in defs] + [os.path.join(HERE, "ref_loop.cpp"), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    return objs


DEFAULT_MTLX = os.path.join(OUT, "hdgatling_default_material.mtlx")


def extract_default_material(reference="/root/reference") -> str:
    """hdGatling's fallback material -- the one reference-authored MaterialX document in the tree that is not a git-LFS stub -- cut verbatim out of
    src/hdGatling/renderDelegate.cpp (the raw string `_defaultMaterialXMaterial`) into the git-ignored oracle/_ref/, so that the GPU box's tests can feed it
    through gtl::giCreateMaterialFromMtlxStr (tests/test_mtlx_parity.py).  Generated at build time, never committed."""
    src = open(os.path.join(reference, "src", "hdGatling", "renderDelegate.cpp")).read()
    m = re.search(r'_defaultMaterialXMaterial\s*=\s*R"\((.*?)\)";', src, flags=re.S)
    if not m:
        raise SystemExit("renderDelegate.cpp: _defaultMaterialXMaterial not found")
    os.makedirs(OUT, exist_ok=True)
    open(DEFAULT_MTLX, "w").write(m.group(1))
    return DEFAULT_MTLX


def build(reference="/root/reference", verbose=False) -> str:
    shaders = os.path.join(reference, "src", "gi", "shaders")
    if not os.path.isdir(shaders):
        raise FileNotFoundError(shaders)
    extract_default_material(reference)
    os.makedirs(os.path.join(GEN, "interface"), exist_ok=True)
    open(os.path.join(GEN, "interface", "gtl.h"), "w").write("/* stub: ref_shim.cpp defines the GLSL side of interface/gtl.h (its C++ side needs glm) */\n")
    for rel in WHOLE:
        open(os.path.join(GEN, rel), "w").write(to_cpp(open(os.path.join(shaders, rel)).read()))
    for rel, names in FUNCTIONS.items():
        src = open(os.path.join(shaders, rel)).read()
        for n in names:
            open(os.path.join(GEN, f"fn_{n}.h"), "w").write(f"// {rel}: {n}\n" + to_cpp(cut_function(src, n)))
    objs = build_loop(shaders, verbose)
    oracle_dir = os.path.dirname(HERE)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused", "-I", GEN, "-I", HERE,
           os.path.join(HERE, "ref_shim.cpp")] + objs + ["-o", LIB, "-L", oracle_dir, "-lgi_oracle", "-Wl,-rpath," + oracle_dir]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    a = ap.parse_args()
    print(build(a.reference, verbose=True))
