"""The C++ face of the boundary: include/gtl/gi/Gi.h + gatling_amd/csrc/gtl_shim.cpp (same API shape as the reference's
Gi.h:199-261).  CPU: a C++ client compiles and links against the library with plain g++.  GPU: it runs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "gtl_smoke.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "gtl_smoke")


def _build():
    lib_dir = os.path.join(ROOT, "gatling_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE, "-L", lib_dir, "-lgatling_gi",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return EXE


def test_cpp_client_compiles_and_links():
    from gatling_amd import capi
    capi.load_library()  # the library must exist (no compute)
    exe = _build()
    assert os.path.exists(exe)
    syms = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "gatling_amd", "libgatling_gi.so")], text=True)
    for name in ("giInitialize", "giCreateMesh", "giRender", "giCreateMaterialFromMtlxStr", "giGetRenderBufferMem", "giSetRectLightTangents"):
        assert any(name in line and "gtl" in line for line in syms.splitlines()), name


@pytest.mark.gpu
def test_cpp_client_renders(gi):
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gtl_smoke ok" in out.stdout
