"""CPU-side checks of the product library: it loads, exports every symbol include/gi_c.h declares, fails loudly
without a GPU, and its host-side BVH8 builder honours the conservativeness contract.  No device compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gatling_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    text = open(os.path.join(ROOT, "include", "gi_c.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(giC[A-Za-z0-9]+)\s*\(", text))
    bound = {name for name, _, _ in capi.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    L = capi.load_library()
    for name in declared:
        assert getattr(L, name) is not None


def test_struct_layouts_match_header():
    assert C.sizeof(capi.GiCCameraDesc) == 64 and C.sizeof(capi.GiCMaterialDesc) == 200
    assert C.sizeof(capi.GiCRenderSettings) == 72 and C.sizeof(capi.GiCAovBinding) == 32
    from gatling_amd.scene import VERTEX_DTYPE
    assert VERTEX_DTYPE.itemsize == 48  # GiVertex, Gi.h:110-118


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present")
def test_no_cpu_fallback():
    """On a machine without a GPU the product must refuse to run rather than fall back."""
    L = capi.load_library()
    assert L.giCInitialize(0) != capi.GI_C_OK
    assert b"no HIP device" in L.giCGetLastError()
    assert not L.giCCreateScene()


@pytest.mark.parametrize("n,seed", [(0, 0), (1, 1), (3, 2), (4, 3), (46, 4), (777, 5), (20000, 6)])
def test_bvh8_builder_is_conservative(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n, 1, 3))
    v = (c + rng.normal(0, 0.02, (n, 3, 3))).astype(np.float32)
    nodes, depth = C.c_uint32(), C.c_uint32()
    L = capi.load_library()
    bad = L.giCDebugValidateBvh(v.ctypes.data_as(capi._FP), n, C.byref(nodes), C.byref(depth))
    assert bad == 0
    assert nodes.value >= 1 and depth.value >= 1
    if n > 24:
        assert nodes.value >= n // 24


def test_bvh8_builder_degenerate_inputs():
    """Axis-aligned planes (flat boxes), coincident triangles and huge coordinates."""
    L = capi.load_library()
    quad = np.float32([[[-1, -1, 0], [1, -1, 0], [1, 1, 0]], [[-1, -1, 0], [1, 1, 0], [-1, 1, 0]]])
    same = np.repeat(quad[:1], 50, axis=0)
    big = (quad * 1e6 + 3e7).astype(np.float32)
    for v in (quad, same, big, np.concatenate([quad, same, big])):
        v = np.ascontiguousarray(v, np.float32)
        assert L.giCDebugValidateBvh(v.ctypes.data_as(capi._FP), len(v), None, None) == 0
