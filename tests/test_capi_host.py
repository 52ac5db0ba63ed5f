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


def _write_png(path, pixels, color_type, depth, filters, palette=None, trns=None):
    """Minimal PNG writer (zlib + the five scanline filters) to exercise the in-library decoder."""
    import struct
    import zlib
    h, w = pixels.shape[:2]
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    if depth == 16:
        rows = [pixels[y].astype(">u2").tobytes() for y in range(h)]
    elif depth == 8:
        rows = [pixels[y].astype(np.uint8).tobytes() for y in range(h)]
    else:  # packed palette indices
        rows = []
        for y in range(h):
            bits = "".join(format(int(v), "0%db" % depth) for v in pixels[y].reshape(-1))
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    bpp = max(1, channels * depth // 8)
    raw, prev = b"", bytes(len(rows[0]))
    for y, line in enumerate(rows):
        f = filters[y % len(filters)]
        out = bytearray(len(line))
        for i, v in enumerate(line):
            a = line[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 0: pred = 0
            elif f == 1: pred = a
            elif f == 2: pred = b
            elif f == 3: pred = (a + b) // 2
            else:
                p = a + b - c; pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (v - pred) & 0xFF
        raw += bytes([f]) + bytes(out)
        prev = line

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    comp = zlib.compress(raw, 6)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        data += chunk(b"tRNS", bytes(trns))
    half = len(comp) // 2
    data += chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")
    open(path, "wb").write(data)


def test_png_decoder(tmp_path):
    """gi_image.cpp: PNG colour types / bit depths / all five filters / split IDAT, and the sRGB EOTF on 8-bit colour."""
    L = capi.load_library()
    rng = np.random.default_rng(4)

    def decode(path, srgb):
        w, h = C.c_uint32(), C.c_uint32()
        assert L.giCDebugDecodeImage(str(path).encode(), int(srgb), C.byref(w), C.byref(h), None, 0) == 1
        out = np.zeros((h.value, w.value, 4), np.float32)
        assert L.giCDebugDecodeImage(str(path).encode(), int(srgb), C.byref(w), C.byref(h), out.ctypes.data_as(capi._FP), out.size) == 1
        return out
    rgb = rng.integers(0, 256, (7, 5, 3))
    _write_png(tmp_path / "rgb8.png", rgb, 2, 8, [0, 1, 2, 3, 4])
    got = decode(tmp_path / "rgb8.png", False)
    assert np.array_equal(got[..., :3], (rgb / 255.0).astype(np.float32)) and np.all(got[..., 3] == 1.0)
    lin = decode(tmp_path / "rgb8.png", True)
    c = rgb / 255.0
    assert np.allclose(lin[..., :3], np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4), atol=1e-6)
    rgba16 = rng.integers(0, 65536, (4, 6, 4))
    _write_png(tmp_path / "rgba16.png", rgba16, 6, 16, [4, 3, 1])
    assert np.array_equal(decode(tmp_path / "rgba16.png", True), (rgba16 / 65535.0).astype(np.float32))  # 16-bit data is linear
    ga = rng.integers(0, 256, (3, 9, 2))
    _write_png(tmp_path / "ga8.png", ga, 4, 8, [2])
    got = decode(tmp_path / "ga8.png", False)
    assert np.array_equal(got[..., 0], (ga[..., 0] / 255.0).astype(np.float32)) and np.array_equal(got[..., 3], (ga[..., 1] / 255.0).astype(np.float32))
    idx = rng.integers(0, 4, (5, 7))
    pal = [255, 0, 0, 0, 255, 0, 0, 0, 255, 10, 20, 30]
    _write_png(tmp_path / "pal2.png", idx, 3, 2, [0, 4], palette=pal, trns=[255, 128])
    got = decode(tmp_path / "pal2.png", False)
    exp = np.float32(pal).reshape(4, 3)[idx] / np.float32(255.0)
    assert np.array_equal(got[..., :3], exp)
    assert np.array_equal(got[..., 3], np.where(idx == 1, np.float32(128 / 255.0), np.float32(1.0)).astype(np.float32))
    open(tmp_path / "bad.png", "wb").write(b"\x89PNG\r\n\x1a\nxxxx")
    assert L.giCDebugDecodeImage(str(tmp_path / "bad.png").encode(), 0, None, None, None, 0) == 0


def test_image_decoders_survive_corrupt_files(tmp_path):
    """Image files are untrusted input: truncated, bit-flipped, spliced and size-bombed .png / .hdr / .pfm files are either decoded or
    refused -- never a crash, an uncaught exception or an allocation sized by a corrupt header (giCDebugDecodeImage needs no device)."""
    L = capi.load_library()
    rng = np.random.default_rng(0)
    _write_png(tmp_path / "a.png", rng.integers(0, 255, (9, 7, 4)).astype(np.uint8), 6, 8, [0, 1, 2, 3, 4])
    with open(tmp_path / "a.hdr", "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 5 +X 9\n" + rng.integers(1, 255, 5 * 9 * 4).astype(np.uint8).tobytes())
    with open(tmp_path / "a.pfm", "wb") as f:
        f.write(b"PF\n6 4\n-1.0\n" + rng.random(6 * 4 * 3).astype(np.float32).tobytes())

    def decode(path):
        w, h = C.c_uint32(), C.c_uint32()
        ok = L.giCDebugDecodeImage(str(path).encode(), 1, C.byref(w), C.byref(h), None, 0)
        if ok and w.value * h.value < 1 << 20:
            buf = (C.c_float * (w.value * h.value * 4))()
            ok = L.giCDebugDecodeImage(str(path).encode(), 1, C.byref(w), C.byref(h), buf, len(buf))
        return bool(ok)
    for name in ("a.png", "a.hdr", "a.pfm"):
        assert decode(tmp_path / name)
        blob = open(tmp_path / name, "rb").read()
        for it in range(400):
            b = bytearray(blob)
            if it % 4 == 0:
                b = b[:rng.integers(0, len(b))]
            elif it % 4 == 1:
                for _ in range(rng.integers(1, 6)):
                    b[rng.integers(0, len(b))] = rng.integers(0, 256)
            elif it % 4 == 2:
                i = rng.integers(0, len(b)); b[i:i] = bytes(rng.integers(0, 256, rng.integers(1, 40)).astype(np.uint8))
            else:
                i = rng.integers(0, max(1, len(b) - 4)); b[i:i + 4] = bytes([255, 255, 255, 127])
            p = tmp_path / ("m" + name[1:])
            open(p, "wb").write(bytes(b))
            decode(p)  # must return
    for bomb in (b"#?RADIANCE\n\n-Y 60000 +X 60000\n", b"PF\n2000000000 2000000000\n-1.0\n"):
        open(tmp_path / "bomb.hdr", "wb").write(bomb + b"\0" * 64)
        assert not decode(tmp_path / "bomb.hdr")
