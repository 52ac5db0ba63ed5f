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
    # the ABI version the library was built with is the header's (a caller checks it before trusting struct layouts, include/gi_c.h)
    assert L.giCGetApiVersion() == int(re.search(r"#define\s+GI_C_API_VERSION\s+(\d+)u", text).group(1))


def test_struct_layouts_match_header():
    assert C.sizeof(capi.GiCCameraDesc) == 64 and C.sizeof(capi.GiCMaterialDesc) == 264
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


def test_bvh8_builder_collapse_rules_and_thread_counts(monkeypatch):
    """Both collapse rules stay conservative; the cost-optimal one fills the 8-wide nodes (fewer nodes for the same triangles) and is
    independent of the number of builder threads (large enough an input for the parallel paths: > 2^16 triangles, levels > 2048 nodes)."""
    rng = np.random.default_rng(11)
    n = 150_000
    v = (rng.uniform(-1, 1, (n, 1, 3)) + rng.normal(0, 0.01, (n, 3, 3))).astype(np.float32)
    L = capi.load_library()
    got = {}
    for collapse, threads in ((0, 8), (1, 1), (1, 8)):
        monkeypatch.setenv("GATLING_OPTIONS", f"bvh_collapse={collapse}"); monkeypatch.setenv("GATLING_BUILD_THREADS", str(threads))
        nodes, depth = C.c_uint32(), C.c_uint32()
        assert L.giCDebugValidateBvh(v.ctypes.data_as(capi._FP), n, C.byref(nodes), C.byref(depth)) == 0
        got[(collapse, threads)] = (nodes.value, depth.value)
    assert got[(1, 1)] == got[(1, 8)]
    assert got[(1, 8)][0] < 0.95 * got[(0, 8)][0] and got[(1, 8)][1] <= 16


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
        return out[::-1]  # the library returns imgio's orientation (row 0 = bottom scanline); the checks below are in file order
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
    _write_jpeg(tmp_path / "a.jpg", rng.integers(0, 255, (19, 23, 3)).astype(np.uint8), hv=(2, 2), dri=1)
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
    for name in ("a.png", "a.jpg", "a.hdr", "a.pfm"):
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


# ---- a small baseline JPEG encoder (test side only): custom single-length Huffman tables, optional chroma subsampling and restarts ----
def _zigzag():
    idx = sorted(((y + x, (y if (y + x) % 2 else x), y, x) for y in range(8) for x in range(8)))
    return [y * 8 + x for _, _, y, x in idx]


def _write_jpeg(path, img, hv=(1, 1), qscale=3, dri=0):
    """Writes `img` (uint8 [h, w] or [h, w, 3]) as baseline JPEG and returns the image a decoder defined as 'float IDCT, +128, round,
    chroma replicated, JFIF colour matrix, round' must produce (uint8, same shape)."""
    from scipy.fft import dctn, idctn
    img = np.asarray(img, np.uint8)
    h, w = img.shape[:2]
    colour = img.ndim == 3
    if colour:
        r, g, b = [img[..., k].astype(np.float64) for k in range(3)]
        planes = [0.299 * r + 0.587 * g + 0.114 * b, -0.168736 * r - 0.331264 * g + 0.5 * b + 128.0, 0.5 * r - 0.418688 * g - 0.081312 * b + 128.0]
        samp = [hv, (1, 1), (1, 1)]
    else:
        planes, samp = [img.astype(np.float64)], [(1, 1)]
    hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
    mw, mh = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    zz = _zigzag()
    q = np.clip((1 + np.add.outer(np.arange(8), np.arange(8))) * qscale, 2, 255).astype(np.int64)  # one table for all components
    comp_q, comp_rec = [], []
    for p, (sh, sv) in zip(planes, samp):
        full = np.pad(p, ((0, mh * 8 * vmax - h), (0, mw * 8 * hmax - w)), mode="edge")
        fy, fx = vmax // sv, hmax // sh
        sub = full.reshape(full.shape[0] // fy, fy, full.shape[1] // fx, fx).mean(axis=(1, 3))
        sub = np.clip(np.floor(sub + 0.5), 0, 255)
        by, bx = sub.shape[0] // 8, sub.shape[1] // 8
        blocks = sub.reshape(by, 8, bx, 8).transpose(0, 2, 1, 3) - 128.0
        coef = np.clip(np.rint(dctn(blocks, axes=(2, 3), norm="ortho") / q), -1023, 1023).astype(np.int64)
        comp_q.append(coef)
        rec = np.clip(np.floor(idctn((coef * q).astype(np.float64), axes=(2, 3), norm="ortho") + 128.5), 0, 255)
        comp_rec.append(np.repeat(np.repeat(rec.transpose(0, 2, 1, 3).reshape(by * 8, bx * 8), fy, axis=0), fx, axis=1)[:h, :w])
    if colour:
        y, cb, cr = comp_rec
        rgb = np.stack([y + 1.402 * (cr - 128), y - 0.344136 * (cb - 128) - 0.714136 * (cr - 128), y + 1.772 * (cb - 128)], axis=-1)
        expected = np.clip(np.floor(rgb + 0.5), 0, 255).astype(np.uint8)
    else:
        expected = comp_rec[0].astype(np.uint8)
    # Huffman tables: DC = 12 symbols of 4 bits, AC = 162 symbols of 8 bits (canonical codes = index)
    dc_vals = list(range(12))
    ac_vals = [0x00, 0xF0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]
    ac_code = {v: i for i, v in enumerate(ac_vals)}
    bits = []

    def put(code, n):
        bits.extend((code >> (n - 1 - k)) & 1 for k in range(n))

    def put_value(v):
        s = int(abs(v)).bit_length()
        return s, (v if v >= 0 else v + (1 << s) - 1)
    out = bytearray(b"\xff\xd8")

    def seg(marker, payload):
        out.extend(bytes([0xff, marker]) + (len(payload) + 2).to_bytes(2, "big") + payload)
    seg(0xdb, bytes([0]) + bytes(int(q.reshape(-1)[zz[k]]) for k in range(64)))
    seg(0xc0, bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([len(planes)]) +
        b"".join(bytes([c + 1, (samp[c][0] << 4) | samp[c][1], 0]) for c in range(len(planes))))
    seg(0xc4, bytes([0x00] + [12 if k == 4 else 0 for k in range(1, 17)] + dc_vals))
    seg(0xc4, bytes([0x10] + [162 if k == 8 else 0 for k in range(1, 17)] + ac_vals))
    if dri:
        seg(0xdd, dri.to_bytes(2, "big"))
    seg(0xda, bytes([len(planes)]) + b"".join(bytes([c + 1, 0x00]) for c in range(len(planes))) + bytes([0, 63, 0]))

    def flush():
        while len(bits) % 8:
            bits.append(1)
        for k in range(0, len(bits), 8):
            byte = int("".join(map(str, bits[k:k + 8])), 2)
            out.append(byte)
            if byte == 0xff:
                out.append(0)
        bits.clear()
    pred = [0] * len(planes)
    count = 0
    for my in range(mh):
        for mx in range(mw):
            if dri and count and count % dri == 0:
                flush()
                out.extend(bytes([0xff, 0xd0 + ((count // dri - 1) % 8)]))
                pred = [0] * len(planes)
            for c, (sh, sv) in enumerate(samp):
                for by in range(sv):
                    for bx in range(sh):
                        blk = comp_q[c][my * sv + by, mx * sh + bx].reshape(-1)
                        diff = int(blk[0]) - pred[c]; pred[c] = int(blk[0])
                        s, val = put_value(diff)
                        put(s, 4)
                        if s:
                            put(val, s)
                        run = 0
                        last = max([k for k in range(1, 64) if blk[zz[k]] != 0], default=0)
                        for k in range(1, last + 1):
                            v = int(blk[zz[k]])
                            if v == 0:
                                run += 1
                                continue
                            while run > 15:
                                put(ac_code[0xF0], 8); run -= 16
                            s, val = put_value(v)
                            put(ac_code[(run << 4) | s], 8); put(val, s)
                            run = 0
                        if last < 63:
                            put(ac_code[0x00], 8)
            count += 1
    flush()
    out.extend(b"\xff\xd9")
    open(path, "wb").write(bytes(out))
    return expected


@pytest.mark.parametrize("case", ["grey", "colour444", "colour420+restarts", "colour422"])
def test_jpeg_decoder(tmp_path, case):
    """Baseline JPEG (the format most UsdPreviewSurface assets ship their textures in; the reference decodes it through imgio):
    Huffman decoding, dequantisation, IDCT, chroma upsampling, restart intervals, sizes that are not MCU multiples, 0xFF stuffing."""
    L = capi.load_library()
    rng = np.random.default_rng(3)
    h, w = (21, 13) if case != "grey" else (17, 24)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 3.0), 128 + 90 * np.cos(yy / 4.0), 40 + 8 * ((xx + yy) % 9)], axis=-1) + rng.normal(0, 6, (h, w, 3))
    img = np.clip(base, 0, 255).astype(np.uint8)
    if case == "grey":
        expected = _write_jpeg(tmp_path / "a.jpg", img[..., 0])
        expected = np.repeat(expected[..., None], 3, axis=2)
        img = np.repeat(img[..., :1], 3, axis=2)
    else:
        hv = {"colour444": (1, 1), "colour420+restarts": (2, 2), "colour422": (2, 1)}[case]
        expected = _write_jpeg(tmp_path / "a.jpg", img, hv=hv, dri=2 if "restarts" in case else 0)
    wv, hv_ = C.c_uint32(), C.c_uint32()
    buf = (C.c_float * (w * h * 4))()
    assert L.giCDebugDecodeImage(str(tmp_path / "a.jpg").encode(), 0, C.byref(wv), C.byref(hv_), buf, len(buf)) == 1
    assert (wv.value, hv_.value) == (w, h)
    got = np.rint(np.ctypeslib.as_array(buf).reshape(h, w, 4)[::-1, :, :3] * 255.0).astype(np.int64)  # [::-1]: imgio orientation -> file order
    diff = np.abs(got - expected.astype(np.int64))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02, (diff.max(), (diff > 0).mean())   # float32 vs float64 IDCT: rare rounding ties
    assert np.abs(got - img.astype(np.int64)).mean() < 12                                  # and it resembles the source image
    # progressive files are refused, not mis-decoded
    blob = bytearray(open(tmp_path / "a.jpg", "rb").read())
    i = blob.index(b"\xff\xc0"); blob[i + 1] = 0xc2
    open(tmp_path / "p.jpg", "wb").write(bytes(blob))
    assert L.giCDebugDecodeImage(str(tmp_path / "p.jpg").encode(), 0, C.byref(wv), C.byref(hv_), None, 0) == 0


# REF_4C / REF_4C_JPG of the reference's imgio tests (/root/reference/src/imgio/impl/main.cpp:53-61): the 2x2 images
# tests/golden/imgio_4c/4c.{png,hdr,jpg} (copies of src/imgio/testenv/) must load as red, blue / white, green -- i.e. row 0 of a
# loaded image is the file's BOTTOM scanline (every imgio decoder ends with _FlipImage).  These are the only golden vectors
# /root/reference holds near this path; they pin the orientation of every file texture and dome-light image.
REF_4C = [255, 0, 0, 255, 0, 0, 255, 255, 255, 255, 255, 255, 0, 255, 0, 255]
REF_4C_JPG = [254, 0, 0, 255, 0, 0, 254, 255, 255, 255, 255, 255, 1, 255, 1, 255]


@pytest.mark.parametrize("name,ref", [("4c.png", REF_4C), ("4c.hdr", REF_4C), ("4c.jpg", REF_4C_JPG)])
def test_imgio_load_oriented_golden(name, ref):
    L = capi.load_library()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "imgio_4c", name)
    w, h = C.c_uint32(), C.c_uint32()
    buf = (C.c_float * 16)()
    assert L.giCDebugDecodeImage(path.encode(), 0, C.byref(w), C.byref(h), buf, 16) == 1
    assert (w.value, h.value) == (2, 2)
    got = np.rint(np.clip(np.ctypeslib.as_array(buf), 0.0, 1.0) * 255.0).astype(np.int64)
    if name.endswith(".jpg"):  # IDCT rounding differs between decoders by at most one code value
        assert np.abs(got - np.int64(ref)).max() <= 1, got
    else:
        assert got.tolist() == ref, got


# ---------------------------------------------------------------------------------------------------------------
# Asset reader and image loader hooks (giCRegisterAssetReader / giCSetImageLoader): the reference reads EVERY image through the
# registered GiAssetReader and decodes with imgio (/root/reference/src/gi/impl/TextureManager.cpp:39-52, rendererPlugin.cpp:95-143, 189)
# ---------------------------------------------------------------------------------------------------------------
class _MemoryAssets:
    """A GiCAssetReader over a dict {path: bytes}: serves paths no fopen could (package-relative names, URIs)."""

    def __init__(self, files):
        self.files, self.open_assets, self.log = dict(files), {}, []
        self._next = 1

        def _open(user, path):
            p = path.decode()
            self.log.append(("open", p))
            if p not in self.files:
                return None
            h = self._next; self._next += 1
            self.open_assets[h] = C.create_string_buffer(self.files[p], len(self.files[p]))
            return h

        def _size(user, a):
            return len(self.open_assets[a])

        def _data(user, a):
            return C.addressof(self.open_assets[a])

        def _close(user, a):
            self.log.append(("close", a)); del self.open_assets[a]

        self.cbs = (capi.ASSET_OPEN(_open), capi.ASSET_SIZE(_size), capi.ASSET_DATA(_data), capi.ASSET_CLOSE(_close))
        self.struct = capi.GiCAssetReader(None, *self.cbs)


def _decode(L, path, srgb=0, n=64):
    w, h = C.c_uint32(), C.c_uint32()
    buf = (C.c_float * n)()
    ok = L.giCDebugDecodeImage(path.encode(), srgb, C.byref(w), C.byref(h), buf, n)
    return ok, w.value, h.value, np.ctypeslib.as_array(buf).copy()


def test_asset_reader_serves_images_under_non_file_paths():
    """With a reader registered every image path goes open -> size -> data -> close through it -- a name that is no file loads, and bit for
    bit like the same bytes read from disk; a path the reader does not know fails (no fall-through to the file system)."""
    L = capi.load_library()
    g = os.path.join(ROOT, "tests", "golden", "imgio_4c")
    disk = {n: _decode(L, os.path.join(g, n), srgb=1) for n in ("4c.png", "4c.hdr", "4c.jpg")}
    assert all(d[0] == 1 for d in disk.values())
    mem = _MemoryAssets({"usdz://scene.usdz[textures/" + n + "]": open(os.path.join(g, n), "rb").read() for n in disk})
    L.giCRegisterAssetReader(C.byref(mem.struct))
    try:
        for n, ref in disk.items():
            ok, w, h, px = _decode(L, "usdz://scene.usdz[textures/" + n + "]", srgb=1)
            assert ok == 1 and (w, h) == ref[1:3] and np.array_equal(px.view(np.uint32), ref[3].view(np.uint32)), n
        assert _decode(L, os.path.join(g, "4c.png"))[0] == 0  # the reader does not know the real file's path: not served
        assert not mem.open_assets and sum(1 for e in mem.log if e[0] == "close") == 3  # every opened asset was closed
    finally:
        L.giCRegisterAssetReader(None)
    assert _decode(L, os.path.join(g, "4c.png"))[0] == 1  # unregistered: direct file reads again


def test_image_loader_hook_is_asked_first_and_can_decline():
    """A registered loader decodes what the library cannot (here: a made-up "EXR") in imgio's formats, is released once per load, and a
    0 return falls through to the in-library decoders."""
    L = capi.load_library()
    half = np.array([[0.5, 2.0, -1.0, 1.0], [65504.0, 6e-8, 0.0, 0.25]], np.float16)  # incl. the largest half and a subnormal
    rgba8 = np.array([[10, 128, 255, 64]], np.uint8)
    rgb16 = np.array([[0.5, 2.0, -1.0], [3e-5, 1024.0, 0.125]], np.float16)  # three channels: alpha comes back 1
    r32 = np.array([[0.75], [1.0e-3]], np.float32)                            # one channel: replicated, alpha 1
    keep, calls = [], []

    def _load(user, path, data, size, keep_hdr, out):
        blob = C.string_at(data, size)
        calls.append((path.decode(), blob[:4], keep_hdr))
        if blob[:4] == b"EXR!":
            arr, fmt = {b"h": (half, capi.IMAGE_RGBA16_FLOAT), b"b": (rgba8, capi.IMAGE_RGBA8_UNORM), b"3": (rgb16, capi.IMAGE_RGB16_FLOAT), b"r": (r32, capi.IMAGE_R32_FLOAT)}[blob[4:5]]
            keep.append(arr)
            out[0].format = fmt
            out[0].width, out[0].height = arr.shape[0], 1
            out[0].pixels = arr.ctypes.data
            out[0].handle = len(keep)
            return 1
        return 0

    released = []
    loader = capi.GiCImageLoader(None, capi.IMAGE_LOAD(_load), capi.IMAGE_RELEASE(lambda user, img: released.append(img[0].handle)))
    mem = _MemoryAssets({"a.exr": b"EXR!h", "b.exr": b"EXR!b", "d.exr": b"EXR!3", "e.exr": b"EXR!r", "c.png": open(os.path.join(ROOT, "tests", "golden", "imgio_4c", "4c.png"), "rb").read()})
    L.giCRegisterAssetReader(C.byref(mem.struct)); L.giCSetImageLoader(C.byref(loader))
    try:
        ok, w, h, px = _decode(L, "a.exr")
        assert ok == 1 and (w, h) == (2, 1) and np.array_equal(px[:8], half.astype(np.float32).ravel())
        ok, w, h, px = _decode(L, "b.exr", srgb=1)
        c = rgba8[0, :3] / np.float32(255.0)
        lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
        assert ok == 1 and (w, h) == (1, 1) and np.allclose(px[:3], lin, rtol=1e-6) and px[3] == np.float32(64) / np.float32(255)
        ok, w, h, px = _decode(L, "d.exr")
        want = np.concatenate([rgb16.astype(np.float32), np.ones((2, 1), np.float32)], axis=1).ravel()
        assert ok == 1 and (w, h) == (2, 1) and np.array_equal(px[:8], want)
        ok, w, h, px = _decode(L, "e.exr")
        want = np.concatenate([np.repeat(r32, 3, axis=1), np.ones((2, 1), np.float32)], axis=1).ravel()
        assert ok == 1 and (w, h) == (2, 1) and np.array_equal(px[:8], want)
        ok, w, h, px = _decode(L, "c.png")  # declined by the loader: the in-library PNG decoder takes it
        assert ok == 1 and (w, h) == (2, 2)
        assert released == [1, 2, 3, 4] and [c[0] for c in calls] == ["a.exr", "b.exr", "d.exr", "e.exr", "c.png"]
    finally:
        L.giCSetImageLoader(None); L.giCRegisterAssetReader(None)
    assert _decode(L, os.path.join(ROOT, "tests", "golden", "imgio_4c", "4c.png"))[0] == 1


def test_image_loader_hook_with_bogus_dimensions_fails_cleanly():
    """A loader that answers with a size it has no pixels for (2^31-1 squared, or 2^16 x 2^16 = beyond the 2^28-texel cap) must cost a failed load, not the
    process: no allocation is attempted, `release` and the asset's `close` still run, and the next load works."""
    L = capi.load_library()
    tiny = np.zeros(4, np.float32)
    sizes = iter([(0x7fffffff, 0x7fffffff), (1 << 16, 1 << 16), (0, 5), (1, 1)])

    def _load(user, path, data, size, keep_hdr, out):
        out[0].format = capi.IMAGE_RGBA32_FLOAT
        out[0].width, out[0].height = next(sizes)
        out[0].pixels = tiny.ctypes.data
        out[0].handle = 7
        return 1

    released = []
    loader = capi.GiCImageLoader(None, capi.IMAGE_LOAD(_load), capi.IMAGE_RELEASE(lambda user, img: released.append(img[0].handle)))
    mem = _MemoryAssets({"x.exr": b"EXR!....."})
    L.giCRegisterAssetReader(C.byref(mem.struct)); L.giCSetImageLoader(C.byref(loader))
    try:
        for _ in range(3):
            assert _decode(L, "x.exr")[0] == 0
        ok, w, h, px = _decode(L, "x.exr")
        assert ok == 1 and (w, h) == (1, 1)
        assert released == [7, 7, 7, 7] and not mem.open_assets
    finally:
        L.giCSetImageLoader(None); L.giCRegisterAssetReader(None)


def test_shade_class_of_materials(monkeypatch):
    """Host classification behind the k_shade variants (gi_build.cpp shadeClassOf; the reference's per-material feature #defines, GlslShaderGen.cpp:204-274): an OpenPBR
    material is BASE (3) only when every optional lobe is absent and every parameter finite; anything else keeps its BSDF class."""
    from gatling_amd.scene import MAT_DIFFUSE, MaterialDesc, P_FUZZ_COLOR
    O = MaterialDesc.open_pbr
    assert capi.shade_class(MaterialDesc.usd_preview_surface(klass=MAT_DIFFUSE)) == 0
    assert capi.shade_class(MaterialDesc.usd_preview_surface(metallic=0.5)) == 1
    for base in (O(), O(base_metalness=1.0, specular_roughness=0.05), O(base_diffuse_roughness=0.8, base_weight=0.5), O(emission_luminance=3.0), O(specular_weight=0.0),
                 O(geometry_opacity=0.5), O(coat_color=(0.1, 0.2, 0.3), coat_roughness=0.9), O(fuzz_color=(0.3, 0.3, 0.3)), O(transmission_color=(0.2, 0.3, 0.4), transmission_depth=2.0),
                 O(coat_rotation=0.3), O(specular_rotation=0.3)):   # (a turned tangent without an anisotropic lobe is not read)
        assert capi.shade_class(base) == 3
    for full in (O(coat_weight=0.1), O(fuzz_weight=0.2), O(transmission_weight=1e-3), O(geometry_thin_walled=True), O(specular_roughness_anisotropy=0.3),
                 O(coat_roughness_anisotropy=0.3), O(coat_weight=0.5, coat_roughness_anisotropy=0.3, coat_rotation=0.2), O(specular_roughness_anisotropy=0.3, specular_rotation=0.2), O(thin_film_weight=0.5), O(subsurface_weight=0.5), O(geometry_thin_walled=True, subsurface_weight=0.5)):
        assert capi.shade_class(full) == 2
    for k, bad in ((P_FUZZ_COLOR, np.nan), (0, np.inf), (22, -1.0)):  # (coat ior -1: the coat's F0 is infinite -- 0 * inf is not 0)
        m = O(); m.params[k] = bad
        assert capi.shade_class(m) == 2
    monkeypatch.setenv("GATLING_OPTIONS", "shade_variants=0")
    assert capi.shade_class(O()) == 2
