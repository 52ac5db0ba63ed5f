"""Known-answer tests that pin the CPU oracle to the reference's in-tree shader code.

The reference has no numeric test of RNG / sampling / codecs; these vectors were derived by
restating /root/reference/src/gi/shaders/common.glsl and Gi.cpp (SURVEY.md section 8c "Golden vectors / KATs"):
independent numpy restatements below double-check the C oracle.
"""
import ctypes as C
import math

import numpy as np
import pytest

U32 = 0xFFFFFFFF


def _np_hash_init(x):  # common.glsl:74-82
    x &= U32
    x ^= x >> 16; x = (x * 0x21F0AAAD) & U32
    x ^= x >> 15; x = (x * 0xD35A2D97) & U32
    x ^= x >> 15
    return x


def _np_next(state):  # common.glsl:85-96
    s = (state * 747796405 + 2891336453) & U32
    word = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & U32
    out = ((word >> 22) ^ word) & U32
    f = np.frombuffer(np.uint32(0x3F800000 | (out >> 9)).tobytes(), np.float32)[0] - np.float32(1.0)
    return out, float(f)


RNG_KATS = [  # (pixelIndex, sampleIndex, init, 4 state words or None, 4 floats)
    (0, 0, 0x00000000, (0x07BB2FE2, 0x30BE035E, 0x7FDDB461, 0x8D324821), (0.0301998854, 0.190399289, 0.499476671, 0.551548481)),
    (1, 0, 0x06D3FA73, (0x7BA9633D, 0x92D4EA04, 0xF4D72A0E, 0x306A062B), (0.483053327, 0.573561311, 0.956408143, 0.189117789)),
    (131328, 0, 0xC61F3147, (0xF2CD8E41, 0xB63A86F1, 0x26050D49, 0x6DD21DBF), (0.948449016, 0.711830497, 0.148514509, 0.428987265)),
    (131328, 5, 0xE809591D, None, (0.493243456, 0.251541376, 0.794243932, 0.217875957)),
    (2073599, 1023, 0x627C2970, None, (0.455858827, 0.640787244, 0.873230934, 0.159768939)),
]


@pytest.mark.parametrize("pixel,sample,init,words,floats", RNG_KATS)
def test_rng_kat(orc, pixel, sample, init, words, floats):
    L = orc.lib()
    assert L.orc_rng_init(pixel, sample) == init == _np_hash_init((pixel * (sample + 1)) & U32)
    st = C.c_uint32(init)
    py_state = init
    for k in range(4):
        f = L.orc_rng_next1f(C.byref(st))
        py_state, pf = _np_next(py_state)
        assert st.value == py_state
        if words:
            assert st.value == words[k]
        assert f == pytest.approx(floats[k], abs=5e-9) and f == pf
        assert 0.0 <= f < 1.0


def test_rng_pixel0_same_sequence_every_sample(orc):
    """seed = pixelIndex*(sampleIndex+1): pixel 0 repeats one sequence for all samples."""
    L = orc.lib()
    assert {L.orc_rng_init(0, s) for s in range(64)} == {0}
    assert L.orc_rng_init(6, 1) == L.orc_rng_init(3, 3) == L.orc_rng_init(12, 0)


OCT_KATS = [((0, 0, 1), 0x80008000), ((0, 0, -1), 0xFFFFFFFF), ((1, 0, 0), 0x8000FFFF), ((0, -1, 0), 0x00008000),
            ((1, 1, 1), 0xAAAAAAAA)]


@pytest.mark.parametrize("v,code", OCT_KATS)
def test_octahedral_encode_kat(orc, v, code):
    L = orc.lib()
    assert L.orc_encode_direction((C.c_float * 3)(*v)) == code


def test_octahedral_decode_kat(orc):
    L = orc.lib()
    out = (C.c_float * 3)()
    L.orc_decode_direction(0x80008000, out)  # axis-aligned normals do not round-trip exactly
    assert out[0] == pytest.approx(1.5259e-5, rel=1e-3) and out[1] == pytest.approx(1.5259e-5, rel=1e-3) and out[2] == pytest.approx(1.0, abs=1e-7)
    L.orc_decode_direction(0xFFFFFFFF, out)
    assert tuple(out) == (0.0, 0.0, -1.0)
    L.orc_decode_direction(0x8000FFFF, out)
    assert out[0] == pytest.approx(1.0, abs=1e-7) and out[1] == pytest.approx(0.0, abs=1e-12) and out[2] == pytest.approx(-1.5259e-5, rel=1e-3)
    L.orc_decode_direction(0x00008000, out)
    assert out[1] == pytest.approx(-1.0, abs=1e-7) and out[2] == pytest.approx(-1.5259e-5, rel=1e-3)


def test_octahedral_roundtrip_random(orc):
    L = orc.lib()
    rng = np.random.default_rng(7)
    v = rng.normal(size=(2000, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    out = (C.c_float * 3)()
    for x in v:
        L.orc_decode_direction(L.orc_encode_direction((C.c_float * 3)(*x)), out)
        d = np.array(out[:], np.float32)
        assert abs(np.linalg.norm(d) - 1) < 1e-6
        assert np.dot(d, x) > 1 - 2e-9 * 65535  # 16-bit quantisation: angular error ~ 1/65535


def test_fis_gauss_kat(orc):
    L = orc.lib()
    out = (C.c_float * 2)()
    L.orc_fis_gauss(0.948449016, 0.711830497, out)
    assert out[0] == pytest.approx(-0.0289808, abs=2e-7) and out[1] == pytest.approx(-0.1185154, abs=2e-7)
    L.orc_fis_gauss(0.0, 0.25, out)  # u1 = max(1e-38, 0): finite, r = .375*sqrt(-2 ln 1e-38)
    assert np.isfinite(out[0]) and out[1] == pytest.approx(0.375 * math.sqrt(-2 * math.log(1e-38)), rel=1e-5)


def test_polynomial_transcendentals(orc):
    """Arithmetic contract: sincos2pi / logf are polynomial kernels; accuracy vs libm documents their error."""
    L = orc.lib()
    s, c = C.c_float(), C.c_float()
    xs = np.concatenate([np.linspace(0, 1, 4097, dtype=np.float32), np.float32([1e-7, 0.125, 0.375, 0.5, 0.625, 0.875, 0.99999994])])
    for x in xs:
        L.orc_sincos2pi(float(x), C.byref(s), C.byref(c))
        assert abs(s.value - math.sin(2 * math.pi * float(x))) < 4e-7
        assert abs(c.value - math.cos(2 * math.pi * float(x))) < 4e-7
    for x in np.concatenate([np.float32(10.0) ** np.linspace(-37.9, 0, 500), np.float32([1e-40, 1e-38, 0.5, 0.70710678, 1.0])]).astype(np.float32):
        got = L.orc_logf(float(x))
        assert got == pytest.approx(math.log(float(x)), rel=3e-7, abs=3e-7)


def test_offset_ray_origin(orc):
    """common.glsl:143-162: integer-ulp offset for |p| >= 1/32, float offset below."""
    L = orc.lib()
    out = (C.c_float * 3)()
    p = np.float32([1.0, -2.0, 0.01]); n = np.float32([0.0, 0.0, 1.0])
    L.orc_offset_ray_origin((C.c_float * 3)(*p), (C.c_float * 3)(*n), out)
    assert out[0] == 1.0 and out[1] == -2.0 and out[2] == np.float32(0.01) + np.float32(1.0 / 65536.0)
    p = np.float32([1.0, -2.0, 0.5]); n = np.float32([1.0, 1.0, -1.0]) / np.float32(math.sqrt(3))
    L.orc_offset_ray_origin((C.c_float * 3)(*p), (C.c_float * 3)(*n), out)
    io = int(float(n[0]) * 64.0)  # 36
    exp = [np.frombuffer(np.int32(np.float32(1.0).view(np.int32) + io).tobytes(), np.float32)[0],
           np.frombuffer(np.int32(np.float32(-2.0).view(np.int32) - io).tobytes(), np.float32)[0],
           np.frombuffer(np.int32(np.float32(0.5).view(np.int32) - io).tobytes(), np.float32)[0]]
    assert [out[0], out[1], out[2]] == [float(e) for e in exp]
    assert out[0] > 1.0 and out[1] > -2.0 and out[2] < 0.5  # moved along +n on every axis


def test_half_pack(orc):
    L = orc.lib()
    assert L.orc_pack_half2x16(1.0, 1.0) == 0x3C003C00  # light diffuse/specular default (Gi.cpp:2591)
    assert L.orc_pack_half2x16(0.1, 100.0) == (0x5640 << 16) | 0x2E66  # cornell clippingRange
    out = (C.c_float * 2)()
    L.orc_unpack_half2x16(0x56402E66, out)
    assert out[0] == pytest.approx(0.0999755859375) and out[1] == 100.0
    for v in np.float16(np.random.default_rng(3).normal(size=200)):
        L.orc_unpack_half2x16(L.orc_pack_half2x16(float(v), 0.0), out)
        assert out[0] == float(v)


def test_orthonormal_basis(orc):
    L = orc.lib()
    b1, b2 = (C.c_float * 3)(), (C.c_float * 3)()
    rng = np.random.default_rng(11)
    for n in rng.normal(size=(200, 3)):
        n = (n / np.linalg.norm(n)).astype(np.float32)
        L.orc_orthonormal_basis((C.c_float * 3)(*n), b1, b2)
        a, b = np.array(b1[:]), np.array(b2[:])
        assert abs(a @ b) < 1e-6 and abs(a @ n) < 1e-6 and abs(b @ n) < 1e-6 and abs(np.linalg.norm(a) - 1) < 1e-6
        assert np.cross(a, b) @ n > 0.999


def test_cornell_camera_facts():
    """vfov = 2*atan(2.025/(2*5.0)) = 0.39959649 rad; position (0,-7,0); forward (0,1,4.37e-8)."""
    from gatling_amd.scenes import cornell_box
    cam = cornell_box().camera
    assert cam.vfov == pytest.approx(0.39959649, abs=3e-8)
    assert tuple(cam.position) == (0.0, -7.0, 0.0)
    assert cam.forward[1] == 1.0 and cam.forward[2] == pytest.approx(4.371139e-8, rel=1e-6)
    assert cam.up[2] == 1.0 and cam.up[1] == pytest.approx(-4.371139e-8, rel=1e-6)
    d = 1.0 / (2.0 * math.tan(cam.vfov * 0.5))
    assert d == pytest.approx(2.4691358, rel=1e-6)


def test_atan2_acos_polynomials(orc):
    """Arithmetic contract of the dome lookup (rp_main.miss:48-49): polynomial atan2 / acos, accuracy vs libm."""
    L = orc.lib()
    rng = np.random.default_rng(3)
    for y, x in np.concatenate([rng.normal(0, 1, (2000, 2)), [[0, 1], [0, -1], [1, 0], [-1, 0], [1e-20, 1], [1, 1e-20], [-1e-20, -1], [0, 0]]]).astype(np.float32):
        assert L.orc_atan2f(float(y), float(x)) == pytest.approx(math.atan2(float(y), float(x)), abs=4e-7)
    for x in np.concatenate([np.linspace(-1, 1, 4001), [-1.0000001, 1.0000001, 0.5, -0.5, 0.50000006, 0.99999994]]).astype(np.float32):
        xc = min(1.0, max(-1.0, float(x)))  # inputs are clamped: normalize() may overshoot 1 by an ulp
        assert L.orc_acosf(float(x)) == pytest.approx(math.acos(xc), abs=5e-7)


def test_texture_runtime_kat(orc):
    """mdl_interface.glsl:8-38, 127-145 over the software sampler: texel centres are exact, wrap modes as the runtime defines them."""
    tex = np.zeros((4, 8, 4), np.float32)
    tex[..., 0] = np.arange(8)[None, :]          # r = column
    tex[..., 1] = np.arange(4)[:, None]          # g = row
    tex[..., 2] = 1.0; tex[..., 3] = 0.5
    CLAMP, REPEAT, MIRROR, CLIP = 0, 1, 2, 3
    for col in range(8):
        for row in range(4):
            t = orc.tex_lookup(tex, (col + 0.5) / 8, (row + 0.5) / 4, REPEAT, REPEAT)
            assert t.tolist() == [col, row, 1.0, 0.5]
    # halfway between two texel centres: the average; across the border REPEAT blends with the opposite edge
    assert orc.tex_lookup(tex, 1.0 / 8, 0.5 / 4, REPEAT, REPEAT)[0] == pytest.approx(0.5)
    assert orc.tex_lookup(tex, 0.0, 0.5 / 4, REPEAT, REPEAT)[0] == pytest.approx(3.5)   # (0 + 7) / 2
    assert orc.tex_lookup(tex, 1.25, 0.5 / 4, REPEAT, REPEAT)[0] == pytest.approx(orc.tex_lookup(tex, 0.25, 0.5 / 4, REPEAT, REPEAT)[0])
    # CLAMP: coordinate clamped to [0.5/res, 1 - 0.5/res] -> edge texel, no bleed from the opposite edge
    assert orc.tex_lookup(tex, -3.0, 0.5 / 4, CLAMP, REPEAT)[0] == 0.0
    assert orc.tex_lookup(tex, 7.3, 0.5 / 4, CLAMP, REPEAT)[0] == 7.0
    # MIRRORED_REPEAT: odd periods run backwards
    assert orc.tex_lookup(tex, 1.0 + 0.5 / 8, 0.5 / 4, MIRROR, REPEAT)[0] == pytest.approx(7.0)
    assert orc.tex_lookup(tex, 2.0 + 0.5 / 8, 0.5 / 4, MIRROR, REPEAT)[0] == pytest.approx(0.0)
    assert orc.tex_lookup(tex, -0.5 / 8, 0.5 / 4, MIRROR, REPEAT)[0] == pytest.approx(0.0)
    # CLIP: black outside [0, 1]
    assert orc.tex_lookup(tex, 1.01, 0.5, CLIP, REPEAT).tolist() == [0, 0, 0, 0]
    assert orc.tex_lookup(tex, 0.5, -0.01, REPEAT, CLIP).tolist() == [0, 0, 0, 0]
    assert orc.tex_lookup(tex, 1.0, 0.5, CLIP, CLIP)[2] == 1.0
