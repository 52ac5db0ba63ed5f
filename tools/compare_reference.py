#!/usr/bin/env python3
"""is OUR image the reference Vulkan path's image, within what Monte-Carlo noise allows?

The reference cannot run in this repo's containers (Vulkan ray tracing + OpenUSD + the MDL SDK); this is the comparator that answers BASELINE.json's "output images
must match the reference Vulkan path on identical USD input and RNG seed to within a stated per-channel float tolerance" the day a box has both.  Stated tolerance:

  (1) per-channel RMSE(ours, reference) <= 2 x the Monte-Carlo standard error of OUR estimate, measured from two of our renders with DISJOINT sample offsets
      (per-pixel variance of one render ~ (A - B)^2 / 2; the image-level standard error is the RMS of that over the pixels).  Two unbiased renderers at the same
      spp differ by sqrt(2) standard errors; a systematic difference -- a BSDF weight off by a few per cent -- adds its own RMS on top and crosses 2;
  (2) relative mean-luminance error <= 0.5 % (luminance as /root/reference/src/gi/shaders/common.glsl:253-256);
  (3) for information, the reference's own test criterion: the number of differing sRGB8 bytes after hdGatling's float -> sRGB8 conversion
      (/root/reference/src/hdGatling/main.cpp:136-143 `_AccurateLinearToSrgb`, :463-487 truncating `uint8_t(r * 255.0)`, :353-368 the count).

An 8-bit reference (the .png `gatling` writes) is clipped to [0, 1] and quantised: both of ours go through the same encoding first, the comparison runs in that domain,
and the quantisation step's own RMS (1 / (255 sqrt 12)) joins the standard error.

  python tools/compare_reference.py ours_a.pfm ours_b.pfm reference.png|.pfm|.hdr     -> one JSON object; exit code 0 = (1) and (2) hold
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LUMA = np.array([0.2126, 0.7152, 0.0722], np.float64)
RMSE_FACTOR, LUMA_TOL = 2.0, 0.005


def accurate_linear_to_srgb(x):
    """hdGatling/main.cpp:136-143, in fp32 like the reference."""
    x = np.asarray(x, np.float32)
    lo = x * np.float32(12.92)
    hi = np.power(np.abs(x), np.float32(1.0 / 2.4)) * np.float32(1.055) - np.float32(0.055)
    return np.where(x <= np.float32(0.0031308), lo, hi).astype(np.float32)


def to_srgb8(rgb):
    """hdGatling/main.cpp:463-487: gamma-encode, then `uint8_t(r * 255.0)` -- a truncating conversion of a value that is not clamped first; out-of-range floats are
    undefined behaviour in C++, here they saturate."""
    v = accurate_linear_to_srgb(rgb).astype(np.float64) * 255.0
    return np.clip(np.trunc(v), 0, 255).astype(np.uint8)


def compare(ours_a, ours_b, ref, ref_is_srgb8):
    """ours_a / ours_b: float [h, w, >=3] linear, two independent renders (disjoint sample offsets) at the spp being compared; ref: float linear, or uint8 sRGB when
    ref_is_srgb8.  Returns the dict the tool prints."""
    a, b = np.asarray(ours_a, np.float32)[..., :3], np.asarray(ours_b, np.float32)[..., :3]
    if ref_is_srgb8:
        r8 = np.asarray(ref)[..., :3].astype(np.uint8)
        r = r8.astype(np.float64) / 255.0
        ea, eb = (np.clip(accurate_linear_to_srgb(x).astype(np.float64), 0.0, 1.0) for x in (a, b))
        quant = 1.0 / (255.0 * np.sqrt(12.0))
        # luminance needs linear light: decode the reference's bytes (inverse of the encoding above), clip ours as the file format clipped the reference
        rl = np.where(r <= 0.0031308 * 12.92, r / 12.92, np.power((r + 0.055) / 1.055, 2.4))
        al = np.clip(a.astype(np.float64), 0.0, 1.0)
    else:
        r = np.asarray(ref, np.float64)[..., :3]
        ea, eb, quant = a.astype(np.float64), b.astype(np.float64), 0.0
        rl, al = r, a.astype(np.float64)
    if ea.shape != r.shape:
        raise ValueError(f"image sizes differ: ours {ea.shape}, reference {r.shape}")
    se = np.sqrt(np.mean((ea - eb) ** 2, axis=(0, 1)) / 2.0 + quant ** 2)   # per channel
    rmse = np.sqrt(np.mean((ea - r) ** 2, axis=(0, 1)))
    la, lr = float(np.mean(al @ LUMA)), float(np.mean(rl @ LUMA))
    luma_err = abs(la - lr) / max(abs(lr), 1e-12)
    out = {"rmse": [float(x) for x in rmse], "standard_error": [float(x) for x in se], "rmse_over_standard_error": [float(x / max(y, 1e-30)) for x, y in zip(rmse, se)],
           "rmse_ok": bool(np.all(rmse <= RMSE_FACTOR * se)), "mean_luminance_ours": la, "mean_luminance_reference": lr, "mean_luminance_rel_error": luma_err,
           "luminance_ok": bool(luma_err <= LUMA_TOL), "domain": "sRGB-encoded [0,1]" if ref_is_srgb8 else "linear float",
           "tolerance": f"RMSE <= {RMSE_FACTOR} x MC standard error (two of our renders at disjoint sample offsets); mean luminance within {LUMA_TOL * 100:.1f} %"}
    ours8 = to_srgb8(a)
    ref8 = np.asarray(ref)[..., :3].astype(np.uint8) if ref_is_srgb8 else to_srgb8(np.asarray(ref, np.float32)[..., :3])
    out["srgb8_differing_bytes"] = int(np.count_nonzero(ours8 != ref8))   # hdGatling/main.cpp:353-368 (alpha is always 255 on both sides)
    out["srgb8_bytes"] = int(ours8.size)
    out["pass"] = out["rmse_ok"] and out["luminance_ok"]
    return out


def load_image(path):
    """-> (array, is_srgb8).  Floats (.pfm, .hdr) come back linear in the library's orientation; an 8-bit file comes back as its raw bytes."""
    sys.path.insert(0, ROOT)
    import ctypes as C
    from gatling_amd import capi
    L = capi.load_library()
    w, h = C.c_uint32(), C.c_uint32()
    if not L.giCDebugDecodeImage(path.encode(), 0, C.byref(w), C.byref(h), None, 0):
        raise SystemExit(f"cannot decode {path} (in-library decoders: .png, baseline .jpg, .hdr, .pfm; convert an .exr first)")
    buf = np.empty(w.value * h.value * 4, np.float32)
    L.giCDebugDecodeImage(path.encode(), 0, C.byref(w), C.byref(h), buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size)
    img = buf.reshape(h.value, w.value, 4)
    if path.lower().endswith((".png", ".jpg", ".jpeg")):
        return np.rint(img * 255.0).astype(np.uint8), True
    return img, False


def main():
    if len(sys.argv) != 4:
        raise SystemExit(__doc__)
    (a, _), (b, _), (r, r8) = (load_image(p) for p in sys.argv[1:4])
    out = compare(a, b, r, r8)
    print(json.dumps(out, indent=1))
    sys.exit(0 if out["pass"] else 1)


if __name__ == "__main__":
    main()
