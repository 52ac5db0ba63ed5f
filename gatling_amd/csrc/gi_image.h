// gi_image.h -- the few image decoders the library carries itself (the reference reaches PNG / JPEG / EXR / HDR / TIFF through
// imgio, src/imgio/impl/*: out of scope).  Everything decodes to float RGBA, row 0 = first image row.
#pragma once

#include <cstdint>
#include <vector>

namespace gi {

// .hdr (Radiance RGBE), .pfm, .png (8/16-bit gray, gray+alpha, RGB, RGBA, palette; non-interlaced), .jpg (baseline / extended sequential, Huffman).  srgbToLinear applies the sRGB EOTF
// to the colour channels of 8-bit PNGs (UsdUVTexture sourceColorSpace = sRGB).
bool loadImageFile(const char* path, bool srgbToLinear, uint32_t& width, uint32_t& height, std::vector<float>& rgba);

} // namespace gi
