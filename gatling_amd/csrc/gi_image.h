// gi_image.h -- the few image decoders the library carries itself (the reference reaches PNG / JPEG / EXR / HDR / TIFF through
// imgio, src/imgio/impl/*: out of scope).  Everything decodes to float RGBA in imgio's
// orientation: row 0 = the file's BOTTOM scanline (imgio flips after decoding; REF_4C in src/imgio/impl/main.cpp:53-61 pins it).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace gi {

// .hdr (Radiance RGBE), .pfm, .png (8/16-bit gray, gray+alpha, RGB, RGBA, palette; non-interlaced), .jpg (baseline / extended sequential, Huffman).
// srgbToLinear applies the sRGB EOTF to the colour channels of 8-bit PNGs (UsdUVTexture sourceColorSpace = sRGB).
// (from the file's bytes: they may come from an asset reader instead of the file system, giCRegisterAssetReader)
bool decodeImageBytes(const uint8_t* bytes, size_t size, bool srgbToLinear, uint32_t& width, uint32_t& height, std::vector<float>& rgba);
bool readFileBytes(const char* path, std::vector<uint8_t>& bytes);
float srgb8ToLinear(uint8_t v);

} // namespace gi
