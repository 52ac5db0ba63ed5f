// gi_image.cpp -- see gi_image.h
#include "gi_image.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>

namespace gi {
namespace {

// Image files are untrusted input: dimensions are bounded (and checked against the bytes actually present) BEFORE anything is
// allocated, so a corrupt header can neither exhaust memory nor index out of bounds.
constexpr uint64_t MAX_IMAGE_DIM = 65536, MAX_IMAGE_PIXELS = 1ull << 28;
inline bool saneDims(long long w, long long h) { return w > 0 && h > 0 && (uint64_t)w <= MAX_IMAGE_DIM && (uint64_t)h <= MAX_IMAGE_DIM && (uint64_t)w * (uint64_t)h <= MAX_IMAGE_PIXELS; }

// Minimal decoders for dome-light images: Radiance .hdr (RGBE, flat or new-style RLE scanlines, -Y +X orientation) and
// .pfm (PF, little or big endian, rows bottom-up).  Output: float RGBA, row 0 = first image row (top).
static bool decodeHdrOrPfm(const std::vector<uint8_t>& d, uint32_t& w, uint32_t& h, std::vector<float>& out)
{
  size_t pos = 0;
  auto line = [&]() { std::string l; while (pos < d.size() && d[pos] != '\n') l.push_back((char)d[pos++]); if (pos < d.size()) pos++; return l; };
  if (d.size() > 2 && d[0] == 'P' && d[1] == 'F') { // PFM
    line();
    int iw = 0, ih = 0; { std::string l = line(); if (sscanf(l.c_str(), "%d %d", &iw, &ih) != 2) { std::string l2 = line(); iw = atoi(l.c_str()); ih = atoi(l2.c_str()); } }
    const float scale = (float)atof(line().c_str());
    if (!saneDims(iw, ih) || pos + (size_t)iw * ih * 12 > d.size()) return false;
    w = (uint32_t)iw; h = (uint32_t)ih; out.assign((size_t)w * h * 4, 1.0f);
    for (uint32_t y = 0; y < h; y++)
      for (uint32_t x = 0; x < w; x++)
        for (int c = 0; c < 3; c++) {
          uint8_t b[4]; memcpy(b, &d[pos + (((size_t)y * w + x) * 3 + c) * 4], 4);
          if (scale > 0.0f) std::swap(b[0], b[3]), std::swap(b[1], b[2]); // positive scale = big endian
          float v; memcpy(&v, b, 4);
          out[((size_t)(h - 1 - y) * w + x) * 4 + c] = v;
        }
    return true;
  }
  std::string first = line();
  if (first.rfind("#?", 0) != 0) return false;
  for (;;) { std::string l = line(); if (l.empty()) break; if (pos >= d.size()) return false; }
  int ih = 0, iw = 0; { std::string l = line(); if (sscanf(l.c_str(), "-Y %d +X %d", &ih, &iw) != 2) return false; }
  if (!saneDims(iw, ih)) return false;
  { // the smallest encoding of a scanline: flat = 4 bytes per pixel; RLE = 4-byte header + per channel one (count, value) pair per 127 pixels
    const size_t rle = 4u + 8u * (((size_t)iw + 126u) / 127u), flat = (size_t)iw * 4u;
    if ((d.size() - std::min(pos, d.size())) / std::min(rle, flat) < (size_t)ih) return false;
  }
  w = (uint32_t)iw; h = (uint32_t)ih; out.assign((size_t)w * h * 4, 1.0f);
  std::vector<uint8_t> scan((size_t)w * 4);
  for (uint32_t y = 0; y < h; y++) {
    if (pos + 4 <= d.size() && d[pos] == 2 && d[pos + 1] == 2 && (((uint32_t)d[pos + 2] << 8) | d[pos + 3]) == w && w >= 8 && w < 32768) {
      pos += 4;
      for (int c = 0; c < 4; c++) {
        uint32_t x = 0;
        while (x < w) {
          if (pos >= d.size()) return false;
          uint8_t n = d[pos++];
          if (n > 128) { n -= 128; if (pos >= d.size() || x + n > w) return false; uint8_t v = d[pos++]; for (uint8_t k = 0; k < n; k++) scan[(size_t)(x++) * 4 + c] = v; }
          else { if (n == 0 || pos + n > d.size() || x + n > w) return false; for (uint8_t k = 0; k < n; k++) scan[(size_t)(x++) * 4 + c] = d[pos++]; }
        }
      }
    } else {
      if (pos + (size_t)w * 4 > d.size()) return false;
      memcpy(scan.data(), &d[pos], (size_t)w * 4); pos += (size_t)w * 4;
    }
    for (uint32_t x = 0; x < w; x++) {
      const uint8_t* p4 = &scan[(size_t)x * 4];
      const float sc = p4[3] ? ldexpf(1.0f, (int)p4[3] - 136) : 0.0f;
      float* o = &out[((size_t)y * w + x) * 4];
      o[0] = (float)p4[0] * sc; o[1] = (float)p4[1] * sc; o[2] = (float)p4[2] * sc;
    }
  }
  return true;
}


inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// PNG (ISO/IEC 15948): IHDR / PLTE / tRNS / IDAT chunks, zlib inflate, the five scanline filters; no Adam7 interlace.
static bool decodePng(const std::vector<uint8_t>& d, bool srgbToLinear, uint32_t& w, uint32_t& h, std::vector<float>& out)
{
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (d.size() < 8 || memcmp(d.data(), sig, 8) != 0) return false;
  size_t pos = 8;
  uint32_t depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  while (pos + 12 <= d.size()) {
    const uint32_t len = be32(&d[pos]);
    const uint8_t* tag = &d[pos + 4];
    if (pos + 12 + (size_t)len > d.size()) return false;
    const uint8_t* body = &d[pos + 8];
    if (!memcmp(tag, "IHDR", 4) && len >= 13) { w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
    else if (!memcmp(tag, "PLTE", 4)) plte.assign(body, body + len);
    else if (!memcmp(tag, "tRNS", 4)) trns.assign(body, body + len);
    else if (!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
    else if (!memcmp(tag, "IEND", 4)) break;
    pos += 12 + (size_t)len;
  }
  if (!saneDims(w, h) || interlace != 0 || (depth != 8 && depth != 16 && !(ctype == 3 && (depth == 1 || depth == 2 || depth == 4)))) return false;
  if (ctype == 3 && depth > 8) return false; // palette indices are at most 8 bits
  const uint32_t channels = ctype == 0 ? 1u : ctype == 2 ? 3u : ctype == 3 ? 1u : ctype == 4 ? 2u : ctype == 6 ? 4u : 0u;
  if (!channels || (ctype == 3 && plte.empty())) return false;
  const size_t bpp = std::max<size_t>(1, channels * depth / 8), stride = ((size_t)w * channels * depth + 7) / 8;
  if ((stride + 1) * (size_t)h > idat.size() * 1040u + 65536u) return false; // deflate cannot expand by more than ~1032:1
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf rawLen = (uLongf)raw.size();
  if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return false;
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  out.assign((size_t)w * h * 4, 1.0f);
  auto toLinear = [&](float c) { return srgbToLinear ? (c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f)) : c; };
  for (uint32_t y = 0; y < h; y++) {
    const uint8_t* line = &raw[(stride + 1) * y];
    const uint8_t filter = line[0];
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int pred = 0;
      if (filter == 1) pred = a; else if (filter == 2) pred = b; else if (filter == 3) pred = (a + b) / 2;
      else if (filter == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
      else if (filter != 0) return false;
      cur[i] = (uint8_t)(line[1 + i] + pred);
    }
    for (uint32_t x = 0; x < w; x++) {
      float* o = &out[((size_t)y * w + x) * 4];
      auto sample = [&](uint32_t ch) -> float {
        if (depth == 16) { const size_t k = ((size_t)x * channels + ch) * 2; return (float)(((uint32_t)cur[k] << 8) | cur[k + 1]) / 65535.0f; }
        return (float)cur[(size_t)x * channels + ch] / 255.0f;
      };
      if (ctype == 3) {
        const uint32_t bitPos = x * depth, idx = (cur[bitPos / 8] >> (8 - depth - (bitPos % 8))) & ((1u << depth) - 1u);
        for (int k = 0; k < 3; k++) o[k] = toLinear(idx * 3 + k < plte.size() ? (float)plte[idx * 3 + k] / 255.0f : 0.0f);
        o[3] = idx < trns.size() ? (float)trns[idx] / 255.0f : 1.0f;
      } else if (channels <= 2) {
        const float g = depth == 8 ? toLinear(sample(0)) : sample(0);
        o[0] = o[1] = o[2] = g; o[3] = channels == 2 ? sample(1) : 1.0f;
      } else {
        for (int k = 0; k < 3; k++) o[k] = depth == 8 ? toLinear(sample(k)) : sample(k);
        o[3] = channels == 4 ? sample(3) : 1.0f;
      }
    }
    prev.swap(cur);
  }
  return true;
}

} // namespace

bool loadImageFile(const char* path, bool srgbToLinear, uint32_t& w, uint32_t& h, std::vector<float>& out)
{
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  std::vector<uint8_t> d;
  { uint8_t buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + n); }
  fclose(f);
  try {
    if (d.size() >= 8 && d[0] == 0x89 && d[1] == 'P') return decodePng(d, srgbToLinear, w, h, out);
    return decodeHdrOrPfm(d, w, h, out);
  } catch (const std::exception&) { return false; } // allocation failure: the caller reports "cannot decode"
}

} // namespace gi
