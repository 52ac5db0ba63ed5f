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
inline bool saneDims(long long w, long long h)
{ return w > 0 && h > 0 && (uint64_t)w <= MAX_IMAGE_DIM && (uint64_t)h <= MAX_IMAGE_DIM && (uint64_t)w * (uint64_t)h <= MAX_IMAGE_PIXELS; }

// Minimal decoders for dome-light images: Radiance .hdr (RGBE, flat or new-style RLE scanlines, -Y +X orientation) and
// .pfm (PF, little or big endian, rows bottom-up).  Output: float RGBA, row 0 = first image row (top).
static bool decodeHdrOrPfm(const std::vector<uint8_t>& d, uint32_t& w, uint32_t& h, std::vector<float>& out)
{
  size_t pos = 0;
  auto line = [&]() { std::string l; while (pos < d.size() && d[pos] != '\n') l.push_back((char)d[pos++]); if (pos < d.size()) pos++; return l; };
  if (d.size() > 2 && d[0] == 'P' && d[1] == 'F') { // PFM
    line();
    int iw = 0, ih = 0;
        { std::string l = line(); if (sscanf(l.c_str(), "%d %d", &iw, &ih) != 2) { std::string l2 = line(); iw = atoi(l.c_str()); ih = atoi(l2.c_str()); } }
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
          if (n > 128) { n -= 128; if (pos >= d.size() || x + n > w) return false; uint8_t v = d[pos++];
              for (uint8_t k = 0; k < n; k++) scan[(size_t)(x++) * 4 + c] = v; }
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
      else if (filter == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
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


// JPEG (ITU-T T.81 / JFIF): baseline and extended sequential DCT (SOF0 / SOF1), Huffman coding, 8-bit samples, grey or YCbCr with
// sampling factors up to 4 x 4, restart intervals.  Progressive (SOF2), arithmetic coding and 12-bit files are refused.  The decode is
// defined here as: float IDCT -> +128 -> clamp -> round to 8 bits; chroma upsampled by replication; JFIF YCbCr -> RGB in float.
struct JpegHuff { uint8_t bits[17] = {0}; uint8_t vals[256] = {0}; int mincode[17], maxcode[18], valptr[17]; bool defined = false; };
struct JpegComp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; int bw = 0, bh = 0; std::vector<uint8_t> plane; };
struct JpegBits {
  const uint8_t* d; size_t n, pos; uint32_t acc = 0; int cnt = 0; bool hitMarker = false;
  int bit()
  {
    if (cnt == 0) {
      if (pos >= n) { hitMarker = true; return 0; }
      uint8_t b = d[pos];
      if (b == 0xff) {
        if (pos + 1 < n && d[pos + 1] == 0x00) pos += 2;       // stuffed byte
        else { hitMarker = true; return 0; }                    // a marker: stop feeding bits (the scan ends or a restart follows)
      } else pos++;
      acc = b; cnt = 8;
    }
    cnt--;
    return (acc >> cnt) & 1;
  }
  int receive(int s) { int v = 0; for (int i = 0; i < s; i++) v = (v << 1) | bit(); return v; }
};
static void jpegBuildHuff(JpegHuff& h)
{
  int code = 0, k = 0;
  for (int len = 1; len <= 16; len++) {
    h.valptr[len] = k; h.mincode[len] = code;
    code += h.bits[len]; k += h.bits[len];
    h.maxcode[len] = h.bits[len] ? code - 1 : -1;
    code <<= 1;
  }
  h.maxcode[17] = 0x7fffffff; h.defined = true;
}
static int jpegDecodeSymbol(JpegBits& br, const JpegHuff& h)
{
  int code = 0;
  for (int len = 1; len <= 16; len++) {
    code = (code << 1) | br.bit();
    if (h.bits[len] && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
  }
  return -1; // corrupt stream
}
static inline int jpegExtend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }

static bool decodeJpeg(const std::vector<uint8_t>& d, bool srgbToLinear, uint32_t& w, uint32_t& h, std::vector<float>& out)
{
  static const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62,
                                         63};
  if (d.size() < 4 || d[0] != 0xff || d[1] != 0xd8) return false;
  uint16_t qt[4][64]; bool qtDefined[4] = {false, false, false, false};
  JpegHuff hdc[4], hac[4];
  std::vector<JpegComp> comps;
  int restartInterval = 0, hmax = 1, vmax = 1;
  bool haveFrame = false;
  float cosTab[8][8]; // cosTab[x][u] = C(u)/2 * cos((2x+1) u pi / 16)
  for (int x = 0; x < 8; x++) for (int u = 0; u < 8; u++) cosTab[x][u] = (u == 0
      ? 0.35355339059327379f : 0.5f) * cosf((float)((2 * x + 1) * u) * 0.19634954084936207f);
  size_t pos = 2;
  while (pos + 4 <= d.size()) {
    if (d[pos] != 0xff) { pos++; continue; }
    const uint8_t m = d[pos + 1];
    if (m == 0xff) { pos++; continue; }
    if (m == 0xd8 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) { pos += 2; continue; }
    if (m == 0xd9) break;
    const size_t len = ((size_t)d[pos + 2] << 8) | d[pos + 3];
    if (len < 2 || pos + 2 + len > d.size()) return false;
    const uint8_t* b = &d[pos + 4]; const size_t n = len - 2;
    if (m == 0xdb) { // DQT
      size_t i = 0;
      while (i < n) {
        const int pq = b[i] >> 4, tq = b[i] & 15; i++;
        if (tq > 3 || pq > 1 || i + (size_t)64 * (pq + 1) > n) return false;
        for (int k = 0; k < 64; k++) { qt[tq][zigzag[k]] = pq ? (uint16_t)((b[i] << 8) | b[i + 1]) : b[i]; i += pq + 1; }
        qtDefined[tq] = true;
      }
    } else if (m == 0xc4) { // DHT
      size_t i = 0;
      while (i < n) {
        if (i + 17 > n) return false;
        const int tc = b[i] >> 4, th = b[i] & 15;
        if (tc > 1 || th > 3) return false;
        JpegHuff& ht = tc ? hac[th] : hdc[th];
        int total = 0;
        for (int k = 1; k <= 16; k++) { ht.bits[k] = b[i + k]; total += b[i + k]; }
        i += 17;
        if (total > 256 || i + (size_t)total > n) return false;
        memcpy(ht.vals, b + i, (size_t)total); i += (size_t)total;
        jpegBuildHuff(ht);
      }
    } else if (m == 0xc0 || m == 0xc1) { // SOF0 / SOF1
      if (n < 6 || b[0] != 8) return false;
      h = ((uint32_t)b[1] << 8) | b[2]; w = ((uint32_t)b[3] << 8) | b[4];
      const int nc = b[5];
      if (!saneDims(w, h) || (nc != 1 && nc != 3) || n < (size_t)6 + 3 * (size_t)nc) return false;
      comps.assign((size_t)nc, JpegComp());
      for (int c = 0; c < nc; c++) {
        comps[c].id = b[6 + 3 * c]; comps[c].h = b[7 + 3 * c] >> 4; comps[c].v = b[7 + 3 * c] & 15; comps[c].tq = b[8 + 3 * c];
        if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) return false;
        hmax = std::max(hmax, comps[c].h); vmax = std::max(vmax, comps[c].v);
      }
      haveFrame = true;
    } else if (m == 0xc2 || (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) {
      return false; // progressive, lossless, arithmetic: not supported
    } else if (m == 0xdd) { // DRI
      if (n < 2) return false;
      restartInterval = (b[0] << 8) | b[1];
    } else if (m == 0xda) { // SOS: one interleaved scan with all components (what baseline encoders write)
      if (!haveFrame || n < 1 || b[0] != comps.size() || n < 1 + 2 * comps.size() + 3) return false;
      for (size_t c = 0; c < comps.size(); c++) {
        JpegComp* cp = nullptr;
        for (JpegComp& k : comps) if (k.id == b[1 + 2 * c]) cp = &k;
        if (!cp) return false;
        cp->td = b[2 + 2 * c] >> 4; cp->ta = b[2 + 2 * c] & 15;
        if (cp->td > 3 || cp->ta > 3 || !hdc[cp->td].defined || !hac[cp->ta].defined || !qtDefined[cp->tq]) return false;
      }
      const uint32_t mcuW = 8u * (uint32_t)hmax, mcuH = 8u * (uint32_t)vmax, mcusX = (w + mcuW - 1) / mcuW, mcusY = (h + mcuH - 1) / mcuH;
      if ((uint64_t)mcusX * mcusY > (d.size() - pos) * 64ull + 64ull) return false; // an MCU takes at least a few bits: bound the work by the bytes present
      for (JpegComp& c : comps) { c.bw = (int)mcusX * c.h * 8; c.bh = (int)mcusY * c.v * 8; c.plane.assign((size_t)c.bw * c.bh, 128); c.pred = 0; }
      JpegBits br{d.data(), d.size(), pos + 2 + len};
      int untilRestart = restartInterval;
      for (uint32_t my = 0; my < mcusY; my++)
        for (uint32_t mx = 0; mx < mcusX; mx++) {
          if (restartInterval && untilRestart == 0) { // expect RSTn: realign, reset predictors
            br.cnt = 0; br.hitMarker = false;
            while (br.pos + 1 < br.n && !(br.d[br.pos] == 0xff && br.d[br.pos + 1] >= 0xd0 && br.d[br.pos + 1] <= 0xd7)) br.pos++;
            if (br.pos + 1 >= br.n) return false;
            br.pos += 2;
            for (JpegComp& c : comps) c.pred = 0;
            untilRestart = restartInterval;
          }
          for (JpegComp& c : comps)
            for (int by = 0; by < c.v; by++)
              for (int bx = 0; bx < c.h; bx++) {
                float coef[64] = {0.0f};
                const JpegHuff& hd = hdc[c.td]; const JpegHuff& ha = hac[c.ta];
                const int t = jpegDecodeSymbol(br, hd);
                if (t < 0 || t > 11) return false;
                c.pred += jpegExtend(br.receive(t), t);
                coef[0] = (float)(c.pred * (int)qt[c.tq][0]);
                for (int k = 1; k < 64;) {
                  const int rs = jpegDecodeSymbol(br, ha);
                  if (rs < 0) return false;
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz == 0) { if (r == 15) { k += 16; continue; } break; } // ZRL / EOB
                  k += r;
                  if (k > 63) return false;
                  coef[zigzag[k]] = (float)(jpegExtend(br.receive(sz), sz) * (int)qt[c.tq][zigzag[k]]);
                  k++;
                }
                float tmp[64]; // separable IDCT: rows (over u), then columns (over v)
                for (int v = 0; v < 8; v++)
                  for (int x = 0; x < 8; x++) { float a = 0.0f; for (int u = 0; u < 8; u++) a += coef[v * 8 + u] * cosTab[x][u]; tmp[v * 8 + x] = a; }
                const size_t ox = ((size_t)mx * c.h + bx) * 8, oy = ((size_t)my * c.v + by) * 8;
                for (int y = 0; y < 8; y++)
                  for (int x = 0; x < 8; x++) {
                    float a = 0.0f; for (int v = 0; v < 8; v++) a += tmp[v * 8 + x] * cosTab[y][v];
                    a = floorf(a + 128.5f);
                    c.plane[(oy + y) * (size_t)c.bw + ox + x] = (uint8_t)(a < 0.0f ? 0.0f : (a > 255.0f ? 255.0f : a));
                  }
              }
          if (restartInterval) untilRestart--;
        }
      // upsample by replication, convert, write row 0 = top row
      out.assign((size_t)w * h * 4, 1.0f);
      auto toLinear = [&](float c) { return srgbToLinear ? (c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f)) : c; };
      for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
          float s[3] = {0.0f, 128.0f, 128.0f};
          for (size_t c = 0; c < comps.size(); c++) s[c] = (float)comps[c].plane[((size_t)y * comps[c].v / vmax) * comps[c].bw + (size_t)x * comps[c].h / hmax];
          float r = s[0], g = s[0], bl = s[0];
          if (comps.size() == 3) { r = s[0] + 1.402f * (s[2] - 128.0f); g = (s[0] - 0.344136f * (s[1] - 128.0f)) - 0.714136f * (s[2] - 128.0f);
              bl = s[0] + 1.772f * (s[1] - 128.0f); }
          float* o = &out[((size_t)y * w + x) * 4];
          const float rgb[3] = {r, g, bl};
          for (int k = 0; k < 3; k++) { float v8 = floorf(rgb[k] + 0.5f); v8 = v8 < 0.0f ? 0.0f : (v8 > 255.0f ? 255.0f : v8); o[k] = toLinear(v8 / 255.0f); }
        }
      return true;
    }
    pos += 2 + len;
  }
  return false; // no scan
}

} // namespace

bool decodeImageBytes(const uint8_t* bytes, size_t size, bool srgbToLinear, uint32_t& w, uint32_t& h, std::vector<float>& out)
{
  if (!bytes || size < 4) return false;
  const std::vector<uint8_t> d(bytes, bytes + size);
  try {
    bool ok;
    if (d.size() >= 8 && d[0] == 0x89 && d[1] == 'P') ok = decodePng(d, srgbToLinear, w, h, out);
    else if (d.size() >= 4 && d[0] == 0xff && d[1] == 0xd8) ok = decodeJpeg(d, srgbToLinear, w, h, out);
    else ok = decodeHdrOrPfm(d, w, h, out);
    if (!ok) return false;
    // imgio's orientation: every decoder ends with _FlipImage, so row 0 of a loaded image is the file's BOTTOM scanline
    // (src/imgio/impl/PngDecoder.cpp:78, HdrDecoder.cpp:41, JpegDecoder.cpp; pinned by REF_4C, src/imgio/impl/main.cpp:53-61) and
    // texture coordinate v = 0 addresses the bottom of the picture.  The decoders above produce file order; flip once here.
    const size_t rowFloats = (size_t)w * 4u;
    for (uint32_t y = 0; y < h / 2u; y++) std::swap_ranges(out.begin() + (size_t)y * rowFloats, out.begin() + (size_t)(y + 1u) * rowFloats,
        out.begin() + (size_t)(h - 1u - y) * rowFloats);
    return true;
  } catch (const std::exception&) { return false; } // allocation failure: the caller reports "cannot decode"
}

bool readFileBytes(const char* path, std::vector<uint8_t>& d)
{
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  { uint8_t buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + n); }
  fclose(f);
  return true;
}

// 8-bit sRGB -> linear with the PNG decoder's arithmetic (an image handed over as RGBA8
// by an external decoder equals the in-library decode of the same PNG bit for bit)
float srgb8ToLinear(uint8_t v) { const float c = (float)v / 255.0f; return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f); }

} // namespace gi
