// gi_oracle.cpp -- CPU ORACLE (test infrastructure only; see gi_oracle.h header comment).
//
// Every function cites the reference file:line whose behaviour it restates.  Written from
// scratch; no reference source is copied.  Build: see oracle/Makefile
// (g++ -O2 -ffp-contract=off -fno-fast-math: the arithmetic contract forbids contraction).
//
// "parity unpinned" for BSDF arithmetic (MDL SDK / MaterialX are not in /root/reference) and the HW traversal's
// tie-breaking; see gi_oracle.h.  PINNED against the reference's own code: the helpers exported as orc_* / orc_dbg_*
// (RNG, hashes, orthonormal basis, ray offset, octahedral codec, sampling maps, colour maps, payload bit fields, Russian
// roulette, volume sampling, dome rotation, wrap / crop, normal adaption, sampleLight) are compared bit for bit -- within a
// few ulp where sin / cos / log are involved -- with those functions compiled from /root/reference/src/gi/shaders
// (oracle/ref/build_ref.py -> oracle/_ref/libgi_ref.so, tests/test_oracle_ref.py); the render loop as a whole is compared with the
// reference's rgen / chit / miss shaders run on the CPU (oracle/ref/ref_loop.cpp, tests/test_oracle_ref_loop.py).

#include "gi_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// vec3 helpers.  Operation order is part of the arithmetic contract.
// ---------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(const float* p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalize(V3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
inline float fmax2(float a, float b) { return a > b ? a : b; } // GLSL max(a,b) = a<b ? b : a (NaN-agnostic here)
inline float fmin2(float a, float b) { return a < b ? a : b; }

inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

const float ORC_PI = 3.1415926535897932384626433832795f; // common.glsl:9
const float ORC_FLT_MAX = 3.402823466e38f;               // common.glsl:7
const float ORC_FLT_MIN = 1.175494351e-38f;              // common.glsl:8

// ---------------------------------------------------------------------------------------------
// Polynomial transcendentals (arithmetic contract).  Cephes single-precision kernels
// (public domain, S. Moshier) evaluated with plain mul/add.
// ---------------------------------------------------------------------------------------------
// sin/cos of 2*pi*x for x in [0,1].  Octant reduction is exact in fp32.
inline void sincos2pi(float x, float* s, float* c)
{
  float y = x * 8.0f;              // exact
  int q = (int)y;                  // 0..8
  int j = (q + 1) >> 1;            // 0..4, nearest even octant boundary / 2
  float z = y - (float)(2 * j);    // exact, in [-1,1]
  float t = z * 0.78539816339744830962f; // * pi/4
  float t2 = t * t;
  float sp = ((-1.9515295891e-4f * t2 + 8.3321608736e-3f) * t2 - 1.6666654611e-1f) * t2 * t + t;
  float cp = ((2.443315711809948e-5f * t2 - 1.388731625493765e-3f) * t2 + 4.166664568298827e-2f) * t2 * t2 - 0.5f * t2 + 1.0f;
  switch (j & 3) {
    case 0: *s = sp; *c = cp; break;
    case 1: *s = cp; *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
  }
}

// sin/cos of an arbitrary angle (radians) through the same kernel.
inline void sincosr(float a, float* s, float* c)
{
  float r = a * 0.15915494309189533577f; // 1/(2 pi)
  float f = r - floorf(r);               // [0,1)
  sincos2pi(f, s, c);
}

// natural log, x > 0 (denormals handled).
inline float logf_poly(float x)
{
  uint32_t b = f2u(x);
  int e = 0;
  if (b < 0x00800000u) { x = x * 16777216.0f; b = f2u(x); e = -24; }
  e += (int)((b >> 23) & 0xffu) - 126;
  float m = u2f((b & 0x007fffffu) | 0x3f000000u); // [0.5,1)
  if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
             + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
             + 3.3333331174e-1f) * m * z;
  float fe = (float)e;
  y = y + (-2.12194440e-4f * fe);
  y = y + (-0.5f * z);
  float r = m + y;
  r = r + 0.693359375f * fe;
  return r;
}

// exp(x) for x <= 0 (Beer-Lambert transmittance), Cephes expf kernel with plain mul/add.
inline float expf_poly(float x)
{
  if (x < -87.0f) return 0.0f;
  if (x > 0.0f) x = 0.0f;
  float fx = floorf(x * 1.44269504088896341f + 0.5f);
  x = x - fx * 0.693359375f;
  x = x - fx * -2.12194440e-4f;
  float z = x * x;
  float y = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
  return y * u2f((uint32_t)((int)fx + 127) << 23);
}

// atan2(y, x) and acos(x) for the equirectangular dome lookup (rp_main.miss:46-53): Cephes atanf / asinf kernels with
// plain mul/add (part of the arithmetic contract; |err| < 3e-7).
inline float atanf_poly(float xx)
{
  float x = fabsf(xx), y;
  if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
  else y = 0.0f;
  float z = x * x;
  y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
  return xx < 0.0f ? -y : y;
}
inline float atan2f_poly(float y, float x)
{
  if (x == 0.0f) return y > 0.0f ? 1.5707963267948966f : (y < 0.0f ? -1.5707963267948966f : 0.0f);
  float z = atanf_poly(y / x);
  if (x < 0.0f) z = z + (y >= 0.0f ? 3.14159265358979323846f : -3.14159265358979323846f);
  return z;
}
inline float asinf_poly(float xx)
{
  float a = fabsf(xx), x, z; bool flag = false;
  if (a > 0.5f) { z = 0.5f * (1.0f - a); x = sqrtf(z); flag = true; }
  else { x = a; z = x * x; }
  z = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
  if (flag) { z = z + z; z = 1.5707963267948966f - z; }
  return xx < 0.0f ? -z : z;
}
inline float acosf_poly(float x)
{
  x = fmin2(fmax2(x, -1.0f), 1.0f);
  if (x < -0.5f) return 3.14159265358979323846f - 2.0f * asinf_poly(sqrtf(0.5f * (1.0f + x)));
  if (x > 0.5f) return 2.0f * asinf_poly(sqrtf(0.5f * (1.0f - x)));
  return 1.5707963267948966f - asinf_poly(x);
}

// ---------------------------------------------------------------------------------------------
// half <-> float (glm::packHalf2x16 / GLSL unpackHalf2x16), round-to-nearest-even.
// ---------------------------------------------------------------------------------------------
inline uint16_t f32_to_f16(float f)
{
  uint32_t x = f2u(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t absx = x & 0x7fffffffu;
  if (absx >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u : 0u));
  if (absx >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); // overflow -> inf (>= 65520)
  if (absx < 0x33000001u) return (uint16_t)sign;              // underflow -> 0 (<= 2^-25)
  int exp = (int)(absx >> 23) - 127;
  uint32_t man = (absx & 0x7fffffu) | 0x800000u;
  if (exp < -14) { // subnormal half
    int shift = -14 - exp + 13;
    uint32_t r = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    return (uint16_t)(sign | r);
  }
  uint32_t r = ((uint32_t)(exp + 15) << 10) | ((man >> 13) & 0x3ffu);
  uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
  return (uint16_t)(sign | r);
}

inline float f16_to_f32(uint16_t h)
{
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  if (exp == 0) {
    if (man == 0) return u2f(sign);
    float v = (float)man * 5.9604644775390625e-8f; // 2^-24
    return (sign ? -v : v);
  }
  if (exp == 31) return u2f(sign | 0x7f800000u | (man << 13));
  return u2f(sign | ((exp + 112u) << 23) | (man << 13));
}

inline uint32_t pack_half2x16(float a, float b) { return (uint32_t)f32_to_f16(a) | ((uint32_t)f32_to_f16(b) << 16); }

// ---------------------------------------------------------------------------------------------
// common.glsl restatement
// ---------------------------------------------------------------------------------------------
// common.glsl:74-82 (hash_theironborn)
inline uint32_t hash_init(uint32_t x)
{
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0xd35a2d97u; x ^= x >> 15;
  return x;
}
// common.glsl:85-90; note the inout state is LCG-advanced and then the caller overwrites it
// with the returned word (rng1d_next1f, common.glsl:92-96).
inline uint32_t hash_pcg32(uint32_t& state)
{
  state = state * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
  return (word >> 22) ^ word;
}
inline float uint_as_float01(uint32_t v) { return u2f(0x3f800000u | (v >> 9)) - 1.0f; } // common.glsl:44-47
inline float next1f(uint32_t& s) { s = hash_pcg32(s); return uint_as_float01(s); }
inline uint32_t rng_init(uint32_t pixelIndex, uint32_t sampleIndex) { return hash_init(pixelIndex * (sampleIndex + 1u)); } // :121-124

// common.glsl:128-137 (Duff et al.)
inline void orthonormal_basis(V3 n, V3& b1, V3& b2)
{
  float nsign = (n.z >= 0.0f) ? 1.0f : -1.0f;
  float a = -1.0f / (nsign + n.z);
  float b = n.x * n.y * a;
  b1 = v3(1.0f + nsign * n.x * n.x * a, nsign * b, -nsign * n.x);
  b2 = v3(b, nsign + n.y * n.y * a, -n.y);
}

// common.glsl:143-162 (Waechter-Binder with origin 1/32, floatScale 1/65536, intScale 64)
inline float offset_component(float p, float n)
{
  int io = (int)(n * 64.0f); // GLSL ivec3() truncates
  int32_t pi = (int32_t)f2u(p);
  int32_t moved = pi + ((p >= 0.0f) ? io : -io);
  float ip = u2f((uint32_t)moved);
  float fp = p + n * (1.0f / 65536.0f);
  return (fabsf(p) >= (1.0f / 32.0f)) ? ip : fp;
}
inline V3 offset_ray_origin(V3 p, V3 n) { return v3(offset_component(p.x, n.x), offset_component(p.y, n.y), offset_component(p.z, n.z)); }

// Gi.cpp:287-300 (_EncodeOctahedral/_EncodeDirection) with glm::packUnorm2x16 = round(clamp(v,0,1)*65535)
inline uint32_t encode_direction(V3 v)
{
  v = normalize(v);
  float s = fabsf(v.x) + fabsf(v.y) + fabsf(v.z);
  v = v / s;
  float px = v.x >= 0.0f ? 1.0f : -1.0f, py = v.y >= 0.0f ? 1.0f : -1.0f;
  float ex, ey;
  if (v.z < 0.0f) { ex = (1.0f - fabsf(v.y)) * px; ey = (1.0f - fabsf(v.x)) * py; } else { ex = v.x; ey = v.y; }
  ex = ex * 0.5f + 0.5f; ey = ey * 0.5f + 0.5f;
  ex = fmin2(fmax2(ex, 0.0f), 1.0f); ey = fmin2(fmax2(ey, 0.0f), 1.0f);
  uint32_t ux = (uint32_t)nearbyintf(ex * 65535.0f), uy = (uint32_t)nearbyintf(ey * 65535.0f);
  return ux | (uy << 16);
}

// common.glsl:181-207 (decode_octahedral/decode_direction); unpackUnorm2x16: c / 65535.0
inline V3 decode_direction(uint32_t e)
{
  float ex = (float)(e & 0xffffu) / 65535.0f, ey = (float)(e >> 16) / 65535.0f;
  ex = ex * 2.0f - 1.0f; ey = ey * 2.0f - 1.0f;
  V3 v = v3(ex, ey, 1.0f - fabsf(ex) - fabsf(ey));
  float t = fmax2(-v.z, 0.0f);
  v.x += (v.x >= 0.0f) ? -t : t;
  v.y += (v.y >= 0.0f) ? -t : t;
  return normalize(v);
}

inline float luminance(V3 c) { return dot(c, v3(0.2126f, 0.7152f, 0.0722f)); } // common.glsl:254-257
inline float safe_div(float a, float b) { return (b == 0.0f) ? 0.0f : (a / b); }        // :18-21
inline V3 safe_div(V3 v, float f) { return (f == 0.0f) ? v3(0, 0, 0) : (v / f); }      // :23-26

// common.glsl:210-220 (sample_hemisphere: cosine-weighted, z up)
inline V3 sample_hemisphere(float x0, float x1)
{
  float a = sqrtf(x0);
  float s, c; sincos2pi(x1, &s, &c);
  return v3(a * c, a * s, sqrtf(1.0f - x0));
}
// common.glsl:223-230 (sample_sphere)
inline V3 sample_sphere(float x0, float x1, V3 radius)
{
  float a = 1.0f - 2.0f * x0;
  float b = sqrtf(1.0f - a * a);
  float s, c; sincos2pi(x1, &s, &c);
  return v3(b * c, b * s, a) * radius;
}
// common.glsl:233-252 (sample_disk, concentric)
inline void sample_disk(float x0, float x1, float rx, float ry, float& ox, float& oy)
{
  float a = 2.0f * x0 - 1.0f, b = 2.0f * x1 - 1.0f;
  float r0, r1, phi;
  if ((a * a) > (b * b)) { r0 = rx * a; r1 = ry * a; phi = (ORC_PI / 4.0f) * (b / a); }
  else { r0 = rx * b; r1 = ry * b; phi = (ORC_PI / 2.0f) - (ORC_PI / 4.0f) * safe_div(a, b); }
  float s, c; sincosr(phi, &s, &c);
  ox = r0 * c; oy = r1 * s;
}

// rp_main.rgen:118-130 (fisGauss, Box-Muller, sigma 0.375)
inline void fis_gauss(float x0, float x1, float& ox, float& oy)
{
  float u1 = fmax2(1e-38f, x0);
  float r = 0.375f * sqrtf(-2.0f * logf_poly(u1));
  float s, c; sincos2pi(x1, &s, &c);
  ox = c * r; oy = s * r;
}

// ---------------------------------------------------------------------------------------------
// Scene preparation (host packing restated)
// ---------------------------------------------------------------------------------------------
struct FVertex { V3 pos; float bsign; uint32_t n, t; float u, v; }; // rp_main.h:58-64
struct Instance {
  uint32_t mesh; int32_t instanceId;
  float o2w[12]; // rows of the 3x4 object-to-world (Gi.cpp:1191)
  float w2o[9];  // inverse of the 3x3 part, row-major
};
struct Tri { V3 v0, e1, e2; uint32_t instance, prim; float cutout; /* mdl_cutout_opacity of the material; 1 = opaque */ int32_t opacityTexMat; /* >= 0: that material's opacity is textured */ };
struct MeshData { std::vector<FVertex> verts; const uint32_t* faces; uint32_t faceCount; int material; uint32_t flags; int32_t objectId; std::vector<uint8_t> faceIdData; uint32_t faceIdStride; const OrcMesh* src; };

// Light structs as the device sees them (rp_main.h:73-113), derived fields per Gi.cpp setters.
struct SphereL { V3 pos; uint32_t ds; V3 em; float area; V3 radius; };
struct DistantL { V3 dir; float angle; V3 em; uint32_t ds; float invPdf; };
struct RectL { V3 origin; float width; V3 em; float height; uint32_t t0, t1, ds; };
struct DiskL { V3 origin; float rx; V3 em; float ry; uint32_t t0, t1, ds; };

struct BvhNode { float lo[3], hi[3]; uint32_t left, count; }; // count>0: leaf [left,left+count)

struct Prepared {
  std::vector<MeshData> meshes;
  std::vector<Instance> instances;
  std::vector<Tri> tris;
  std::vector<SphereL> sphere; std::vector<DistantL> distant; std::vector<RectL> rect; std::vector<DiskL> disk;
  const OrcMaterial* materials; uint32_t materialCount;
  const OrcTexture* textures; uint32_t textureCount; const OrcDomeLight* dome;
  std::vector<BvhNode> bvh; std::vector<uint32_t> bvhTris; // used when tris.size() > 64
};

// world = local * M_prim * M_instance (USD row vectors; Gi.cpp:641-658, 1191).  Returns the 3x4 affine
// A (column-vector form) as 12 floats, rows first.  glm mat4*mat4 order of operations.
void compose_transform(const float* prim, const float* inst, float out[12])
{
  // P[r][c] row-major USD matrices.  Column-vector form: A = (P_prim * P_inst)^T.
  // (P_prim*P_inst)[r][c] = sum_k prim[r][k]*inst[k][c]; A[c][r] = that.
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 4; r++) {
      float acc = prim[r * 4 + 0] * inst[0 * 4 + c];
      acc = acc + prim[r * 4 + 1] * inst[1 * 4 + c];
      acc = acc + prim[r * 4 + 2] * inst[2 * 4 + c];
      acc = acc + prim[r * 4 + 3] * inst[3 * 4 + c];
      out[c * 4 + r] = acc;
    }
}

void invert3x3(const float a[12], float inv[9])
{
  double m[3][3] = {{a[0], a[1], a[2]}, {a[4], a[5], a[6]}, {a[8], a[9], a[10]}};
  double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
  double c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
  double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  double det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
  double id = 1.0 / det;
  inv[0] = (float)(c00 * id);
  inv[1] = (float)((m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id);
  inv[2] = (float)((m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id);
  inv[3] = (float)(c01 * id);
  inv[4] = (float)((m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id);
  inv[5] = (float)((m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id);
  inv[6] = (float)(c02 * id);
  inv[7] = (float)((m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id);
  inv[8] = (float)((m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id);
}

// gl_ObjectToWorldEXT * vec4(p, w)
inline V3 xform_point(const float a[12], V3 p, float w)
{
  return v3(((a[0] * p.x + a[1] * p.y) + a[2] * p.z) + a[3] * w,
            ((a[4] * p.x + a[5] * p.y) + a[6] * p.z) + a[7] * w,
            ((a[8] * p.x + a[9] * p.y) + a[10] * p.z) + a[11] * w);
}
// v * mat3(gl_WorldToObjectEXT)  == transpose(inv3x3) * v
inline V3 xform_normal(const float w[9], V3 n)
{
  return v3((n.x * w[0] + n.y * w[3]) + n.z * w[6],
            (n.x * w[1] + n.y * w[4]) + n.z * w[7],
            (n.x * w[2] + n.y * w[5]) + n.z * w[8]);
}

void build_bvh(Prepared& P);
inline float cutout_opacity(const OrcMaterial& m);
inline float cutout_rule(uint32_t klass, float op, float th);

void prepare(const OrcScene* s, Prepared& P)
{
  P.materials = s->materials; P.materialCount = s->materialCount;
  P.textures = s->textures; P.textureCount = s->textureCount;
  P.dome = (s->dome && s->dome->texture >= 0 && (uint32_t)s->dome->texture < s->textureCount) ? s->dome : nullptr; // a dome light whose image failed to load is ignored (Gi.cpp:2221-2230)
  P.meshes.resize(s->meshCount);
  for (uint32_t mi = 0; mi < s->meshCount; mi++) {
    const OrcMesh& m = s->meshes[mi];
    MeshData& d = P.meshes[mi];
    d.faces = m.faces; d.faceCount = m.faceCount; d.material = m.material; d.src = &m;
    d.flags = (m.isLeftHanded ? 1u : 0u) | (m.isDoubleSided ? 2u : 0u); // rp_main.h:115-116
    d.objectId = m.id;
    d.faceIdStride = m.maxFaceId <= 255u ? 1u : (m.maxFaceId <= 65535u ? 2u : 4u); // Gi.cpp:878-885
    d.faceIdData.assign(((size_t)m.faceCount * d.faceIdStride + 3) / 4 * 4, 0);
    for (uint32_t i = 0; i < m.faceCount; i++) { int32_t fid = m.faceIds ? m.faceIds[i] : 0; memcpy(&d.faceIdData[(size_t)i * d.faceIdStride], &fid, d.faceIdStride); }
    d.verts.resize(m.vertexCount);
    for (uint32_t i = 0; i < m.vertexCount; i++) { // Gi.cpp:848-861
      const OrcVertex& v = m.vertices[i];
      d.verts[i] = FVertex{v3(v.pos), v.bitangentSign, encode_direction(v3(v.norm)), encode_direction(v3(v.tangent)), v.u, v.v};
    }
    if (!m.visible || m.faceCount == 0 || m.material < 0) continue; // Gi.cpp:801-804, 818-822, 834-837
    for (uint32_t ii = 0; ii < m.instanceCount; ii++) {             // Gi.cpp:1188-1202
      Instance inst; inst.mesh = mi; inst.instanceId = m.instanceIds ? m.instanceIds[ii] : (int32_t)ii;
      compose_transform(m.transform, m.instanceTransforms + 16 * ii, inst.o2w);
      invert3x3(inst.o2w, inst.w2o);
      uint32_t instIdx = (uint32_t)P.instances.size();
      P.instances.push_back(inst);
      for (uint32_t f = 0; f < m.faceCount; f++) {
        V3 p0 = xform_point(inst.o2w, d.verts[m.faces[3 * f + 0]].pos, 1.0f);
        V3 p1 = xform_point(inst.o2w, d.verts[m.faces[3 * f + 1]].pos, 1.0f);
        V3 p2 = xform_point(inst.o2w, d.verts[m.faces[3 * f + 2]].pos, 1.0f);
        const OrcMaterial& mat = s->materials[m.material];
        const bool opTex = mat.tex[ORC_TEX_OPACITY].texture >= 0 && (uint32_t)mat.tex[ORC_TEX_OPACITY].texture < s->textureCount;
        P.tris.push_back(Tri{p0, p1 - p0, p2 - p0, instIdx, f, cutout_opacity(mat), opTex ? (int32_t)m.material : -1});
      }
    }
  }
  for (uint32_t i = 0; i < s->sphereLightCount; i++) { // Gi.cpp:2573-2666
    const OrcSphereLight& l = s->sphereLights[i];
    float ab = powf(l.radius[0] * l.radius[1], 1.6f), ac = powf(l.radius[0] * l.radius[2], 1.6f), bc = powf(l.radius[1] * l.radius[2], 1.6f);
    float area = (float)(powf((ab + ac + bc) / 3.0f, 1.0f / 1.6f) * 4.0f * M_PI);
    P.sphere.push_back(SphereL{v3(l.pos), pack_half2x16(l.diffuse, l.specular), v3(l.baseEmission), area, v3(l.radius)});
  }
  for (uint32_t i = 0; i < s->distantLightCount; i++) { // Gi.cpp:2668-2746
    const OrcDistantLight& l = s->distantLights[i];
    float half = 0.5f * l.angle;
    float invPdf = (half > 0.0f) ? (float)(2.0f * M_PI * (1.0f - cosf(half))) : 1.0f;
    P.distant.push_back(DistantL{v3(l.direction), l.angle, v3(l.baseEmission), pack_half2x16(l.diffuse, l.specular), invPdf});
  }
  for (uint32_t i = 0; i < s->rectLightCount; i++) { // Gi.cpp:2748-2850
    const OrcRectLight& l = s->rectLights[i];
    P.rect.push_back(RectL{v3(l.origin), l.width, v3(l.baseEmission), l.height, encode_direction(v3(l.t0)), encode_direction(v3(l.t1)), pack_half2x16(l.diffuse, l.specular)});
  }
  for (uint32_t i = 0; i < s->diskLightCount; i++) { // Gi.cpp:2852-2940
    const OrcDiskLight& l = s->diskLights[i];
    P.disk.push_back(DiskL{v3(l.origin), l.radiusX, v3(l.baseEmission), l.radiusY, encode_direction(v3(l.t0)), encode_direction(v3(l.t1)), pack_half2x16(l.diffuse, l.specular)});
  }
  if (P.tris.size() > 64) build_bvh(P);
}

// ---------------------------------------------------------------------------------------------
// Traversal.  The reference uses the HW (traceRayEXT, rp_main.rgen:381-393/412-424); this is
// the software contract: two-sided Moeller-Trumbore in world space on pre-transformed
// triangles, accept tMin < t < tBest, ties broken towards the lower global triangle index.
// ---------------------------------------------------------------------------------------------
struct Hit { float t, u, v; uint32_t tri; };

// mdl_cutout_opacity of the closed forms: UsdPreviewSurface opacity with the opacityThreshold switch, OpenPBR geometry_opacity
inline float cutout_rule(uint32_t klass, float op, float th)
{
  if (klass == ORC_MAT_OPEN_PBR) return fmin2(fmax2(op, 0.0f), 1.0f);
  if (th > 0.0f) return (op >= th) ? 1.0f : 0.0f;
  return fmin2(fmax2(op, 0.0f), 1.0f);
}
inline float cutout_opacity(const OrcMaterial& m) { return cutout_rule(m.klass, m.p[ORC_P_OPACITY], m.p[ORC_P_OPACITY_THRESHOLD]); }
// The same with a textured opacity input, evaluated at the CANDIDATE's st (rp_main.ahit:51-60 sets up the shading state of the
// candidate hit and evaluates the material's cutout expression there); defined after the texture runtime.
float cutout_opacity_textured(const struct Prepared& P, const struct Tri& T, float u, float v);
// Any-hit randomness (rp_main.ahit:51-60 draws next1f per candidate, in the driver's traversal order -- the one draw whose
// order the reference leaves implementation-defined).  Restated order-independently: a stateless
// hash of the path's rng state and the candidate's scene-order triangle id; the state itself is not advanced.
inline float cutout_random(uint32_t rng, uint32_t triId)
{
  uint32_t st = (rng ^ (triId * 0x9e3779b9u + 0x85ebca6bu)) * 747796405u + 2891336453u;
  uint32_t word = ((st >> ((st >> 28) + 4u)) ^ st) * 277803737u;
  return uint_as_float01((word >> 22) ^ word);
}

inline bool tri_test(const Prepared& P, const Tri& T, V3 o, V3 d, float tMin, float& tBest, uint32_t idx, Hit& h, uint32_t rng)
{
  V3 pv = cross(d, T.e2);
  float det = dot(T.e1, pv);
  if (det == 0.0f) return false;
  float inv = 1.0f / det;
  V3 tv = o - T.v0;
  float u = dot(tv, pv) * inv;
  if (!(u >= 0.0f)) return false;
  V3 qv = cross(tv, T.e1);
  float v = dot(d, qv) * inv;
  if (!(v >= 0.0f) || !(u + v <= 1.0f)) return false;
  float t = dot(T.e2, qv) * inv;
  if (!(t > tMin)) return false;
  if (t < tBest || (t == tBest && h.tri != 0xffffffffu && idx < h.tri)) {
    if (T.opacityTexMat >= 0) { if (cutout_random(rng, idx) > cutout_opacity_textured(P, T, u, v)) return false; }
    else if (T.cutout < 1.0f && cutout_random(rng, idx) > T.cutout) return false; // ignoreIntersectionEXT (rp_main.ahit:57-60)
    tBest = t; h = Hit{t, u, v, idx}; return true;
  }
  return false;
}

// (nodes are handed out by an atomic counter from a vector sized up front, so the two halves of a large range can be built by different threads: the TREE is the
// same whatever the thread timing -- splits depend on the range alone -- only node numbers differ, and nothing depends on those)
// (the counter belongs to ONE build: two scenes may be prepared at the same time -- tests/fuzz_parity.py --concurrent found it shared)
void build_bvh_rec(Prepared& P, std::atomic<uint32_t>& next, uint32_t node, uint32_t begin, uint32_t end, const std::vector<V3>& lo, const std::vector<V3>& hi, int depth = 0)
{
  BvhNode& n = P.bvh[node];
  for (int a = 0; a < 3; a++) { n.lo[a] = ORC_FLT_MAX; n.hi[a] = -ORC_FLT_MAX; }
  float clo[3] = {ORC_FLT_MAX, ORC_FLT_MAX, ORC_FLT_MAX}, chi[3] = {-ORC_FLT_MAX, -ORC_FLT_MAX, -ORC_FLT_MAX};
  for (uint32_t i = begin; i < end; i++) {
    uint32_t t = P.bvhTris[i];
    const float l[3] = {lo[t].x, lo[t].y, lo[t].z}, h[3] = {hi[t].x, hi[t].y, hi[t].z};
    for (int a = 0; a < 3; a++) {
      n.lo[a] = fmin2(n.lo[a], l[a]); n.hi[a] = fmax2(n.hi[a], h[a]);
      float c = 0.5f * (l[a] + h[a]); clo[a] = fmin2(clo[a], c); chi[a] = fmax2(chi[a], c);
    }
  }
  if (end - begin <= 4) { n.left = begin; n.count = end - begin; return; }
  int axis = 0; float ext = chi[0] - clo[0];
  for (int a = 1; a < 3; a++) if (chi[a] - clo[a] > ext) { ext = chi[a] - clo[a]; axis = a; }
  uint32_t mid = (begin + end) / 2;
  auto key = [&](uint32_t t) { const float l[3] = {lo[t].x, lo[t].y, lo[t].z}, h[3] = {hi[t].x, hi[t].y, hi[t].z}; return l[axis] + h[axis]; };
  std::nth_element(P.bvhTris.begin() + begin, P.bvhTris.begin() + mid, P.bvhTris.begin() + end,
                   [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
  uint32_t left = next.fetch_add(2u);
  P.bvh[node].left = left; P.bvh[node].count = 0;
  if (depth < 6 && end - begin > 200000u) { // a 10 M-triangle scene: 64 subtrees in parallel (the single-threaded build was most of the big-scene tests' time)
    std::thread other([&, left, begin, mid, depth] { build_bvh_rec(P, next, left, begin, mid, lo, hi, depth + 1); });
    build_bvh_rec(P, next, left + 1, mid, end, lo, hi, depth + 1);
    other.join();
    return;
  }
  build_bvh_rec(P, next, left, begin, mid, lo, hi, depth + 1);
  build_bvh_rec(P, next, left + 1, mid, end, lo, hi, depth + 1);
}

void build_bvh(Prepared& P)
{
  size_t n = P.tris.size();
  std::vector<V3> lo(n), hi(n);
  for (size_t i = 0; i < n; i++) {
    const Tri& T = P.tris[i];
    V3 a = T.v0, b = T.v0 + T.e1, c = T.v0 + T.e2;
    lo[i] = v3(fmin2(a.x, fmin2(b.x, c.x)), fmin2(a.y, fmin2(b.y, c.y)), fmin2(a.z, fmin2(b.z, c.z)));
    hi[i] = v3(fmax2(a.x, fmax2(b.x, c.x)), fmax2(a.y, fmax2(b.y, c.y)), fmax2(a.z, fmax2(b.z, c.z)));
    // generous conservative padding: the box test must never cull a triangle tri_test would accept
    V3 pad = v3(1e-4f * (1.0f + fabsf(hi[i].x - lo[i].x) + fabsf(hi[i].x)), 1e-4f * (1.0f + fabsf(hi[i].y - lo[i].y) + fabsf(hi[i].y)),
                1e-4f * (1.0f + fabsf(hi[i].z - lo[i].z) + fabsf(hi[i].z)));
    lo[i] = lo[i] - pad; hi[i] = hi[i] + pad;
  }
  P.bvhTris.resize(n);
  for (size_t i = 0; i < n; i++) P.bvhTris[i] = (uint32_t)i;
  P.bvh.assign(2 * n + 2, BvhNode{}); // a binary tree over n leaves of >= 1 triangle has < 2 n nodes
  std::atomic<uint32_t> next{1u};
  build_bvh_rec(P, next, 0, 0, (uint32_t)n, lo, hi);
  P.bvh.resize(next.load());
}

inline bool box_test(const BvhNode& n, V3 o, V3 inv, float tMin, float tMax)
{
  float t0 = tMin, t1 = tMax;
  const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
  for (int a = 0; a < 3; a++) {
    float ta = (n.lo[a] - oo[a]) * ii[a], tb = (n.hi[a] - oo[a]) * ii[a];
    if (ta > tb) { float x = ta; ta = tb; tb = x; }
    if (ta != ta || tb != tb) continue; // 0*inf: ray inside slab plane -> do not cull
    t0 = fmax2(t0, ta); t1 = fmin2(t1, tb);
  }
  return t0 <= t1 * 1.0001f + 1e-6f;
}

// box_test that also reports the entry distance (the same arithmetic)
inline bool box_entry(const BvhNode& n, V3 o, V3 inv, float tMin, float tMax, float& entry)
{
  float t0 = tMin, t1 = tMax;
  const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
  for (int a = 0; a < 3; a++) {
    float ta = (n.lo[a] - oo[a]) * ii[a], tb = (n.hi[a] - oo[a]) * ii[a];
    if (ta > tb) { float x = ta; ta = tb; tb = x; }
    if (ta != ta || tb != tb) continue;
    t0 = fmax2(t0, ta); t1 = fmin2(t1, tb);
  }
  entry = t0;
  return t0 <= t1 * 1.0001f + 1e-6f;
}
// Children are visited nearer first and a stacked node is dropped when its entry distance has been overtaken (r04: the unordered walk made the big-scene tests
// minutes long).  Hits do not depend on the visiting order (tMin < t < tBest, ties to the lower triangle index), and the set of nodes visited is a superset of the
// unordered walk's (a node that walk would enter passes both the test at push time and the distance check at pop time), so results are unchanged -- the golden
// fixtures pin that.  ANY: stop at the first accepted hit (the boolean is the same).
template <bool ANY>
bool trace_walk(const Prepared& P, V3 o, V3 d, float tMin, float tMax, Hit& h, uint32_t rng)
{
  float tBest = tMax; bool any = false; h.tri = 0xffffffffu;
  // accept tMin < t < tMax; h.tri == ~0 marks 'no hit yet' so the tie rule cannot admit t == tMax
  if (P.bvh.empty()) {
    for (uint32_t i = 0; i < P.tris.size(); i++) { any |= tri_test(P, P.tris[i], o, d, tMin, tBest, i, h, rng); if (ANY && any) return true; }
    return any;
  }
  V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  struct Item { uint32_t node; float entry; };
  Item stack[160]; int sp = 0;
  { float e; if (box_entry(P.bvh[0], o, inv, tMin, tBest, e)) stack[sp++] = Item{0u, e}; }
  while (sp) {
    const Item it = stack[--sp];
    if (it.entry > tBest * 1.0001f + 1e-6f) continue;
    const BvhNode& n = P.bvh[it.node];
    if (n.count) {
      for (uint32_t i = n.left; i < n.left + n.count; i++) { uint32_t t = P.bvhTris[i]; any |= tri_test(P, P.tris[t], o, d, tMin, tBest, t, h, rng); }
      if (ANY && any) return true;
    } else {
      float el, er;
      const bool hl = box_entry(P.bvh[n.left], o, inv, tMin, tBest, el), hr = box_entry(P.bvh[n.left + 1], o, inv, tMin, tBest, er);
      if (hl && hr) {
        if (el <= er) { stack[sp++] = Item{n.left + 1, er}; stack[sp++] = Item{n.left, el}; }
        else { stack[sp++] = Item{n.left, el}; stack[sp++] = Item{n.left + 1, er}; }
      } else if (hl) stack[sp++] = Item{n.left, el};
      else if (hr) stack[sp++] = Item{n.left + 1, er};
    }
  }
  return any;
}
bool trace_closest(const Prepared& P, V3 o, V3 d, float tMin, float tMax, Hit& h, uint32_t rng = 0u) { return trace_walk<false>(P, o, d, tMin, tMax, h, rng); }

// Shadow rays: gl_RayFlagsTerminateOnFirstHitEXT (rp_main.rgen:405); any hit in (tMin,tMax) occludes.
bool trace_any(const Prepared& P, V3 o, V3 d, float tMin, float tMax, uint32_t rng)
{
  Hit h; return trace_walk<true>(P, o, d, tMin, tMax, h, rng);
}

// ---------------------------------------------------------------------------------------------
// Shading state (mdl_shading_state.glsl:4-98)
// ---------------------------------------------------------------------------------------------
struct State {
  V3 normal, geomNormal, position, tangentU, tangentV; float u, v; bool frontFace;
  // Bsdf_sample_data.ior1 / ior2 (rp_main.chit:188-189): ior of the medium the ray travels in / of the other side;
  // < 0 = BSDF_USE_MATERIAL_IOR.  Defaults = empty medium stack (vacuum outside).
  float ior1 = 0.0f, ior2 = 0.0f;
  V3 cameraPosition = {0, 0, 0}; float frame = 0.0f; // ubo.cameraPosition / ubo.frame for the CAMERA_POSITION / FRAME scene-data names
  bool thinWalled = false; // mdl_thin_walled (rp_main.chit:155-157): both sides of the surface see the medium the ray travels in
  bool sssVolume = false;  // the render keeps a medium stack (mediumStackSize > 0): OpenPBR's volumetric subsurface_bsdf can be walked
  // OpenPBR geometry_coat_normal (open_pbr_surface.mtlx:87, 560): the coat lobe's own shading frame, set by resolve_material when the input is mapped
  bool hasCoatFrame = false; V3 coatNormal = {0, 0, 1}, coatTangentU = {1, 0, 0}, coatTangentV = {0, 1, 0};
  // renderer state of the hit for scene-data lookups (mdl_interface.glsl:281-301)
  const MeshData* mesh = nullptr; uint32_t prim = 0, hitIndices[3] = {0, 0, 0}; int32_t instanceId = 0; float bu = 0.0f, bv = 0.0f;
};
inline float relative_eta(const State& st, float materialEta)
{
  const bool outside = st.frontFace || st.thinWalled; // rp_main.chit:188-189
  float e1 = st.ior1 == 0.0f ? (outside ? 1.0f : -1.0f) : st.ior1;
  float e2 = st.ior2 == 0.0f ? (outside ? -1.0f : 1.0f) : st.ior2;
  if (e1 < 0.0f) e1 = materialEta;
  if (e2 < 0.0f) e2 = materialEta;
  return e2 / e1;
}

void setup_shading_state(const Prepared& P, const Hit& h, V3 rayDir, State& st, const MeshData*& meshOut)
{
  const Tri& T = P.tris[h.tri];
  const Instance& inst = P.instances[T.instance];
  const MeshData& m = P.meshes[inst.mesh];
  meshOut = &m;
  const FVertex& a = m.verts[m.faces[3 * T.prim + 0]];
  const FVertex& b = m.verts[m.faces[3 * T.prim + 1]];
  const FVertex& c = m.verts[m.faces[3 * T.prim + 2]];
  float bx = 1.0f - h.u - h.v, by = h.u, bz = h.v;                        // :17
  V3 localPos = (a.pos * bx + b.pos * by) + c.pos * bz;                  // :24
  st.position = xform_point(inst.o2w, localPos, 1.0f);                   // :25
  V3 gn = normalize(cross(b.pos - a.pos, c.pos - a.pos));                // :27
  gn = normalize(xform_normal(inst.w2o, gn));                            // :28
  V3 n0 = decode_direction(a.n), n1 = decode_direction(b.n), n2 = decode_direction(c.n); // :31-33
  V3 ln = normalize((n0 * bx + n1 * by) + n2 * bz);                      // :35
  V3 n = normalize(xform_normal(inst.w2o, ln));                          // :36
  st.frontFace = dot(gn, -rayDir) >= 0.0f;                               // :39
  if (!st.frontFace) { gn = -gn; n = -n; }                               // :41-45
  V3 t0 = decode_direction(a.t), t1 = decode_direction(b.t), t2 = decode_direction(c.t); // :48-50
  V3 lt = normalize((t0 * bx + t1 * by) + t2 * bz);                      // :52
  V3 tg = normalize(xform_point(inst.o2w, lt, 0.0f));                    // :53
  tg = normalize(tg - n * dot(tg, n));                                   // :56
  float bs = (bx * a.bsign + by * b.bsign) + bz * c.bsign;               // :58
  st.tangentU = tg; st.tangentV = cross(n, tg) * bs;                     // :59
  st.u = (bx * a.u + by * b.u) + bz * c.u; st.v = (bx * a.v + by * b.v) + bz * c.v; // :62-65
  st.normal = n; st.geomNormal = gn;
  st.mesh = &m; st.prim = T.prim; st.instanceId = inst.instanceId; st.bu = h.u; st.bv = h.v;
  st.hitIndices[0] = m.faces[3 * T.prim + 0]; st.hitIndices[1] = m.faces[3 * T.prim + 1]; st.hitIndices[2] = m.faces[3 * T.prim + 2];
}

// ---------------------------------------------------------------------------------------------
// Texture runtime (mdl_interface.glsl:8-38 apply_wrap_and_crop, :127-145 tex_lookup_float4_2d) over a software
// sampler: bilinear, REPEAT addressing, LOD 0 (Gi.cpp:388-392; CgpuVk.cpp:1985-1990).  Filter weights are fp32 here
// (the hardware's are 8-bit fixed point: unpinned).
// ---------------------------------------------------------------------------------------------
struct F4v { float x, y, z, w; };
inline F4v sample_bilinear_repeat(const OrcTexture& t, float u, float v)
{
  u = u - floorf(u); v = v - floorf(v);
  float x = u * (float)t.width - 0.5f, y = v * (float)t.height - 0.5f;
  float x0f = floorf(x), y0f = floorf(y);
  float fx = x - x0f, fy = y - y0f;
  int w = (int)t.width, h = (int)t.height;
  int ix0 = (int)x0f, iy0 = (int)y0f;
  if (ix0 < 0) ix0 += w;
  if (iy0 < 0) iy0 += h;
  int ix1 = ix0 + 1; if (ix1 >= w) ix1 -= w;
  int iy1 = iy0 + 1; if (iy1 >= h) iy1 -= h;
  const float* t00 = t.rgba + 4 * ((size_t)iy0 * w + ix0); const float* t10 = t.rgba + 4 * ((size_t)iy0 * w + ix1);
  const float* t01 = t.rgba + 4 * ((size_t)iy1 * w + ix0); const float* t11 = t.rgba + 4 * ((size_t)iy1 * w + ix1);
  float gx = 1.0f - fx, gy = 1.0f - fy, o[4];
  for (int c = 0; c < 4; c++) {
    float top = t00[c] * gx + t10[c] * fx, bot = t01[c] * gx + t11[c] * fx;
    o[c] = top * gy + bot * fy;
  }
  return F4v{o[0], o[1], o[2], o[3]};
}
inline float apply_wrap_and_crop(float coord, int wrap, int res) // crop = (0, 1)
{
  if (wrap == ORC_TEX_WRAP_REPEAT) coord = coord - floorf(coord);
  else {
    if (wrap == ORC_TEX_WRAP_MIRRORED_REPEAT) {
      float tmp = floorf(coord);
      if (((int)tmp & 1) != 0) coord = 1.0f - (coord - tmp); else coord = coord - tmp;
    }
    float inv_hdim = 0.5f / (float)res;
    coord = fmin2(fmax2(coord, inv_hdim), 1.0f - inv_hdim);
  }
  return coord;
}
// UsdTransform2d between the primvar reader and a UsdUVTexture's `st` (UsdPreviewSurface specification: result = in * scale, rotated counter-clockwise by
// `rotation` degrees, + translation), folded by the front end into six floats; the reference compiles the node through MaterialX -> MDL
// (src/mc/impl/MtlxMdlCodeGen.cpp:186-215).  Fixed association, no contraction.
inline void tex_transform_st(const OrcTexBinding& b, float& u, float& v)
{
  if (!b.hasTransform) return;
  const float s = u, t = v;
  u = (b.xf[0] * s + b.xf[1] * t) + b.xf[2];
  v = (b.xf[3] * s + b.xf[4] * t) + b.xf[5];
}
inline F4v tex_lookup_float4_2d(const OrcTexture& t, float u, float v, int wrapU, int wrapV)
{
  if ((wrapU == ORC_TEX_WRAP_CLIP && (u < 0.0f || u > 1.0f)) || (wrapV == ORC_TEX_WRAP_CLIP && (v < 0.0f || v > 1.0f))) return F4v{0, 0, 0, 0};
  u = apply_wrap_and_crop(u, wrapU, (int)t.width);
  v = apply_wrap_and_crop(v, wrapV, (int)t.height);
  return sample_bilinear_repeat(t, u, v);
}

// The remaining texture entry points of the MDL renderer runtime.  Only MDL-generated code calls them (none of the closed forms does); restated so the
// runtime is complete, held against the reference's text in tests/test_oracle_ref.py and against the device in tests/test_gpu_parity.py.
// tex_texel_float4_2d (mdl_interface.glsl:167-186): integer texel fetch, (0,0,0,0) for the invalid texture and outside the image
inline F4v tex_texel_float4_2d(const OrcTexture* t, int x, int y)
{
  if (!t) return F4v{0, 0, 0, 0};
  if (x < 0 || x >= (int)t->width || y < 0 || y >= (int)t->height) return F4v{0, 0, 0, 0};
  const float* p = t->rgba + 4 * ((size_t)y * t->width + (size_t)x);
  return F4v{p[0], p[1], p[2], p[3]};
}
// tex_resolution_2d (:208-221)
inline void tex_resolution_2d(const OrcTexture* t, int out[2]) { out[0] = t ? (int)t->width : 0; out[1] = t ? (int)t->height : 0; }
// 3-D textures (:45-65, 86-105): width x height x depth texels, slice by slice; the sampler is the 2-D one's trilinear extension (D5: software weights)
struct Tex3 { const float* rgba; uint32_t width, height, depth; };
inline F4v sample_trilinear_repeat(const Tex3& t, float u, float v, float w)
{
  w = w - floorf(w);
  const float z = w * (float)t.depth - 0.5f, z0f = floorf(z), fz = z - z0f, gz = 1.0f - fz;
  int d = (int)t.depth, iz0 = (int)z0f;
  if (iz0 < 0) iz0 += d;
  int iz1 = iz0 + 1; if (iz1 >= d) iz1 -= d;
  const size_t slice = (size_t)t.width * t.height * 4;
  const OrcTexture s0{t.rgba + slice * (size_t)iz0, t.width, t.height}, s1{t.rgba + slice * (size_t)iz1, t.width, t.height};
  const F4v a = sample_bilinear_repeat(s0, u, v), b = sample_bilinear_repeat(s1, u, v);
  return F4v{a.x * gz + b.x * fz, a.y * gz + b.y * fz, a.z * gz + b.z * fz, a.w * gz + b.w * fz};
}
inline F4v tex_lookup_float4_3d(const Tex3* t, float u, float v, float w, int wrapU, int wrapV, int wrapW)
{
  if (!t || (wrapU == ORC_TEX_WRAP_CLIP && (u < 0.0f || u > 1.0f)) || (wrapV == ORC_TEX_WRAP_CLIP && (v < 0.0f || v > 1.0f)) ||
      (wrapW == ORC_TEX_WRAP_CLIP && (w < 0.0f || w > 1.0f))) return F4v{0, 0, 0, 0};
  u = apply_wrap_and_crop(u, wrapU, (int)t->width);
  v = apply_wrap_and_crop(v, wrapV, (int)t->height);
  w = apply_wrap_and_crop(w, wrapW, (int)t->depth);
  return sample_trilinear_repeat(*t, u, v, w);
}
inline F4v tex_texel_float4_3d(const Tex3* t, int x, int y, int z)
{
  if (!t) return F4v{0, 0, 0, 0};
  if (x < 0 || x >= (int)t->width || y < 0 || y >= (int)t->height || z < 0 || z >= (int)t->depth) return F4v{0, 0, 0, 0};
  const float* p = t->rgba + 4 * (((size_t)z * t->height + (size_t)y) * t->width + (size_t)x);
  return F4v{p[0], p[1], p[2], p[3]};
}
// scene_data_lookup_float4x4 (:476-479): "return default_value; // TODO: not implemented" in the reference -- restated as such
inline void scene_data_lookup_float4x4(const float defaultValue[16], float out[16]) { memcpy(out, defaultValue, 64); }

// mdl_adapt_normal (mdl_interface.glsl:238-256): Iray's shadow-terminator bend of a mapped normal
inline V3 adapt_normal(V3 rayDir, V3 geomNormal, V3 normal)
{
  float dn = dot(rayDir, normal);
  V3 r = normalize(rayDir - normal * (2.0f * dn)); // reflect(I, N)
  float a = fmax2(0.0f, dot(r, -geomNormal));
  float b = dot(normal, geomNormal);
  V3 tangent = normalize(r + normal * (a / b));
  return normalize(-rayDir + tangent);
}

float cutout_opacity_textured(const Prepared& P, const Tri& T, float hu, float hv)
{
  const OrcMaterial& m = P.materials[T.opacityTexMat];
  const OrcTexBinding& b = m.tex[ORC_TEX_OPACITY];
  const MeshData& md = P.meshes[P.instances[T.instance].mesh];
  const FVertex& a = md.verts[md.faces[3 * T.prim + 0]], &bb = md.verts[md.faces[3 * T.prim + 1]], &c = md.verts[md.faces[3 * T.prim + 2]];
  const float bx = 1.0f - hu - hv, by = hu, bz = hv;                       // mdl_shading_state.glsl:17
  const float u = (bx * a.u + by * bb.u) + bz * c.u, v = (bx * a.v + by * bb.v) + bz * c.v; // :62-65
  float tu = u, tv = v; tex_transform_st(b, tu, tv);
  const F4v t = tex_lookup_float4_2d(P.textures[b.texture], tu, tv, b.wrapS, b.wrapT);
  const float val[4] = {t.x * b.scale[0] + b.bias[0], t.y * b.scale[1] + b.bias[1], t.z * b.scale[2] + b.bias[2], t.w * b.scale[3] + b.bias[3]};
  return cutout_rule(m.klass, val[b.channel & 3], m.p[ORC_P_OPACITY_THRESHOLD]);
}
// Per-hit material: the parameter block with its textured inputs evaluated at the hit's uv (UsdUVTexture: texel * scale + bias);
// a normal map replaces the shading normal (tangent space -> world, adapt_normal, tangent frame re-orthonormalised).
inline int shade_slot(int k) { return k < ORC_TEX_OPACITY ? k : k + 2; } // the slots resolved per hit: 0..4, then transmission weight / colour (opacity: any-hit test; coat normal: before the base normal)
const int SHADE_SLOT_COUNT = 7;
inline bool material_textured(const OrcMaterial& m) { for (int k = 0; k < SHADE_SLOT_COUNT; k++) { const int i = shade_slot(k); if (m.tex[i].texture >= 0 || m.primvarInput[i][0]) return true; } return m.tex[ORC_TEX_COAT_NORMAL].texture >= 0; }
// The primvar a scene-data name resolves to for this mesh: instancer primvars first, mesh primvars override (Gi.cpp:913-929)
inline const OrcPrimvar* find_primvar(const OrcMesh& m, const char* name)
{
  const OrcPrimvar* found = nullptr;
  for (uint32_t i = 0; i < m.instancerPrimvarCount; i++) if (!strncmp(m.instancerPrimvars[i].name, name, 64) && m.instancerPrimvars[i].floatCount) { found = &m.instancerPrimvars[i]; break; }
  for (uint32_t i = 0; i < m.primvarCount; i++) if (!strncmp(m.primvars[i].name, name, 64) && m.primvars[i].floatCount) { found = &m.primvars[i]; break; }
  return found;
}
// scene_data_lookup_float3 / _float (mdl_interface.glsl:337-371, 398-424) with get_scene_data_indices (:281-301)
inline bool scene_data_lookup(const State& st, const char* name, int comps, float out[3])
{
  // the two named scene data answered from the UBO (Frontend.cpp:251-252; mdl_interface.glsl:329-334 float3 only, :390-395 float only)
  if (comps == 3 && !strcmp(name, "CAMERA_POSITION")) { out[0] = st.cameraPosition.x; out[1] = st.cameraPosition.y; out[2] = st.cameraPosition.z; return true; }
  if (comps == 1 && !strcmp(name, "FRAME")) { out[0] = st.frame; return true; }
  const OrcPrimvar* pv = st.mesh ? find_primvar(*st.mesh->src, name) : nullptr;
  if (!pv) return false; // not found (SCENE_DATA_INVALID) -> default value
  const bool isInt = pv->type > ORC_PRIMVAR_VEC4;
  const uint32_t stride = (uint32_t)(isInt ? pv->type - ORC_PRIMVAR_INT : pv->type) + 1u;
  uint32_t idx[3];
  if (pv->interpolation == ORC_INTERP_UNIFORM) idx[0] = idx[1] = idx[2] = st.prim;
  else if (pv->interpolation == ORC_INTERP_INSTANCE) idx[0] = idx[1] = idx[2] = (uint32_t)st.instanceId;
  else if (pv->interpolation == ORC_INTERP_CONSTANT) idx[0] = idx[1] = idx[2] = 0u;
  else { idx[0] = st.hitIndices[0]; idx[1] = st.hitIndices[1]; idx[2] = st.hitIndices[2]; }
  const float bx = 1.0f - st.bu - st.bv, by = st.bu, bz = st.bv;
  if (isInt) { // scene_data_lookup_int / _int3 (mdl_interface.glsl:426-476): nearest vertex, per component (index_offset = component), as float
    const uint32_t pick = bx > by ? (bx > bz ? idx[0] : idx[2]) : (by > bz ? idx[1] : idx[2]);
    for (int c = 0; c < comps; c++) {
      size_t o = (size_t)pick * stride + (size_t)((uint32_t)c < stride ? (uint32_t)c : stride - 1u);
      int32_t iv = 0; if (o < pv->floatCount) memcpy(&iv, &pv->data[o], 4);
      out[c] = (float)iv;
    }
    return true;
  }
  for (int c = 0; c < comps; c++) {
    float v[3];
    for (int k = 0; k < 3; k++) { size_t o = (size_t)idx[k] * stride + (size_t)c; v[k] = o < pv->floatCount ? pv->data[o] : 0.0f; }
    out[c] = (v[0] * bx + v[1] * by) + v[2] * bz;
  }
  return true;
}
OrcMaterial resolve_material(const Prepared& P, const OrcMaterial& m, State& st, V3 rayDir)
{
  OrcMaterial r = m;
  { // geometry_coat_normal: a tangent-space map in the surface's own (unmapped) frame -> the coat lobe's frame; same treatment as the base normal map below
    const OrcTexBinding& b = m.tex[ORC_TEX_COAT_NORMAL];
    if (m.klass == ORC_MAT_OPEN_PBR && b.texture >= 0 && (uint32_t)b.texture < P.textureCount) {
      float tu = st.u, tv = st.v; tex_transform_st(b, tu, tv);
      const F4v t = tex_lookup_float4_2d(P.textures[b.texture], tu, tv, b.wrapS, b.wrapT);
      const float val[3] = {t.x * b.scale[0] + b.bias[0], t.y * b.scale[1] + b.bias[1], t.z * b.scale[2] + b.bias[2]};
      V3 n = normalize((st.tangentU * val[0] + st.tangentV * val[1]) + st.normal * val[2]);
      n = adapt_normal(rayDir, st.geomNormal, n);
      const float hs = dot(cross(st.normal, st.tangentU), st.tangentV) >= 0.0f ? 1.0f : -1.0f;
      const V3 tg = normalize(st.tangentU - n * dot(st.tangentU, n));
      st.hasCoatFrame = true; st.coatNormal = n; st.coatTangentU = tg; st.coatTangentV = cross(n, tg) * hs;
    }
  }
  for (int k = 0; k < SHADE_SLOT_COUNT; k++) { // ORC_TEX_OPACITY belongs to the any-hit test (cutout_opacity_textured)
    const int slot = shade_slot(k);
    const OrcTexBinding& b = m.tex[slot];
    if ((b.texture < 0 || (uint32_t)b.texture >= P.textureCount) && m.primvarInput[slot][0] && slot != ORC_TEX_NORMAL) { // primvar-driven input
      float v[3];
      const bool vec = slot == ORC_TEX_BASE_COLOR || slot == ORC_TEX_EMISSION || slot == ORC_TEX_TRANSMISSION_COLOR;
      if (scene_data_lookup(st, m.primvarInput[slot], vec ? 3 : 1, v)) {
        if (slot == ORC_TEX_BASE_COLOR) { r.p[ORC_P_BASE_COLOR] = v[0]; r.p[ORC_P_BASE_COLOR + 1] = v[1]; r.p[ORC_P_BASE_COLOR + 2] = v[2]; }
        else if (slot == ORC_TEX_EMISSION) { r.p[ORC_P_EMISSION] = v[0]; r.p[ORC_P_EMISSION + 1] = v[1]; r.p[ORC_P_EMISSION + 2] = v[2]; }
        else if (slot == ORC_TEX_TRANSMISSION_COLOR) { r.p[ORC_P_TRANSMISSION_COLOR] = v[0]; r.p[ORC_P_TRANSMISSION_COLOR + 1] = v[1]; r.p[ORC_P_TRANSMISSION_COLOR + 2] = v[2]; }
        else if (slot == ORC_TEX_ROUGHNESS) r.p[ORC_P_ROUGHNESS] = v[0];
        else if (slot == ORC_TEX_METALLIC) r.p[ORC_P_METALLIC] = v[0];
        else r.p[ORC_P_TRANSMISSION_WEIGHT] = v[0];
      }
      continue;
    }
    if (b.texture < 0 || (uint32_t)b.texture >= P.textureCount) continue;
    float tu = st.u, tv = st.v; tex_transform_st(b, tu, tv);
    F4v t = tex_lookup_float4_2d(P.textures[b.texture], tu, tv, b.wrapS, b.wrapT);
    float val[4] = {t.x * b.scale[0] + b.bias[0], t.y * b.scale[1] + b.bias[1], t.z * b.scale[2] + b.bias[2], t.w * b.scale[3] + b.bias[3]};
    if (slot == ORC_TEX_BASE_COLOR) { r.p[ORC_P_BASE_COLOR] = val[0]; r.p[ORC_P_BASE_COLOR + 1] = val[1]; r.p[ORC_P_BASE_COLOR + 2] = val[2]; }
    else if (slot == ORC_TEX_EMISSION) { r.p[ORC_P_EMISSION] = val[0]; r.p[ORC_P_EMISSION + 1] = val[1]; r.p[ORC_P_EMISSION + 2] = val[2]; }
    else if (slot == ORC_TEX_TRANSMISSION_COLOR) { r.p[ORC_P_TRANSMISSION_COLOR] = val[0]; r.p[ORC_P_TRANSMISSION_COLOR + 1] = val[1]; r.p[ORC_P_TRANSMISSION_COLOR + 2] = val[2]; }
    else if (slot == ORC_TEX_ROUGHNESS) r.p[ORC_P_ROUGHNESS] = val[b.channel & 3];
    else if (slot == ORC_TEX_METALLIC) r.p[ORC_P_METALLIC] = val[b.channel & 3];
    else if (slot == ORC_TEX_TRANSMISSION_WEIGHT) r.p[ORC_P_TRANSMISSION_WEIGHT] = val[b.channel & 3];
    else { // ORC_TEX_NORMAL
      V3 n = normalize((st.tangentU * val[0] + st.tangentV * val[1]) + st.normal * val[2]);
      n = adapt_normal(rayDir, st.geomNormal, n);
      float hs = dot(cross(st.normal, st.tangentU), st.tangentV) >= 0.0f ? 1.0f : -1.0f;
      V3 tg = normalize(st.tangentU - n * dot(st.tangentU, n));
      st.normal = n; st.tangentU = tg; st.tangentV = cross(n, tg) * hs;
    }
  }
  return r;
}

// ---------------------------------------------------------------------------------------------
// Closed-form BSDFs (replace the MDL-generated mdl_bsdf_scattering_* entry points,
// GlslShaderGen.cpp:181-193; contracts from mdl_types.glsl:158-238).  OUR definitions -- the
// reference arithmetic lives in the MDL SDK and is unpinned (DESIGN.md section "Materials").
// ---------------------------------------------------------------------------------------------
enum { EV_ABSORB = 0, EV_DIFFUSE = 1, EV_GLOSSY = 2, EV_SPECULAR = 4, EV_REFLECTION = 8, EV_TRANSMISSION = 16 }; // mdl_types.glsl:123-137
enum { EV_SUBSURFACE = 64 }; // [ours] beside EV_DIFFUSE | EV_TRANSMISSION: the path entered through the volumetric subsurface lobe -- the medium pushed is the subsurface medium

struct BsdfSample { V3 k2; V3 overPdf; float pdf; uint32_t event; };
struct BsdfEval { V3 diffuse, glossy; float pdf; };

inline V3 to_world(const State& st, V3 l) { return (st.tangentU * l.x + st.tangentV * l.y) + st.normal * l.z; }
inline V3 to_local(const State& st, V3 w) { return v3(dot(w, st.tangentU), dot(w, st.tangentV), dot(w, st.normal)); }
inline V3 to_world_coat(const State& st, V3 l) { return st.hasCoatFrame ? (st.coatTangentU * l.x + st.coatTangentV * l.y) + st.coatNormal * l.z : to_world(st, l); }
inline V3 to_local_coat(const State& st, V3 w) { return st.hasCoatFrame ? v3(dot(w, st.coatTangentU), dot(w, st.coatTangentV), dot(w, st.coatNormal)) : to_local(st, w); }
// geometry_coat_tangent: the turn applied to the local x / y of a direction in the coat's frame (c, s: opbr_params)
inline V3 coat_turn_local(float c, float s, V3 l) { return v3(l.x * c + l.y * s, l.y * c - l.x * s, l.z); }
inline V3 coat_turn_world(float c, float s, V3 l) { return v3(l.x * c - l.y * s, l.x * s + l.y * c, l.z); }

inline float schlick_w(float c) { float m = 1.0f - c; m = fmin2(fmax2(m, 0.0f), 1.0f); float m2 = m * m; return m2 * m2 * m; }
inline float ggx_lambda_term(float a2, float c) { return sqrtf(a2 + (1.0f - a2) * c * c); }

// GGX-Smith reflection lobe with VNDF sampling (Heitz 2018), isotropic alpha, local frame z = normal.
struct GgxOut { V3 l2; float pdf; float g2OverG1; float kh; bool valid; };
inline GgxOut ggx_sample(V3 l1, float alpha, float x0, float x1)
{
  GgxOut o; o.valid = false;
  V3 vh = normalize(v3(alpha * l1.x, alpha * l1.y, l1.z));
  float lensq = vh.x * vh.x + vh.y * vh.y;
  V3 T1 = lensq > 0.0f ? v3(-vh.y, vh.x, 0.0f) * (1.0f / sqrtf(lensq)) : v3(1.0f, 0.0f, 0.0f);
  V3 T2 = cross(vh, T1);
  float r = sqrtf(x0);
  float s, c; sincos2pi(x1, &s, &c);
  float t1 = r * c, t2 = r * s;
  float sm = 0.5f * (1.0f + vh.z);
  t2 = (1.0f - sm) * sqrtf(fmax2(0.0f, 1.0f - t1 * t1)) + sm * t2;
  V3 nh = (T1 * t1 + T2 * t2) + vh * sqrtf(fmax2(0.0f, (1.0f - t1 * t1) - t2 * t2));
  V3 h = normalize(v3(alpha * nh.x, alpha * nh.y, fmax2(0.0f, nh.z)));
  float kh = dot(l1, h);
  V3 l2 = h * (2.0f * kh) - l1;
  if (!(l2.z > 0.0f) || !(kh > 0.0f)) return o;
  float a2 = alpha * alpha;
  float nk1 = l1.z, nk2 = l2.z, nh2 = h.z * h.z;
  float dd = nh2 * (a2 - 1.0f) + 1.0f;
  float D = a2 / (ORC_PI * dd * dd);
  float L1 = ggx_lambda_term(a2, nk1), L2 = ggx_lambda_term(a2, nk2);
  float G1 = 2.0f * nk1 / (nk1 + L1);
  float G2 = 2.0f * nk1 * nk2 / (nk2 * L1 + nk1 * L2);
  o.l2 = l2; o.kh = kh; o.pdf = G1 * D / (4.0f * nk1); o.g2OverG1 = G2 / G1; o.valid = true;
  return o;
}
// returns f*cos (without Fresnel) and pdf for given directions.
inline void ggx_eval(V3 l1, V3 l2, float alpha, float& fcos, float& pdf, float& kh)
{
  fcos = 0.0f; pdf = 0.0f; kh = 0.0f;
  if (!(l1.z > 0.0f) || !(l2.z > 0.0f)) return;
  V3 h = normalize(l1 + l2);
  kh = dot(l1, h);
  float a2 = alpha * alpha;
  float nk1 = l1.z, nk2 = l2.z, nh2 = h.z * h.z;
  float dd = nh2 * (a2 - 1.0f) + 1.0f;
  float D = a2 / (ORC_PI * dd * dd);
  float L1 = ggx_lambda_term(a2, nk1), L2 = ggx_lambda_term(a2, nk2);
  float G1 = 2.0f * nk1 / (nk1 + L1);
  float G2 = 2.0f * nk1 * nk2 / (nk2 * L1 + nk1 * L2);
  fcos = D * G2 / (4.0f * nk1);
  pdf = G1 * D / (4.0f * nk1);
}

// Anisotropic form (specular_roughness_anisotropy / coat_roughness_anisotropy, open_pbr_surface.mtlx:27, 65, 133-136, 552-555): the same VNDF sampling with the
// view stretched by (ax, ay) along (tangentU, tangentV), D = 1 / (pi ax ay (hx^2/ax^2 + hy^2/ay^2 + hz^2)^2) and the Smith term of the stretched direction.
// Used only when ax != ay, so isotropic materials keep the arithmetic above bit for bit.
inline float ggx_lambda_xy(float ax, float ay, V3 v) { return sqrtf(((ax * v.x) * (ax * v.x) + (ay * v.y) * (ay * v.y)) + v.z * v.z); }
inline float ggx_d_xy(float ax, float ay, V3 h)
{
  const float hx = h.x / ax, hy = h.y / ay;
  const float dd = (hx * hx + hy * hy) + h.z * h.z;
  return 1.0f / (((ORC_PI * ax) * ay) * (dd * dd));
}
inline GgxOut ggx_sample_xy(V3 l1, float ax, float ay, float x0, float x1)
{
  GgxOut o; o.valid = false;
  V3 vh = normalize(v3(ax * l1.x, ay * l1.y, l1.z));
  float lensq = vh.x * vh.x + vh.y * vh.y;
  V3 T1 = lensq > 0.0f ? v3(-vh.y, vh.x, 0.0f) * (1.0f / sqrtf(lensq)) : v3(1.0f, 0.0f, 0.0f);
  V3 T2 = cross(vh, T1);
  float r = sqrtf(x0);
  float s, c; sincos2pi(x1, &s, &c);
  float t1 = r * c, t2 = r * s;
  float sm = 0.5f * (1.0f + vh.z);
  t2 = (1.0f - sm) * sqrtf(fmax2(0.0f, 1.0f - t1 * t1)) + sm * t2;
  V3 nh = (T1 * t1 + T2 * t2) + vh * sqrtf(fmax2(0.0f, (1.0f - t1 * t1) - t2 * t2));
  V3 h = normalize(v3(ax * nh.x, ay * nh.y, fmax2(0.0f, nh.z)));
  float kh = dot(l1, h);
  V3 l2 = h * (2.0f * kh) - l1;
  if (!(l2.z > 0.0f) || !(kh > 0.0f)) return o;
  float nk1 = l1.z, nk2 = l2.z;
  float D = ggx_d_xy(ax, ay, h);
  float L1 = ggx_lambda_xy(ax, ay, l1), L2 = ggx_lambda_xy(ax, ay, l2);
  float G1 = 2.0f * nk1 / (nk1 + L1);
  float G2 = 2.0f * nk1 * nk2 / (nk2 * L1 + nk1 * L2);
  o.l2 = l2; o.kh = kh; o.pdf = G1 * D / (4.0f * nk1); o.g2OverG1 = G2 / G1; o.valid = true;
  return o;
}
inline void ggx_eval_xy(V3 l1, V3 l2, float ax, float ay, float& fcos, float& pdf, float& kh)
{
  fcos = 0.0f; pdf = 0.0f; kh = 0.0f;
  if (!(l1.z > 0.0f) || !(l2.z > 0.0f)) return;
  V3 h = normalize(l1 + l2);
  kh = dot(l1, h);
  float nk1 = l1.z, nk2 = l2.z;
  float D = ggx_d_xy(ax, ay, h);
  float L1 = ggx_lambda_xy(ax, ay, l1), L2 = ggx_lambda_xy(ax, ay, l2);
  float G1 = 2.0f * nk1 / (nk1 + L1);
  float G2 = 2.0f * nk1 * nk2 / (nk2 * L1 + nk1 * L2);
  fcos = D * G2 / (4.0f * nk1);
  pdf = G1 * D / (4.0f * nk1);
}
// the lobe's (ax, ay) of OpenPBR's open_pbr_anisotropy node: alpha_t = r^2 sqrt(2 / (1 + (1 - a)^2)), alpha_b = (1 - a) alpha_t (a = 0: both r^2)
inline void opbr_anisotropy(float alpha, float a, float& ax, float& ay)
{
  ax = alpha; ay = alpha;
  if (!(a > 0.0f)) return;
  const float inv = 1.0f - fmin2(a, 1.0f);
  ax = fmax2(alpha * sqrtf(2.0f / (1.0f + inv * inv)), 0.001f); ay = fmax2(inv * ax, 0.001f);
}
inline GgxOut ggx_sample2(V3 l1, float ax, float ay, float x0, float x1) { return ax == ay ? ggx_sample(l1, ax, x0, x1) : ggx_sample_xy(l1, ax, ay, x0, x1); }
inline void ggx_eval2(V3 l1, V3 l2, float ax, float ay, float& fcos, float& pdf, float& kh) { if (ax == ay) ggx_eval(l1, l2, ax, fcos, pdf, kh); else ggx_eval_xy(l1, l2, ax, ay, fcos, pdf, kh); }

struct UpsParams { V3 albedo, F0; float alpha, coat, coatAlpha; };
inline UpsParams ups_params(const OrcMaterial& m)
{
  UpsParams u;
  V3 dc = v3(m.p + ORC_P_BASE_COLOR);
  float r = m.p[ORC_P_ROUGHNESS], cr = m.p[ORC_P_CLEARCOAT_ROUGHNESS];
  u.alpha = fmax2(r * r, 0.001f);
  u.coatAlpha = fmax2(cr * cr, 0.001f);
  u.coat = m.p[ORC_P_CLEARCOAT];
  if (m.p[ORC_P_USE_SPECULAR_WORKFLOW] != 0.0f) { u.F0 = v3(m.p + ORC_P_SPECULAR_COLOR); u.albedo = dc; }
  else {
    float ior = m.p[ORC_P_IOR], metal = m.p[ORC_P_METALLIC];
    float q = (1.0f - ior) / (1.0f + ior); float f0 = q * q;
    u.F0 = v3(f0, f0, f0) * (1.0f - metal) + dc * metal;
    u.albedo = dc * (1.0f - metal);
  }
  return u;
}
inline V3 schlick3(V3 F0, float c) { float w = schlick_w(c); return F0 + (v3(1, 1, 1) - F0) * w; }

// ---- class 2: OpenPBR (lobe graph of src/gi/mtlx/open_pbr_surface.mtlx:99-635, closed forms of our own) ----
// exact dielectric Fresnel, eta = n_t / n_i, c = cos of the incident angle
inline float fresnel_dielectric(float c, float eta)
{
  float sin2t = (1.0f - c * c) / (eta * eta);
  if (!(sin2t < 1.0f)) return 1.0f;
  float ct = sqrtf(1.0f - sin2t);
  float rs = (c - eta * ct) / (c + eta * ct);
  float rp = (eta * c - ct) / (eta * c + ct);
  return 0.5f * (rs * rs + rp * rp);
}
// generalized Schlick with the F82-tint edge colour (metal_bsdf, open_pbr_surface.mtlx:434-441)
inline V3 schlick_f82(V3 F0, V3 tint, float c)
{
  const float cb = 1.0f / 7.0f, w5 = 0.462664366f /* (6/7)^5 */, K = 17.6513846f /* 1 / (cb * (6/7)^6) */;
  V3 one = v3(1, 1, 1);
  V3 fb = F0 + (one - F0) * w5;
  V3 a = (fb * (one - tint)) * K;
  float m = 1.0f - c; m = fmin2(fmax2(m, 0.0f), 1.0f);
  float m2 = m * m, m5 = m2 * m2 * m, m6 = m5 * m;
  V3 f = (F0 + (one - F0) * m5) - a * (c * m6);
  (void)cb;
  return v3(fmin2(fmax2(f.x, 0.0f), 1.0f), fmin2(fmax2(f.y, 0.0f), 1.0f), fmin2(fmax2(f.z, 0.0f), 1.0f));
}
// Roughening of the base specular lobe under the coat (open_pbr_surface.mtlx:101-131):
// effective_specular_roughness = mix(specular_roughness, min(1, 2 coat_roughness^4 + specular_roughness^4)^(1/4), coat_weight)
inline float opbr_effective_roughness(float r, float cr, float coat)
{
  float c4 = (cr * cr) * (cr * cr), r4 = (r * r) * (r * r);
  float ra = sqrtf(sqrtf(fmin2(1.0f, 2.0f * c4 + r4)));
  return ra * coat + r * (1.0f - coat);
}
// Darkening of the base under the coat by internal reflections (:470-541): (1 - Kcoat) / (1 - Ebase Kcoat) blended in by coat_weight *
// coat_darkening, with Kcoat = 1 - (1 - coat_F0) / coat_ior^2 and Ebase = mix(base_color, base_color * specular_weight, base_metalness)
inline V3 opbr_base_darkening(V3 baseColor, float sw, float metalness, float coat, float coatF0, float cior, float coatDarkening)
{
  const float w = coat * coatDarkening;
  if (w == 0.0f) return v3(1, 1, 1);
  const float K = 1.0f - (1.0f - coatF0) / (cior * cior);
  const V3 Eb = (baseColor * sw) * metalness + baseColor * (1.0f - metalness);
  const float n = 1.0f - K;
  const V3 bd = v3(n / (1.0f - Eb.x * K), n / (1.0f - Eb.y * K), n / (1.0f - Eb.z * K));
  return bd * w + v3(1, 1, 1) * (1.0f - w);
}
// Emission seen through the coat (:590-619): mix(uncoated, coat_color * generalized_schlick_edf(color0 = 1 - coat_F0, color90 = 0,
// exponent 5), coat_weight); c = cos between the shading normal and the direction the light leaves in
inline V3 opbr_emission_factor(float coat, V3 coatColor, float coatF0, float c)
{
  if (coat == 0.0f) return v3(1, 1, 1);
  const float f = (1.0f - coatF0) * (1.0f - schlick_w(c));
  return (coatColor * f) * coat + v3(1, 1, 1) * (1.0f - coat);
}
// Energy-preserving Oren-Nayar (oren_nayar_diffuse_bsdf with energy_compensation, :200-206; Portsmouth, Kutz, Hill 2024: Fujii's
// qualitative model plus a multiple-scattering lobe, directional albedo by the paper's polynomial fit).  Returns pi * f for albedo rho.
inline float eon_albedo_fit(float mu, float r)
{
  const float mc = 1.0f - mu;
  const float G = mc * (0.0571085289f + mc * (0.491881867f + mc * (-0.332181442f + mc * 0.0714429953f)));
  return (1.0f + r * G) / (1.0f + 0.28779343f * r); // 0.5 - 2 / (3 pi)
}
inline V3 eon_pi_f(V3 rho, float r, V3 l1, V3 l2)
{
  const float mi = l1.z, mo = l2.z;
  const float s = l1.x * l2.x + l1.y * l2.y; // dot(wi, wo) - mu_i mu_o
  const float sOverT = s > 0.0f ? s / fmax2(mi, mo) : s;
  const float AF = 1.0f / (1.0f + 0.28779343f * r);
  const float ss = AF * (1.0f + r * sOverT);
  const float EFo = eon_albedo_fit(mo, r), EFi = eon_albedo_fit(mi, r);
  const float avgEF = AF * (1.0f + 0.07248828f * r); // 2/3 - 28 / (15 pi)
  const float ms = (fmax2(1e-7f, 1.0f - EFo) * fmax2(1e-7f, 1.0f - EFi)) / fmax2(1e-7f, 1.0f - avgEF);
  const V3 rr = rho * rho;
  const V3 rhoMs = v3((rr.x * avgEF) / (1.0f - rho.x * (1.0f - avgEF)), (rr.y * avgEF) / (1.0f - rho.y * (1.0f - avgEF)), (rr.z * avgEF) / (1.0f - rho.z * (1.0f - avgEF)));
  return rho * ss + rhoMs * ms;
}
struct OpbrParams { V3 albedo, metalTint, specColor, transTint, coatTint, sigmaA, sigmaS, baseColor, coatColor, ssColor, fuzzColor, sssSigmaS, sssSigmaT; bool ssVolume; float metalness, alpha, alphaY, coat, coatAlpha, coatAlphaY, coatF0, eta, tw, specWeight, anisotropy, baseWeight, diffRough, ssWeight, ssAniso, fuzzWeight, fuzzAlpha, filmWeight, filmNm, filmIor; bool thinWalled;  bool coatRot, specRot; float coatRotC, coatRotS, coatRelC, coatRelS, specRotC, specRotS; };
inline OpbrParams opbr_params(const OrcMaterial& m, bool sssVolume = false)
{
  OpbrParams o; const float* p = m.p;
  float bw = p[ORC_P_BASE_WEIGHT], sw = p[ORC_P_SPECULAR_WEIGHT];
  o.baseColor = v3(p + ORC_P_BASE_COLOR); o.baseWeight = bw; o.diffRough = p[ORC_P_DIFFUSE_ROUGHNESS]; o.thinWalled = p[ORC_P_THIN_WALLED] != 0.0f;
  o.albedo = v3(p + ORC_P_BASE_COLOR) * bw;
  o.specColor = v3(p + ORC_P_SPECULAR_COLOR);
  o.metalTint = o.specColor * sw;
  o.specWeight = sw;
  o.metalness = p[ORC_P_METALLIC];
  float r = p[ORC_P_ROUGHNESS], cr = p[ORC_P_CLEARCOAT_ROUGHNESS];
  o.coat = p[ORC_P_CLEARCOAT];
  r = opbr_effective_roughness(r, cr, o.coat);
  o.alpha = fmax2(r * r, 0.001f); o.coatAlpha = fmax2(cr * cr, 0.001f);
  opbr_anisotropy(o.alpha, p[ORC_P_SPECULAR_ANISOTROPY], o.alpha, o.alphaY); opbr_anisotropy(o.coatAlpha, p[ORC_P_COAT_ANISOTROPY], o.coatAlpha, o.coatAlphaY); // :133-136, 552-555
  // geometry_coat_tangent (:91, 561: the coat's dielectric_bsdf takes a tangent of its own).  Modelled in the form a document binds to it -- rotate3d of Tworld about
  // the normal, Standard Surface's coat_rotation: the frame's tangent turned by p[36] turns towards its bitangent.  Only an anisotropic coat can tell.
  // geometry_tangent (:89; the tangent of the dielectric and conductor lobes, :385, 402, 410, 449, 457) in the same form (Standard Surface's specular_rotation,
  // glTF's anisotropy_rotation): p[37] turns; opbr_enter turns the shading frame before the lobes run.  A coat without a frame of its own then starts from the turned
  // frame: its turn relative to it (coatRel) keeps it on the geometry tangent + p[36].
  o.specRot = p[ORC_P_SPECULAR_ANISOTROPY] > 0.0f && p[ORC_P_SPECULAR_ROTATION] != 0.0f;
  o.coatRot = p[ORC_P_CLEARCOAT] > 0.0f && p[ORC_P_COAT_ANISOTROPY] > 0.0f && (p[ORC_P_COAT_ROTATION] != 0.0f || o.specRot);
  o.coatRotC = 1.0f; o.coatRotS = 0.0f; o.specRotC = 1.0f; o.specRotS = 0.0f; o.coatRelC = 1.0f; o.coatRelS = 0.0f;
  if (o.coatRot) { const float a = 6.2831855f * p[ORC_P_COAT_ROTATION]; o.coatRotC = cosf(a); o.coatRotS = sinf(a); }
  if (o.specRot) {
    const float a = 6.2831855f * p[ORC_P_SPECULAR_ROTATION], r = 6.2831855f * (p[ORC_P_COAT_ROTATION] - p[ORC_P_SPECULAR_ROTATION]);
    o.specRotC = cosf(a); o.specRotS = sinf(a); o.coatRelC = cosf(r); o.coatRelS = sinf(r);
  }
  float cior = p[ORC_P_COAT_IOR]; float qc = (cior - 1.0f) / (cior + 1.0f); o.coatF0 = qc * qc;
  V3 cc = v3(p + ORC_P_COAT_COLOR); o.coatColor = cc; o.coatTint = v3(1, 1, 1) * (1.0f - o.coat) + cc * o.coat;
  // coat_substrate_attenuated = base_substrate * modulated_base_darkening * coat_attenuation (:538-552)
  o.coatTint = o.coatTint * opbr_base_darkening(o.baseColor, sw, p[ORC_P_METALLIC], o.coat, o.coatF0, cior, p[ORC_P_COAT_DARKENING]);
  // modulated_eta_s (open_pbr_surface.mtlx:306-366): specular_weight scales F0, eta follows
  float ior = p[ORC_P_IOR];
  float ratio = ior / cior, inv = cior / ior;
  float etaCoated = (ratio > 1.0f) ? ratio : inv;
  float etaS = etaCoated * o.coat + ior * (1.0f - o.coat);
  float q = (etaS - 1.0f) / (etaS + 1.0f);
  float f0 = fmin2(fmax2(sw * (q * q), 0.0f), 0.99999f);
  float eps = ((etaS - 1.0f) > 0.0f ? 1.0f : ((etaS - 1.0f) < 0.0f ? -1.0f : 0.0f)) * sqrtf(f0);
  o.eta = (1.0f + eps) / (1.0f - eps);
  o.tw = p[ORC_P_TRANSMISSION_WEIGHT];
  float depth = p[ORC_P_TRANSMISSION_DEPTH];
  V3 tc = v3(p + ORC_P_TRANSMISSION_COLOR);
  o.transTint = (depth > 0.0f) ? v3(1, 1, 1) : tc; // if_transmission_tint (:368-373)
  // dielectric base VDF (open_pbr_surface.mtlx:220-298): extinction = -ln(transmission_color)/depth, scattering =
  // transmission_scatter/depth, absorption = extinction - scattering, shifted to be non-negative
  V3 ext = (depth > 0.0f) ? v3(-logf(fmax2(tc.x, 1e-6f)) / depth, -logf(fmax2(tc.y, 1e-6f)) / depth, -logf(fmax2(tc.z, 1e-6f)) / depth) : v3(0, 0, 0);
  V3 sc = (depth > 0.0f) ? v3(p[ORC_P_TRANSMISSION_SCATTER] / depth, p[ORC_P_TRANSMISSION_SCATTER + 1] / depth, p[ORC_P_TRANSMISSION_SCATTER + 2] / depth) : v3(0, 0, 0);
  V3 ab = ext - sc;
  float mn = fmin2(fmin2(ab.x, ab.y), ab.z);
  if (0.0f > mn) ab = ab - v3(mn, mn, mn);
  o.sigmaA = (depth > 0.0f) ? ab : v3(0, 0, 0);
  o.sigmaS = sc;
  o.anisotropy = p[ORC_P_TRANSMISSION_SCATTER_ANISOTROPY];
  // thin-walled subsurface (open_pbr_surface.mtlx:140-196, 207-218): opaque_base = mix(diffuse_bsdf, subsurface_thin_walled, subsurface_weight); the
  // volumetric subsurface_bsdf of non-thin-walled materials is not modelled (weight treated as 0)
  o.ssWeight = o.thinWalled ? p[ORC_P_SUBSURFACE_WEIGHT] : 0.0f;
  o.ssColor = v3(p + ORC_P_SUBSURFACE_COLOR); o.ssAniso = p[ORC_P_SUBSURFACE_ANISOTROPY];
  // Volumetric subsurface (open_pbr_surface.mtlx:182-192, 207-218: subsurface_bsdf(color, radius = subsurface_radius * subsurface_radius_scale, anisotropy) for
  // materials that are NOT thin-walled), available when the render keeps a medium stack.  MaterialX maps the node onto MDL as a diffuse transmission at the
  // boundary over a scattering volume -- neither library is in the reference tree, so this is our statement of that mapping (unpinned): the subsurface share of the
  // opaque base enters the object by a cosine lobe on the far side (white tint) and pushes a medium with extinction 1 / radius per channel, single-scattering albedo
  // 1 - s^2 with s = 4.09712 + 4.20863 c - sqrt(9.59217 + 41.6808 c + 17.7126 c^2) (van de Hoek's inversion of the multiple-scattering colour c, Kulla & Conty 2017)
  // and Henyey-Greenstein g = subsurface_scatter_anisotropy; the walk and the exit are the medium stack's (rp_main.rgen:317-346, 462-477; the same layered BSDF
  // is met from inside, where the subsurface lobe transmits outwards and pops the medium).
  o.ssVolume = !o.thinWalled && sssVolume && p[ORC_P_SUBSURFACE_WEIGHT] > 0.0f;
  o.sssSigmaS = v3(0, 0, 0); o.sssSigmaT = v3(0, 0, 0);
  if (o.ssVolume) {
    o.ssWeight = fmin2(p[ORC_P_SUBSURFACE_WEIGHT], 1.0f);
    const float c3[3] = {fmax2(o.ssColor.x, 0.0f), fmax2(o.ssColor.y, 0.0f), fmax2(o.ssColor.z, 0.0f)};
    float sS[3], sT[3];
    for (int i = 0; i < 3; i++) {
      const float c = fmin2(c3[i], 1.0f);
      const float sq = sqrtf((9.59217f + 41.6808f * c) + (17.7126f * c) * c);
      const float sv = (4.09712f + 4.20863f * c) - sq;
      const float alb = fmin2(fmax2(1.0f - sv * sv, 0.0f), 1.0f);
      const float r = fmax2(p[ORC_P_SUBSURFACE_RADIUS] * p[ORC_P_SUBSURFACE_RADIUS_SCALE + i], 1e-6f);
      sT[i] = 1.0f / r; sS[i] = alb * sT[i];
    }
    o.sssSigmaS = v3(sS[0], sS[1], sS[2]); o.sssSigmaT = v3(sT[0], sT[1], sT[2]);
  }
  // fuzz layer (open_pbr_surface.mtlx:569-581): sheen_bsdf(fuzz_weight, fuzz_color, fuzz_roughness) on top of the coat
  o.fuzzWeight = fmin2(fmax2(p[ORC_P_FUZZ_WEIGHT], 0.0f), 1.0f); o.fuzzColor = v3(p + ORC_P_FUZZ_COLOR); o.fuzzAlpha = fmin2(fmax2(p[ORC_P_FUZZ_ROUGHNESS], 0.07f), 1.0f);
  // thin film (open_pbr_surface.mtlx:300-304, 404-431, 450-464): thin_film_thickness is in micrometres (:301-304 converts to nanometres)
  o.filmWeight = fmin2(fmax2(p[ORC_P_THIN_FILM_WEIGHT], 0.0f), 1.0f); o.filmNm = fmax2(p[ORC_P_THIN_FILM_THICKNESS], 0.0f) * 1000.0f; o.filmIor = fmax2(p[ORC_P_THIN_FILM_IOR], 1.0f);
  return o;
}

// ---- thin film: our closed form for dielectric_bsdf / generalized_schlick_bsdf(thinfilm_thickness, thinfilm_ior), which MaterialX maps onto MDL's df::thin_film
// (a spectral evaluation inside the MDL SDK, not in the reference tree).  Airy summation of the film's two interfaces for unpolarised light (Born & Wolf 7.6),
//   R = (r12^2 + r23^2 + 2 r12 r23 cos phi) / (1 + r12^2 r23^2 + 2 r12 r23 cos phi),   phi = 4 pi n_f d cos(theta_f) / lambda,   averaged over s and p,
// at three wavelengths standing for R, G, B (611.4, 548.4, 464.3 nm: the dominant wavelengths of the sRGB primaries) -- no spectral integration, so the fringes
// are more saturated than a spectral renderer's.  Indices are relative to the medium the ray comes from; a film of thickness 0 reproduces fresnel_dielectric.
// Metals: the substrate index per channel is the real index with the lobe's F0, n = (1 + sqrt F0) / (1 - sqrt F0).
// Where it enters: thin_film_weight mixes the film's reflectance into the Fresnel factor of the dielectric and metal lobes (mtlx :426-431, :461-464); the lobe
// selection probabilities keep the plain Fresnel term, the weights carry the colour ratio, and everything beneath the dielectric interface is weighted by
// (1 - F_mix) / (1 - F_plain) per channel, so a lobe and what lies under it still share the light.
inline float film_reflectance(float c, float nf, float n3, float d, float lam)
{
  const float s2 = 1.0f - c * c;
  const float s2f = s2 / (nf * nf), s23 = s2 / (n3 * n3);
  if (!(s2f < 1.0f) || !(s23 < 1.0f)) return 1.0f; // total internal reflection
  const float cf = sqrtf(1.0f - s2f), c3 = sqrtf(1.0f - s23);
  const float rs12 = (c - nf * cf) / (c + nf * cf), rp12 = (nf * c - cf) / (nf * c + cf);
  const float rs23 = (nf * cf - n3 * c3) / (nf * cf + n3 * c3), rp23 = (n3 * cf - nf * c3) / (n3 * cf + nf * c3);
  const float ph = ((2.0f * nf) * d * cf) / lam; // phase difference / (2 pi)
  float sn, cs; sincos2pi(ph - floorf(ph), &sn, &cs); (void)sn;
  const float ps = rs12 * rs23, pp = rp12 * rp23;
  const float Rs = ((rs12 * rs12 + rs23 * rs23) + (2.0f * ps) * cs) / ((1.0f + ps * ps) + (2.0f * ps) * cs);
  const float Rp = ((rp12 * rp12 + rp23 * rp23) + (2.0f * pp) * cs) / ((1.0f + pp * pp) + (2.0f * pp) * cs);
  return fmin2(fmax2(0.5f * (Rs + Rp), 0.0f), 1.0f);
}
inline V3 film_fresnel(float c, float nf, V3 n3, float d) { return v3(film_reflectance(c, nf, n3.x, d, 611.4f), film_reflectance(c, nf, n3.y, d, 548.4f), film_reflectance(c, nf, n3.z, d, 464.3f)); }
// Fresnel factor of the dielectric lobe with the film mixed in: eta = relative index of the interface (entering: the material's, leaving: its reciprocal)
inline V3 opbr_film_dielectric(const OpbrParams& o, float c, float eta, float Fplain)
{
  const float nf = (eta < 1.0f) ? o.filmIor * eta : o.filmIor;
  return v3(Fplain, Fplain, Fplain) * (1.0f - o.filmWeight) + film_fresnel(c, nf, v3(eta, eta, eta), o.filmNm) * o.filmWeight;
}
inline V3 opbr_film_metal(const OpbrParams& o, float c, V3 Fplain)
{
  const V3 f0 = v3(fmin2(fmax2(o.albedo.x, 0.0f), 0.98f), fmin2(fmax2(o.albedo.y, 0.0f), 0.98f), fmin2(fmax2(o.albedo.z, 0.0f), 0.98f));
  const V3 r = v3(sqrtf(f0.x), sqrtf(f0.y), sqrtf(f0.z));
  const V3 n3 = v3((1.0f + r.x) / (1.0f - r.x), (1.0f + r.y) / (1.0f - r.y), (1.0f + r.z) / (1.0f - r.z));
  return Fplain * (1.0f - o.filmWeight) + film_fresnel(c, o.filmIor, n3, o.filmNm) * o.filmWeight;
}

// ---- fuzz (sheen) lobe: our closed form for MaterialX sheen_bsdf / MDL df::sheen_bsdf, whose arithmetic is not in the reference tree.
// DEVIATION: the reference graph asks for sheen_bsdf mode="zeltner" (open_pbr_surface.mtlx:575) -- the LTC fit of Zeltner, Burley, Chiang 2022,
// whose tables live in MaterialX's library; this is the Conty-Kulla ("Charlie") lobe instead, i.e. MaterialX's mode="conty_kulla".  Same inputs, a different lobe shape.
// Micro-flake lobe D * V with the "Charlie" distribution D(h) = (2 + 1/a) sin(theta_h)^(1/a) / (2 pi) (Conty & Kulla 2017, a = fuzz_roughness in [0.07, 1]) and
// the Ashikhmin / Neubelt visibility V = 1 / (4 (n.l + n.v - n.l n.v)).  Its directional albedo E(n.v, a) has no closed form: both implementations embed the
// table of tools/gen_fuzz_albedo.py (numerical integration), interpolate it bilinearly and add 0.01 -- an upper bound of the true albedo for every a >= 0.07.
// Layering (MaterialX layer(top = sheen, base), :577-580): the fuzz is chosen with probability P = fuzz_weight * min(E, 1), everything beneath keeps 1 - P;
// where E exceeds 1 (grazing views of smooth fuzz) the lobe is scaled by 1 / E, so no direction reflects more than it receives.
static const float FUZZ_ALBEDO[16][17] = {
// rows: alpha = 1/16 .. 1, columns: mu = 0 .. 1 in steps of 1/16 (tools/gen_fuzz_albedo.py)
  {1.66603482e+00f, 1.05229735e+00f, 7.48142481e-01f, 5.42552471e-01f, 3.94544691e-01f, 2.85340458e-01f, 2.04062358e-01f, 1.43585801e-01f, 9.88851935e-02f, 6.62436262e-02f, 4.28260490e-02f, 2.64271554e-02f, 1.53114153e-02f, 8.10563751e-03f, 3.72325582e-03f, 1.30873080e-03f, 1.95315428e-04f},
  {1.22866476e+00f, 8.68625820e-01f, 6.77242339e-01f, 5.38716376e-01f, 4.31280196e-01f, 3.45276922e-01f, 2.75279015e-01f, 2.17816725e-01f, 1.70480222e-01f, 1.31495669e-01f, 9.95000154e-02f, 7.34108016e-02f, 5.23458906e-02f, 3.55714411e-02f, 2.24667117e-02f, 1.24994880e-02f, 5.20835957e-03f},
  {1.04291248e+00f, 7.75057077e-01f, 6.29255474e-01f, 5.20992339e-01f, 4.34587359e-01f, 3.63185972e-01f, 3.03000093e-01f, 2.51652122e-01f, 2.07522362e-01f, 1.69441670e-01f, 1.36528745e-01f, 1.08096354e-01f, 8.35938677e-02f, 6.25700131e-02f, 4.46479134e-02f, 2.95076892e-02f, 1.68739911e-02f},
  {9.36414123e-01f, 7.16623902e-01f, 5.95793307e-01f, 5.04945695e-01f, 4.31368500e-01f, 3.69544923e-01f, 3.16451848e-01f, 2.70210087e-01f, 2.29553193e-01f, 1.93577752e-01f, 1.61611512e-01f, 1.33137539e-01f, 1.07747734e-01f, 8.51129442e-02f, 6.49628416e-02f, 4.70720194e-02f, 3.12500633e-02f},
  {8.66332769e-01f, 6.76128209e-01f, 5.71131229e-01f, 4.91654038e-01f, 4.26739037e-01f, 3.71650100e-01f, 3.23803574e-01f, 2.81601369e-01f, 2.43972436e-01f, 2.10157394e-01f, 1.79594919e-01f, 1.51856542e-01f, 1.26606807e-01f, 1.03577562e-01f, 8.25508535e-02f, 6.33470416e-02f, 4.58163172e-02f},
  {8.16336453e-01f, 6.46200657e-01f, 5.52162349e-01f, 4.80711669e-01f, 4.22050655e-01f, 3.71954530e-01f, 3.28124613e-01f, 2.89142847e-01f, 2.54061490e-01f, 2.22210526e-01f, 1.93096176e-01f, 1.66342735e-01f, 1.41657025e-01f, 1.18805535e-01f, 9.75991860e-02f, 7.78828189e-02f, 5.95276207e-02f},
  {7.78707266e-01f, 6.23087585e-01f, 5.37092745e-01f, 4.71618980e-01f, 4.17691469e-01f, 3.71446818e-01f, 3.30786139e-01f, 2.94416696e-01f, 2.61475623e-01f, 2.31353760e-01f, 2.03602642e-01f, 1.77881300e-01f, 1.53923839e-01f, 1.31518483e-01f, 1.10493734e-01f, 9.07087103e-02f, 7.20463097e-02f},
  {7.49280632e-01f, 6.04652286e-01f, 5.24816334e-01f, 4.63969946e-01f, 4.13753271e-01f, 3.70571792e-01f, 3.32474768e-01f, 2.98261851e-01f, 2.67132372e-01f, 2.38521293e-01f, 2.12012753e-01f, 1.87290415e-01f, 1.64107352e-01f, 1.42266646e-01f, 1.21608362e-01f, 1.02000684e-01f, 8.33334997e-02f},
  {7.25595057e-01f, 5.89579582e-01f, 5.14612794e-01f, 4.57457095e-01f, 4.10229623e-01f, 3.69543940e-01f, 3.33563626e-01f, 3.01159322e-01f, 2.71578044e-01f, 2.44288355e-01f, 2.18898997e-01f, 1.95112124e-01f, 1.72694832e-01f, 1.51460946e-01f, 1.31258771e-01f, 1.11962639e-01f, 9.34669897e-02f},
  {7.06095338e-01f, 5.77011704e-01f, 5.05992115e-01f, 4.51849908e-01f, 4.07083064e-01f, 3.68470997e-01f, 3.34268302e-01f, 3.03401917e-01f, 2.75156647e-01f, 2.49027595e-01f, 2.24642813e-01f, 2.01718882e-01f, 1.80033773e-01f, 1.59409523e-01f, 1.39700606e-01f, 1.20785803e-01f, 1.02562815e-01f},
  {6.89747393e-01f, 5.66363335e-01f, 4.98608768e-01f, 4.46974337e-01f, 4.04269099e-01f, 3.67407918e-01f, 3.34719449e-01f, 3.05176616e-01f, 2.78094798e-01f, 2.52990693e-01f, 2.29507983e-01f, 2.07374871e-01f, 1.86378047e-01f, 1.66346103e-01f, 1.47138327e-01f, 1.28636986e-01f, 1.10742249e-01f},
  {6.75835013e-01f, 5.57220221e-01f, 4.92211729e-01f, 4.42697257e-01f, 4.01744992e-01f, 3.66382241e-01f, 3.34999412e-01f, 3.06607515e-01f, 2.80547380e-01f, 2.56353587e-01f, 2.33682737e-01f, 2.12272644e-01f, 1.91917211e-01f, 1.72450408e-01f, 1.53735459e-01f, 1.35657445e-01f, 1.18118644e-01f},
  {6.63845837e-01f, 5.49280763e-01f, 4.86614108e-01f, 4.38915670e-01f, 3.99472445e-01f, 3.65406990e-01f, 3.35161716e-01f, 3.07779849e-01f, 2.82623708e-01f, 2.59242892e-01f, 2.37304971e-01f, 2.16555879e-01f, 1.96795553e-01f, 1.77862525e-01f, 1.59623355e-01f, 1.41965449e-01f, 1.24793008e-01f},
  {6.53402805e-01f, 5.42319357e-01f, 4.81673747e-01f, 4.35548574e-01f, 3.97418410e-01f, 3.64487261e-01f, 3.35242122e-01f, 3.08753759e-01f, 2.84402877e-01f, 2.61752009e-01f, 2.40478083e-01f, 2.20333964e-01f, 2.01124832e-01f, 1.82693094e-01f, 1.64908230e-01f, 1.47659644e-01f, 1.30853310e-01f},
  {6.44222379e-01f, 5.36164165e-01f, 4.77280527e-01f, 4.32531655e-01f, 3.95554543e-01f, 3.63623768e-01f, 3.35264921e-01f, 3.09572637e-01f, 2.85943538e-01f, 2.63951302e-01f, 2.43281066e-01f, 2.23691702e-01f, 2.04992920e-01f, 1.87030569e-01f, 1.69676632e-01f, 1.52822360e-01f, 1.36375397e-01f},
  {6.36086643e-01f, 5.30681610e-01f, 4.73347783e-01f, 4.29813147e-01f, 3.93856794e-01f, 3.62814993e-01f, 3.35247070e-01f, 3.10268551e-01f, 2.87289977e-01f, 2.65894860e-01f, 2.45775416e-01f, 2.26695850e-01f, 2.08469898e-01f, 1.90946430e-01f, 1.73999682e-01f, 1.57522544e-01f, 1.41424328e-01f},
};
inline float fuzz_albedo(float mu, float alpha)
{
  const float x = fmin2(fmax2(mu, 0.0f), 1.0f) * 16.0f;
  int i = (int)x; if (i > 15) i = 15;
  const float fx = x - (float)i;
  const float y = alpha * 16.0f - 1.0f;
  int j = (int)y; if (j > 14) j = 14; if (j < 0) j = 0;
  const float fy = y - (float)j;
  const float a = FUZZ_ALBEDO[j][i] * (1.0f - fx) + FUZZ_ALBEDO[j][i + 1] * fx;
  const float b = FUZZ_ALBEDO[j + 1][i] * (1.0f - fx) + FUZZ_ALBEDO[j + 1][i + 1] * fx;
  return (a * (1.0f - fy) + b * fy) + 0.01f;
}
inline float fuzz_dv(V3 l1, V3 l2, float alpha) // D * V for local directions with l1.z, l2.z > 0
{
  const V3 h = normalize(l1 + l2);
  const float s2 = 1.0f - h.z * h.z;
  if (!(s2 > 0.0f)) return 0.0f;
  const float inv = 1.0f / alpha;
  const float D = ((2.0f + inv) * expf_poly((0.5f * inv) * logf_poly(s2))) / (2.0f * ORC_PI);
  return D / (4.0f * ((l2.z + l1.z) - l2.z * l1.z));
}

// subsurface_thin_walled = mix(reflection, transmission, 0.5) (:192-196) with
//   reflection   = oren_nayar_diffuse_bsdf(color = max(subsurface_color, 0), roughness = base_diffuse_roughness) * (subsurface_color * (1 - anisotropy))   (:141-160)
//   transmission = translucent_bsdf(color = max(subsurface_color, 0)) * (subsurface_color * (1 + anisotropy))                                              (:161-177)
// The two lobes are chosen with the mix weight 1/2, so that weight cancels in bsdf / pdf; these are the remaining colour factors.
inline V3 opbr_ss_color(const OpbrParams& o) { return v3(fmax2(o.ssColor.x, 0.0f), fmax2(o.ssColor.y, 0.0f), fmax2(o.ssColor.z, 0.0f)); }
inline V3 opbr_ss_reflect(const OpbrParams& o, V3 l1, V3 l2)
{
  const V3 c = opbr_ss_color(o);
  const V3 rho = (o.diffRough > 0.0f) ? eon_pi_f(c, o.diffRough, l1, l2) : c;
  return rho * (o.ssColor * (1.0f - o.ssAniso));
}
inline V3 opbr_ss_transmit(const OpbrParams& o) { return opbr_ss_color(o) * (o.ssColor * (1.0f + o.ssAniso)); }

static void opbr_sample_base(const OpbrParams& o, const State& st, V3 k1, const float xi[4], bool frontFace, BsdfSample& out) // everything beneath the fuzz
{
  V3 l1 = to_local(st, k1);
  float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
  float z = xi[2];
  // the coat lobe lives in its own frame when geometry_coat_normal is mapped (:560); its Fresnel term -- lobe probability and what it leaves for the base -- follows
  V3 l1c = l1; float nk1c = nk1;
  if (st.hasCoatFrame) { l1c = to_local_coat(st, k1); nk1c = fmax2(l1c.z, 1e-4f); l1c.z = nk1c; }
  if (o.coatRot) l1c = coat_turn_local(o.coatRotC, o.coatRotS, l1c);
  float Fc = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(nk1c));
  if (z < Fc) { // coat reflection
    GgxOut g = ggx_sample2(l1c, o.coatAlpha, o.coatAlphaY, xi[0], xi[1]);
    V3 k2 = to_world_coat(st, o.coatRot ? coat_turn_world(o.coatRotC, o.coatRotS, g.l2) : g.l2);
    if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
    float Fh = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(g.kh));
    float w = (Fh / Fc) * g.g2OverG1;
    out.k2 = k2; out.pdf = Fc * g.pdf; out.overPdf = v3(w, w, w); out.event = EV_GLOSSY | EV_REFLECTION;
    return;
  }
  z = (z - Fc) / (1.0f - Fc);
  if (z < o.metalness) { // metal
    GgxOut g = ggx_sample2(l1, o.alpha, o.alphaY, xi[0], xi[1]);
    V3 k2 = to_world(st, g.l2);
    if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
    V3 F = schlick_f82(o.albedo, o.metalTint, g.kh) * o.specWeight;
    if (o.filmWeight > 0.0f) F = opbr_film_metal(o, g.kh, schlick_f82(o.albedo, o.metalTint, g.kh)) * o.specWeight;
    out.k2 = k2; out.pdf = (1.0f - Fc) * o.metalness * g.pdf; out.overPdf = (F * o.coatTint) * g.g2OverG1; out.event = EV_GLOSSY | EV_REFLECTION;
    return;
  }
  z = (z - o.metalness) / (1.0f - o.metalness);
  float eta = relative_eta(st, o.eta); // ior2 / ior1 (rp_main.chit:188-189); empty stack: eta entering, 1/eta leaving
  (void)frontFace;
  float Fd = fresnel_dielectric(nk1, eta);
  if (z < Fd) { // dielectric reflection
    GgxOut g = ggx_sample2(l1, o.alpha, o.alphaY, xi[0], xi[1]);
    V3 k2 = to_world(st, g.l2);
    if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
    float Fh = fresnel_dielectric(g.kh, eta);
    out.k2 = k2; out.pdf = (1.0f - Fc) * (1.0f - o.metalness) * Fd * g.pdf;
    out.overPdf = (o.specColor * o.coatTint) * ((Fh / Fd) * g.g2OverG1); out.event = EV_GLOSSY | EV_REFLECTION;
    if (o.filmWeight > 0.0f) out.overPdf = (o.specColor * o.coatTint) * (opbr_film_dielectric(o, g.kh, eta, Fh) * (g.g2OverG1 / Fd));
    return;
  }
  // under the film everything beneath the interface is weighted by (1 - F_mix) / (1 - F_plain) per channel
  const V3 under = (o.filmWeight > 0.0f) ? (v3(1, 1, 1) - opbr_film_dielectric(o, nk1, eta, Fd)) * (1.0f / (1.0f - Fd)) : v3(1, 1, 1);
  z = (z - Fd) / (1.0f - Fd);
  if (z < o.tw) { // rough refraction through a VNDF-sampled micro-normal
    GgxOut g = ggx_sample2(l1, o.alpha, o.alphaY, xi[0], xi[1]); // provides the half vector via l2 = reflect(l1, h)
    V3 h = normalize(l1 + g.l2);
    float kh = dot(l1, h);
    if (!g.valid || !(kh > 0.0f)) return;
    float Fh = fresnel_dielectric(kh, eta);
    float sin2t = (1.0f - kh * kh) / (eta * eta);
    if (!(sin2t < 1.0f)) return; // total internal reflection at this micro-normal: absorbed
    float ct = sqrtf(1.0f - sin2t);
    V3 lt = h * (kh / eta - ct) - l1 * (1.0f / eta);
    // thin-walled (MDL: "transmission does not refract"): the micro-facet reflection direction, mirrored through the surface
    if (o.thinWalled) lt = v3(g.l2.x, g.l2.y, -g.l2.z);
    V3 k2 = to_world(st, lt);
    if (!(lt.z < 0.0f) || !(dot(k2, st.geomNormal) < 0.0f)) return;
    float a2 = o.alpha * o.alpha, nk2 = -lt.z;
    float L1 = ggx_lambda_term(a2, nk1), L2 = ggx_lambda_term(a2, nk2);
    if (o.alpha != o.alphaY) { L1 = ggx_lambda_xy(o.alpha, o.alphaY, l1); L2 = ggx_lambda_xy(o.alpha, o.alphaY, lt); }
    float G1 = 2.0f * nk1 / (nk1 + L1), G2 = 2.0f * nk1 * nk2 / (nk2 * L1 + nk1 * L2);
    float w = ((1.0f - Fh) / (1.0f - Fd)) * (G2 / G1);
    out.k2 = normalize(k2); out.pdf = (1.0f - Fc) * (1.0f - o.metalness) * (1.0f - Fd) * o.tw * g.pdf;
    out.overPdf = (o.transTint * o.coatTint) * w; out.event = EV_GLOSSY | EV_TRANSMISSION;
    if (o.filmWeight > 0.0f) out.overPdf = (o.transTint * o.coatTint) * ((v3(1, 1, 1) - opbr_film_dielectric(o, kh, eta, Fh)) * ((G2 / G1) / (1.0f - Fd)));
    return;
  }
  const float pBase = (1.0f - Fc) * (1.0f - o.metalness) * (1.0f - Fd) * (1.0f - o.tw);
  V3 l = sample_hemisphere(xi[0], xi[1]); // cosine-weighted, for every lobe of the opaque base
  if (o.ssWeight > 0.0f) { // thin-walled subsurface takes subsurface_weight of the opaque base, half of it reflected, half transmitted
    z = (z - o.tw) / (1.0f - o.tw);
    if (z < o.ssWeight) {
      const bool through = !((z / o.ssWeight) < 0.5f);
      if (!(l.z > 0.0f)) return;
      if (o.ssVolume) { // volumetric form: the whole subsurface share enters (or, met from inside, leaves) by a cosine lobe on the far side, untinted
        V3 k2 = to_world(st, v3(l.x, l.y, -l.z));
        if (!(dot(k2, st.geomNormal) < 0.0f)) return;
        out.k2 = k2; out.pdf = pBase * o.ssWeight * (l.z / ORC_PI);
        out.overPdf = o.coatTint; out.event = EV_DIFFUSE | EV_TRANSMISSION | EV_SUBSURFACE;
        if (o.filmWeight > 0.0f) out.overPdf = out.overPdf * under;
        return;
      }
      if (through) { // translucent_bsdf: Lambert on the far side
        V3 k2 = to_world(st, v3(l.x, l.y, -l.z));
        if (!(dot(k2, st.geomNormal) < 0.0f)) return;
        out.k2 = k2; out.pdf = pBase * o.ssWeight * 0.5f * (l.z / ORC_PI);
        out.overPdf = opbr_ss_transmit(o) * o.coatTint; out.event = EV_DIFFUSE | EV_TRANSMISSION;
        if (o.filmWeight > 0.0f) out.overPdf = out.overPdf * under;
        return;
      }
      V3 k2 = to_world(st, l);
      if (!(dot(k2, st.geomNormal) > 0.0f)) return;
      out.k2 = k2; out.pdf = pBase * o.ssWeight * 0.5f * (l.z / ORC_PI);
      out.overPdf = opbr_ss_reflect(o, l1, l) * o.coatTint; out.event = EV_DIFFUSE | EV_REFLECTION;
      if (o.filmWeight > 0.0f) out.overPdf = out.overPdf * under;
      return;
    }
  }
  V3 k2 = to_world(st, l); // opaque base: Lambert, or energy-preserving Oren-Nayar when base_diffuse_roughness > 0
  if (!(l.z > 0.0f) || !(dot(k2, st.geomNormal) > 0.0f)) return;
  out.k2 = k2; out.pdf = pBase * (1.0f - o.ssWeight) * (l.z / ORC_PI);
  V3 rho = (o.diffRough > 0.0f) ? eon_pi_f(o.baseColor, o.diffRough, l1, l) * o.baseWeight : o.albedo;
  out.overPdf = rho * o.coatTint; out.event = EV_DIFFUSE | EV_REFLECTION;
  if (o.filmWeight > 0.0f) out.overPdf = out.overPdf * under;
}

// The parameters of a hit and the frame its lobes run in: geometry_tangent turns the frame's tangents (every lobe beneath the coat sees the turned frame -- the
// diffuse ones depend on the normal only), and the coat takes its relative turn unless geometry_coat_normal gave it a frame of its own (made from the unturned tangent).
static OpbrParams opbr_enter(const OrcMaterial& m, const State& stIn, State& st)
{
  OpbrParams o = opbr_params(m, stIn.sssVolume);
  st = stIn;
  if (o.specRot) {
    st.tangentU = stIn.tangentU * o.specRotC + stIn.tangentV * o.specRotS; st.tangentV = stIn.tangentV * o.specRotC - stIn.tangentU * o.specRotS;
    if (!stIn.hasCoatFrame) { o.coatRotC = o.coatRelC; o.coatRotS = o.coatRelS; }
  }
  return o;
}

void opbr_sample(const OrcMaterial& m, const State& stIn, V3 k1, const float xi[4], bool frontFace, BsdfSample& out)
{
  State st; const OpbrParams o = opbr_enter(m, stIn, st);
  if (!(o.fuzzWeight > 0.0f)) { opbr_sample_base(o, st, k1, xi, frontFace, out); return; }
  V3 l1 = to_local(st, k1);
  const float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
  const float Ef = fuzz_albedo(nk1, o.fuzzAlpha), Pf = o.fuzzWeight * fmin2(Ef, 1.0f);
  if (xi[2] < Pf) { // the fuzz lobe, cosine-sampled
    const V3 l = sample_hemisphere(xi[0], xi[1]);
    const V3 k2 = to_world(st, l);
    if (!(l.z > 0.0f) || !(dot(k2, st.geomNormal) > 0.0f)) return;
    out.k2 = k2; out.pdf = Pf * (l.z / ORC_PI);
    out.overPdf = o.fuzzColor * ((fuzz_dv(l1, l, o.fuzzAlpha) * ORC_PI) / Ef);
    out.event = EV_GLOSSY | EV_REFLECTION;
    return;
  }
  const float xb[4] = {xi[0], xi[1], (xi[2] - Pf) / (1.0f - Pf), xi[3]};
  opbr_sample_base(o, st, k1, xb, frontFace, out);
  out.pdf = out.pdf * (1.0f - Pf); // bsdf / pdf is unchanged: the layers beneath are weighted by the same 1 - P they are chosen with
}

static void opbr_evaluate_base(const OpbrParams& o, const State& st, V3 k1, V3 k2, bool frontFace, BsdfEval& out)
{
  V3 l1 = to_local(st, k1), l2 = to_local(st, k2);
  float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
  float eta = relative_eta(st, o.eta); (void)frontFace;
  V3 l1c = l1, l2c = l2; float nk1c = nk1; // the coat lobe's own frame (geometry_coat_normal)
  if (st.hasCoatFrame) { l1c = to_local_coat(st, k1); nk1c = fmax2(l1c.z, 1e-4f); l1c.z = nk1c; l2c = to_local_coat(st, k2); }
  if (o.coatRot) { l1c = coat_turn_local(o.coatRotC, o.coatRotS, l1c); l2c = coat_turn_local(o.coatRotC, o.coatRotS, l2c); }
  float Fc = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(nk1c));
  float Fd = fresnel_dielectric(nk1, eta);
  float fc, pc, khc; ggx_eval2(l1c, l2c, o.coatAlpha, o.coatAlphaY, fc, pc, khc);
  float fs, ps, khs; ggx_eval2(l1, l2, o.alpha, o.alphaY, fs, ps, khs);
  float Fch = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(khc));
  V3 Fm = schlick_f82(o.albedo, o.metalTint, khs) * o.specWeight;
  float Fdh = fresnel_dielectric(khs, eta);
  float cd = l2.z / ORC_PI;
  float base = 1.0f - Fc, diel = 1.0f - o.metalness;
  V3 gl = v3(Fch * fc, Fch * fc, Fch * fc);
  gl = gl + ((Fm * o.coatTint) * fs) * (base * o.metalness);
  gl = gl + ((o.specColor * o.coatTint) * (Fdh * fs)) * (base * diel);
  V3 under = v3(1, 1, 1);
  if (o.filmWeight > 0.0f) { // thin film: colour Fresnel factors in the two glossy lobes, (1 - F_mix) / (1 - F_plain) on what lies beneath the interface
    const V3 FmF = opbr_film_metal(o, khs, schlick_f82(o.albedo, o.metalTint, khs)) * o.specWeight;
    gl = v3(Fch * fc, Fch * fc, Fch * fc);
    gl = gl + ((FmF * o.coatTint) * fs) * (base * o.metalness);
    gl = gl + ((o.specColor * o.coatTint) * (opbr_film_dielectric(o, khs, eta, Fdh) * fs)) * (base * diel);
    // total internal reflection (Fd == 1: back face beyond the critical angle): nothing lies beneath the interface -- without the guard 0 * (1 / 0) = NaN
    under = (Fd < 1.0f) ? (v3(1, 1, 1) - opbr_film_dielectric(o, nk1, eta, Fd)) * (1.0f / (1.0f - Fd)) : v3(0, 0, 0);
  }
  out.glossy = gl;
  V3 rho = (o.diffRough > 0.0f && l2.z > 0.0f) ? eon_pi_f(o.baseColor, o.diffRough, l1, l2) * o.baseWeight : o.albedo;
  const float wBase = cd * base * diel * (1.0f - Fd) * (1.0f - o.tw);
  if (o.ssVolume) { // volumetric subsurface: its share of the opaque base transmits (not reached by NEE); the diffuse lobe keeps 1 - subsurface_weight
    out.diffuse = ((rho * (1.0f - o.ssWeight)) * o.coatTint) * wBase;
    if (o.filmWeight > 0.0f) out.diffuse = out.diffuse * under;
    out.pdf = Fc * pc + base * (o.metalness * ps + diel * (Fd * ps + (1.0f - Fd) * (1.0f - o.tw) * (1.0f - o.ssWeight) * cd));
    return;
  }
  if (o.ssWeight > 0.0f) { // reflection side of the thin-walled subsurface mix (the transmitted half lies below the surface: not reached by NEE)
    const V3 ss = (l2.z > 0.0f) ? opbr_ss_reflect(o, l1, l2) : v3(0, 0, 0);
    out.diffuse = ((rho * (1.0f - o.ssWeight) + ss * (o.ssWeight * 0.5f)) * o.coatTint) * wBase;
    if (o.filmWeight > 0.0f) out.diffuse = out.diffuse * under;
    out.pdf = Fc * pc + base * (o.metalness * ps + diel * (Fd * ps + (1.0f - Fd) * (1.0f - o.tw) * ((1.0f - o.ssWeight) + o.ssWeight * 0.5f) * cd));
    return;
  }
  out.diffuse = (rho * o.coatTint) * wBase;
  if (o.filmWeight > 0.0f) out.diffuse = out.diffuse * under;
  out.pdf = Fc * pc + base * (o.metalness * ps + diel * (Fd * ps + (1.0f - Fd) * (1.0f - o.tw) * cd));
}

void opbr_evaluate(const OrcMaterial& m, const State& stIn, V3 k1, V3 k2, bool frontFace, BsdfEval& out)
{
  State st; const OpbrParams o = opbr_enter(m, stIn, st);
  opbr_evaluate_base(o, st, k1, k2, frontFace, out);
  if (!(o.fuzzWeight > 0.0f)) return;
  V3 l1 = to_local(st, k1); const V3 l2 = to_local(st, k2);
  const float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
  const float Ef = fuzz_albedo(nk1, o.fuzzAlpha), Eb = fmin2(Ef, 1.0f), Pf = o.fuzzWeight * Eb, keep = 1.0f - Pf;
  const float cd = fmax2(l2.z, 0.0f) / ORC_PI;
  const V3 sheen = (l2.z > 0.0f) ? o.fuzzColor * (((o.fuzzWeight * (Eb / Ef)) * fuzz_dv(l1, l2, o.fuzzAlpha)) * l2.z) : v3(0, 0, 0);
  out.glossy = out.glossy * keep + sheen;
  out.diffuse = out.diffuse * keep;
  out.pdf = Pf * cd + keep * out.pdf;
}

void bsdf_sample(const OrcMaterial& m, const State& st, V3 k1, const float xi[4], BsdfSample& out)
{
  out.event = EV_ABSORB; out.pdf = 0.0f; out.overPdf = v3(0, 0, 0); out.k2 = v3(0, 0, 0);
  if (m.klass == ORC_MAT_DIFFUSE) {
    V3 l = sample_hemisphere(xi[0], xi[1]);
    V3 k2 = to_world(st, l);
    if (!(l.z > 0.0f) || !(dot(k2, st.geomNormal) > 0.0f)) return;
    out.k2 = k2; out.pdf = l.z / ORC_PI; out.overPdf = v3(m.p + ORC_P_BASE_COLOR); out.event = EV_DIFFUSE | EV_REFLECTION;
    return;
  }
  if (m.klass == ORC_MAT_USD_PREVIEW_SURFACE) {
    UpsParams u = ups_params(m);
    V3 l1 = to_local(st, k1);
    float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
    float z = xi[2];
    float Fc = u.coat * (0.04f + 0.96f * schlick_w(nk1));
    if (z < Fc) {
      GgxOut g = ggx_sample(l1, u.coatAlpha, xi[0], xi[1]);
      V3 k2 = to_world(st, g.l2);
      if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
      float Fh = u.coat * (0.04f + 0.96f * schlick_w(g.kh));
      float w = (Fh / Fc) * g.g2OverG1;
      out.k2 = k2; out.pdf = Fc * g.pdf; out.overPdf = v3(w, w, w); out.event = EV_GLOSSY | EV_REFLECTION;
      return;
    }
    z = (z - Fc) / (1.0f - Fc);
    V3 Fs = schlick3(u.F0, nk1);
    float ps = fmax2(Fs.x, fmax2(Fs.y, Fs.z));
    if (z < ps) {
      GgxOut g = ggx_sample(l1, u.alpha, xi[0], xi[1]);
      V3 k2 = to_world(st, g.l2);
      if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
      V3 Fh = schlick3(u.F0, g.kh);
      out.k2 = k2; out.pdf = (1.0f - Fc) * ps * g.pdf; out.overPdf = Fh * (g.g2OverG1 / ps); out.event = EV_GLOSSY | EV_REFLECTION;
      return;
    }
    V3 l = sample_hemisphere(xi[0], xi[1]);
    V3 k2 = to_world(st, l);
    if (!(l.z > 0.0f) || !(dot(k2, st.geomNormal) > 0.0f)) return;
    out.k2 = k2; out.pdf = (1.0f - Fc) * (1.0f - ps) * (l.z / ORC_PI);
    out.overPdf = (u.albedo * (v3(1, 1, 1) - Fs)) * (1.0f / (1.0f - ps));
    out.event = EV_DIFFUSE | EV_REFLECTION;
    return;
  }
  if (m.klass == ORC_MAT_OPEN_PBR) { opbr_sample(m, st, k1, xi, st.frontFace, out); return; }
}

void bsdf_evaluate(const OrcMaterial& m, const State& st, V3 k1, V3 k2, BsdfEval& out)
{
  out.diffuse = v3(0, 0, 0); out.glossy = v3(0, 0, 0); out.pdf = 0.0f;
  float nk2 = dot(st.normal, k2);
  if (!(nk2 > 0.0f)) return;
  if (m.klass == ORC_MAT_DIFFUSE) {
    float c = nk2 / ORC_PI;
    out.diffuse = v3(m.p + ORC_P_BASE_COLOR) * c; out.pdf = c;
    return;
  }
  if (m.klass == ORC_MAT_USD_PREVIEW_SURFACE) {
    UpsParams u = ups_params(m);
    V3 l1 = to_local(st, k1), l2 = to_local(st, k2);
    float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
    float Fc = u.coat * (0.04f + 0.96f * schlick_w(nk1));
    V3 Fs = schlick3(u.F0, nk1);
    float ps = fmax2(Fs.x, fmax2(Fs.y, Fs.z));
    float fc, pc, khc; ggx_eval(l1, l2, u.coatAlpha, fc, pc, khc);
    float fs, pss, khs; ggx_eval(l1, l2, u.alpha, fs, pss, khs);
    float Fch = u.coat * (0.04f + 0.96f * schlick_w(khc));
    V3 Fsh = schlick3(u.F0, khs);
    float cd = l2.z / ORC_PI;
    out.glossy = v3(Fch * fc, Fch * fc, Fch * fc) + (Fsh * fs) * (1.0f - Fc);
    out.diffuse = (u.albedo * (v3(1, 1, 1) - Fs)) * (cd * (1.0f - Fc));
    out.pdf = Fc * pc + (1.0f - Fc) * (ps * pss + (1.0f - ps) * cd);
    return;
  }
  if (m.klass == ORC_MAT_OPEN_PBR) { opbr_evaluate(m, st, k1, k2, st.frontFace, out); return; }
}

// mdl_bsdf_scattering_auxiliary stand-in (rp_main.chit:263-289): albedo_diffuse + albedo_glossy of the closed forms
V3 bsdf_albedo(const OrcMaterial& m, const State& st, V3 k1)
{
  float nk1 = fmax2(dot(st.normal, k1), 1e-4f);
  if (m.klass == ORC_MAT_DIFFUSE) return v3(m.p + ORC_P_BASE_COLOR);
  if (m.klass == ORC_MAT_USD_PREVIEW_SURFACE) {
    UpsParams u = ups_params(m);
    float Fc = u.coat * (0.04f + 0.96f * schlick_w(nk1));
    V3 Fs = schlick3(u.F0, nk1);
    V3 diffuse = (u.albedo * (v3(1, 1, 1) - Fs)) * (1.0f - Fc);
    V3 glossy = v3(Fc, Fc, Fc) + Fs * (1.0f - Fc);
    return diffuse + glossy;
  }
  OpbrParams o = opbr_params(m);
  float eta = relative_eta(st, o.eta);
  const float nk1c = st.hasCoatFrame ? fmax2(dot(st.coatNormal, k1), 1e-4f) : nk1; // the coat's Fresnel term in its own frame (geometry_coat_normal)
  float Fc = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(nk1c));
  float Fd = fresnel_dielectric(nk1, eta);
  float base = 1.0f - Fc, diel = 1.0f - o.metalness;
  V3 diffuse = (o.albedo * o.coatTint) * (base * diel * (1.0f - Fd) * (1.0f - o.tw));
  V3 glossy = v3(Fc, Fc, Fc) + ((schlick_f82(o.albedo, o.metalTint, nk1) * o.specWeight) * o.coatTint) * (base * o.metalness)
              + (o.specColor * o.coatTint) * (base * diel * Fd);
  if (o.filmWeight > 0.0f) { // thin film: the two Fresnel factors carry the film's reflectance, what lies beneath the interface its complement
    const V3 Fdf = opbr_film_dielectric(o, nk1, eta, Fd);
    diffuse = ((o.albedo * o.coatTint) * (v3(1, 1, 1) - Fdf)) * (base * diel * (1.0f - o.tw));
    glossy = v3(Fc, Fc, Fc) + ((opbr_film_metal(o, nk1, schlick_f82(o.albedo, o.metalTint, nk1)) * o.specWeight) * o.coatTint) * (base * o.metalness)
             + ((o.specColor * o.coatTint) * Fdf) * (base * diel);
  }
  if (o.fuzzWeight > 0.0f) { // the fuzz layer keeps P = fuzz_weight * min(E, 1) of the light (tinted), what is beneath gets 1 - P
    const float Pf = o.fuzzWeight * fmin2(fuzz_albedo(nk1, o.fuzzAlpha), 1.0f);
    return (diffuse + glossy) * (1.0f - Pf) + o.fuzzColor * Pf;
  }
  return diffuse + glossy;
}

// ---------------------------------------------------------------------------------------------
// Light sampling (rp_main.chit:30-129)
// ---------------------------------------------------------------------------------------------
struct Uniforms {
  uint32_t sphereCount, distantCount, rectCount, diskCount, total;
  float lightIntensityMultiplier, sensorExposureScale;
};

void sample_light(const Prepared& P, const Uniforms& ubo, const float k4[4], V3 surfacePos,
                  V3& dirToLight, float& dist, V3& power, float& invPdf, uint32_t& dsPacked)
{
  float sel = k4[0] * (float)ubo.total;
  if (sel <= (float)ubo.sphereCount) { // :32-53
    uint32_t idx = (uint32_t)(k4[1] * (float)ubo.sphereCount);
    uint32_t last = ubo.sphereCount - 1u; if (idx > last) idx = last;
    SphereL l = idx < P.sphere.size() ? P.sphere[idx] : SphereL{v3(0, 0, 0), 0, v3(0, 0, 0), 0.0f, v3(0, 0, 0)}; // zero-filled store (SyncBuffer.cpp:90)
    V3 samplePos = l.pos + sample_sphere(k4[2], k4[3], l.radius);
    V3 dir = samplePos - surfacePos;
    dist = length(dir);
    dirToLight = safe_div(dir, dist);
    V3 ln = normalize(samplePos - l.pos);
    float cosTheta = fmax2(0.0f, dot(-dirToLight, ln));
    invPdf = safe_div((l.area > 0.0f) ? (l.area * cosTheta) : 1.0f, dist * dist);
    power = l.em * ubo.lightIntensityMultiplier; dsPacked = l.ds;
  } else if (sel <= (float)(ubo.sphereCount + ubo.distantCount)) { // :54-77
    uint32_t idx = (uint32_t)(k4[1] * (float)ubo.distantCount);
    uint32_t last = ubo.distantCount - 1u; if (idx > last) idx = last;
    const DistantL& l = P.distant[idx];
    dist = 100000.0f; dirToLight = -l.dir;
    power = l.em * ubo.lightIntensityMultiplier; invPdf = l.invPdf; dsPacked = l.ds;
    if (l.angle > 0.0f) {
      V3 t1, t2; orthonormal_basis(dirToLight, t1, t2);
      float phi = (k4[2] * 2.0f * ORC_PI) - ORC_PI;
      float theta = k4[3] * l.angle;
      float sp, cp, stt, ct; sincosr(phi, &sp, &cp); sincosr(theta, &stt, &ct);
      dirToLight = normalize((t1 * cp + t2 * sp) * stt + dirToLight * ct);
    }
  } else if (sel <= (float)(ubo.sphereCount + ubo.distantCount + ubo.rectCount)) { // :78-103
    uint32_t idx = (uint32_t)(k4[1] * (float)ubo.rectCount);
    uint32_t last = ubo.rectCount - 1u; if (idx > last) idx = last;
    const RectL& l = P.rect[idx];
    float sx = (k4[2] - 0.5f) * l.width, sy = (k4[3] - 0.5f) * l.height;
    V3 t0 = decode_direction(l.t0), t1 = decode_direction(l.t1);
    V3 samplePos = (l.origin + t0 * sx) + t1 * sy;
    V3 dir = samplePos - surfacePos;
    dist = length(dir); dirToLight = safe_div(dir, dist);
    V3 ln = cross(t1, t0);
    float cosTheta = fmax2(0.0f, dot(-dirToLight, ln));
    float area = l.width * l.height;
    invPdf = safe_div((area > 0.0f) ? (area * cosTheta) : 1.0f, dist * dist);
    power = l.em * ubo.lightIntensityMultiplier; dsPacked = l.ds;
  } else { // :104-127
    uint32_t idx = (uint32_t)(k4[1] * (float)ubo.diskCount);
    uint32_t last = ubo.diskCount - 1u; if (idx > last) idx = last;
    const DiskL& l = P.disk[idx];
    float sx, sy; sample_disk(k4[2], k4[3], l.rx, l.ry, sx, sy);
    V3 t0 = decode_direction(l.t0), t1 = decode_direction(l.t1);
    V3 samplePos = (l.origin + t0 * sx) + t1 * sy;
    V3 dir = samplePos - surfacePos;
    dist = length(dir); dirToLight = safe_div(dir, dist);
    V3 ln = cross(t1, t0);
    float cosTheta = fmax2(0.0f, dot(-dirToLight, ln));
    float area = l.rx * l.ry * ORC_PI;
    invPdf = safe_div((area > 0.0f) ? (area * cosTheta) : 1.0f, dist * dist);
    power = l.em * ubo.lightIntensityMultiplier; dsPacked = l.ds;
  }
  power = power * ubo.sensorExposureScale; // :126 (exp2(sensorExposure))
  invPdf = invPdf * (float)ubo.total;      // :127
}

// ---------------------------------------------------------------------------------------------
// The per-pixel megakernel (rp_main.rgen:185-521 with rp_main.chit/.miss inlined)
// ---------------------------------------------------------------------------------------------
struct Medium { V3 ior, sigma_s, sigma_t; float bias; }; // rp_main_payload.glsl:11-17
const uint32_t MAX_MEDIUM_STACK = 15; // the medium index field has four bits (rp_main_payload.glsl:4-5)
struct Payload { // rp_main_payload.glsl:20-48
  V3 throughput; uint32_t bitfield; V3 radiance; uint32_t rng; V3 origin, dir, neeToLight, neeContrib;
  Medium media[MAX_MEDIUM_STACK]; V3 walkSegmentPdf; float tMaxLast; // MEDIUM_STACK_SIZE > 0 (tMaxLast = gl_RayTmaxEXT of the segment)
};
const uint32_t BOUNCES_MASK = 0x00000fffu, TERMINATE_FLAG = 0x80000000u, MEDIUM_MASK = 0x0f000000u; // rp_main_payload.glsl:3-9
const uint32_t WALK_MISS_FLAG = 0x40000000u, WALK_MASK = 0x00fff000u, WALK_OFFSET = 12;
// shadeRayPayloadGetMediumIdx (rp_main_payload.glsl:76-90): the stored index may exceed the stack size, reads clamp it
inline uint32_t payload_medium_idx(uint32_t bitfield, uint32_t stackSize)
{
  uint32_t idx = (bitfield & MEDIUM_MASK) >> 24, mx = stackSize > 1u ? stackSize : 1u;
  return idx < mx ? idx : mx;
}
// shadeRayPayloadIncrementWalk (rp_main_payload.glsl:60-69), restated literally: the "+ 1" is applied to the MASKED, unshifted
// field, so it lands in bit 0 -- outside the walk mask -- and is OR-ed into the bounce counter; the walk length itself never
// grows (so maxVolumeWalkLength never triggers).  Kept bug-compatible, like the FaceId fetch.
inline void payload_increment_walk(uint32_t& bitfield)
{
  uint32_t b = bitfield & WALK_MASK;
  b = (b + 1u) < WALK_MASK ? (b + 1u) : WALK_MASK;
  bitfield &= ~WALK_MASK;
  bitfield |= b;
}

struct Frame {
  const Prepared* P; const OrcCamera* cam; const OrcSettings* rs; Uniforms ubo;
  V3 camPos, camFwd, camUp, camRight, L; float WX, HY; float lensRadius; V3 background; float clipNear, clipFar;
  uint32_t width, height;
};

// rp_main.chit:132-493
void closest_hit(const Frame& F, const Hit& h, Payload& pl, float hitT)
{
  const Prepared& P = *F.P;
  State st; const MeshData* mesh;
  V3 rayDir = pl.dir;
  setup_shading_state(P, h, rayDir, st, mesh);
  const OrcMaterial& baseMat = P.materials[mesh->material];
  OrcMaterial resolved;
  st.cameraPosition = v3(F.cam->position); st.frame = F.rs->frame;
  if (material_textured(baseMat)) resolved = resolve_material(P, baseMat, st, rayDir);
  const OrcMaterial& mat = material_textured(baseMat) ? resolved : baseMat;
  bool isLeftHanded = (mesh->flags & 1u) != 0, isDoubleSided = (mesh->flags & 2u) != 0;
  V3 throughput = pl.throughput, radiance = pl.radiance;
  (void)isLeftHanded;
  // 3. volume attenuation with an empty medium stack (:160-186): inside (1-bit toggle) -> Beer-Lambert with the HIT material's
  // absorption coefficient (:169-173)
  const uint32_t stackSize = F.rs->mediumStackSize < MAX_MEDIUM_STACK ? F.rs->mediumStackSize : MAX_MEDIUM_STACK;
  uint32_t mediumIdx = payload_medium_idx(pl.bitfield, stackSize);
  float prevMediumIor = 1.0f, nextMediumIor = 1.0f;
  if (mediumIdx > 0) {
    float distance = hitT * F.rs->metersPerSceneUnit;
    if (stackSize == 0) { // :169-173: the HIT material's absorption coefficient
      if (mat.klass == ORC_MAT_OPEN_PBR) {
        OpbrParams o = opbr_params(baseMat); // (a medium's coefficients are the material's constants: no textured input reaches them)
        throughput = throughput * v3(expf_poly(-o.sigmaA.x * distance), expf_poly(-o.sigmaA.y * distance), expf_poly(-o.sigmaA.z * distance));
      }
    } else { // :174-184: the medium on top of the stack
      const Medium& md = pl.media[mediumIdx - 1];
      prevMediumIor = md.ior.x;
      if (mediumIdx > 1) nextMediumIor = pl.media[mediumIdx - 2].ior.x;
      throughput = throughput * v3(expf_poly(-md.sigma_t.x * distance), expf_poly(-md.sigma_t.y * distance), expf_poly(-md.sigma_t.z * distance));
    }
  }
  const bool thinWalled = mat.klass == ORC_MAT_OPEN_PBR && mat.p[ORC_P_THIN_WALLED] != 0.0f; // mdl_thin_walled (:155-157)
  st.thinWalled = thinWalled;
  st.sssVolume = stackSize > 0; // a medium stack exists: OpenPBR's volumetric subsurface lobe is live
  st.ior1 = (st.frontFace || thinWalled) ? prevMediumIor : -1.0f; // iorCurrent / iorOther (:188-189)
  st.ior2 = (st.frontFace || thinWalled) ? -1.0f : nextMediumIor;

  // 5. emission (:293-343).  uniform EDF: edf*intensity == emission colour, pdf>0 iff cos>0
  V3 em = v3(mat.p + ORC_P_EMISSION);
  if (em.x != 0.0f || em.y != 0.0f || em.z != 0.0f) {
    if (st.frontFace || !isDoubleSided) {
      float c = dot(-rayDir, st.normal);
      if (c > 0.0f) {
        if (mat.klass == ORC_MAT_OPEN_PBR) { // emission_edf (open_pbr_surface.mtlx:590-619): seen through the coat
          float cior = mat.p[ORC_P_COAT_IOR], qc = (cior - 1.0f) / (cior + 1.0f);
          em = em * opbr_emission_factor(mat.p[ORC_P_CLEARCOAT], v3(mat.p + ORC_P_COAT_COLOR), qc * qc, c);
        }
        radiance = radiance + throughput * (em * F.ubo.sensorExposureScale);
      }
    }
  }

  // 6. BSDF importance sampling (:361-389)
  float xi[4]; xi[0] = next1f(pl.rng); xi[1] = next1f(pl.rng); xi[2] = next1f(pl.rng); xi[3] = next1f(pl.rng);
  BsdfSample bs; bsdf_sample(mat, st, -rayDir, xi, bs);
  uint32_t eventType = bs.event;
  throughput = throughput * bs.overPdf;
  pl.dir = bs.k2;
  bool isTransmission = (eventType & EV_TRANSMISSION) != 0;

  // 7. NEE (:394-444)
  if (F.rs->nextEventEstimation) {
    if ((eventType & (EV_DIFFUSE | EV_GLOSSY)) != 0) {
      float k4[4]; k4[0] = next1f(pl.rng); k4[1] = next1f(pl.rng); k4[2] = next1f(pl.rng); k4[3] = next1f(pl.rng);
      V3 dirToLight; float lightDist; V3 lightPower; float invPdf; uint32_t ds;
      sample_light(P, F.ubo, k4, st.position, dirToLight, lightDist, lightPower, invPdf, ds);
      V3 nee = v3(0, 0, 0);
      bool valid = (lightDist > 0.0f) && dot(dirToLight, st.geomNormal) > 0.0f;
      if (valid) {
        BsdfEval ev; bsdf_evaluate(mat, st, -rayDir, dirToLight, ev);
        if (ev.pdf > 0.0f) {
          float d = f16_to_f32((uint16_t)(ds & 0xffffu)), s = f16_to_f32((uint16_t)(ds >> 16));
          V3 weight = throughput * (lightPower * invPdf);
          nee = nee + (weight * ev.diffuse) * d;
          nee = nee + (weight * ev.glossy) * s;
        }
      }
      pl.neeToLight = dirToLight * lightDist;
      pl.neeContrib = nee;
    }
  }
  // medium toggle (:447-480): MEDIUM_STACK_SIZE == 0 -> inside/outside bit; the walk counter is reset.  Thin-walled surfaces have the
  // same medium on both sides: nothing changes (:447)
  if (!thinWalled && isTransmission) {
    if (stackSize > 0) { // :450-473
      if (st.frontFace) { // push
        mediumIdx++;
        if (mediumIdx <= stackSize) {
          Medium md; md.ior = v3(1, 1, 1); md.sigma_s = v3(0, 0, 0); md.sigma_t = v3(0, 0, 0); md.bias = 0.0f;
          if (mat.klass == ORC_MAT_OPEN_PBR) { // mdl_ior, mdl_volume_{absorption,scattering}_coefficient, MEDIUM_DIRECTIONAL_BIAS
            OpbrParams o = opbr_params(baseMat, true); // (the material's constants, see above)
            md.ior = v3(o.eta, o.eta, o.eta); md.sigma_s = o.sigmaS; md.sigma_t = o.sigmaA + o.sigmaS; md.bias = o.anisotropy;
            if (eventType & EV_SUBSURFACE) { md.sigma_s = o.sssSigmaS; md.sigma_t = o.sssSigmaT; md.bias = o.ssAniso; } // entered through the subsurface lobe
          }
          pl.media[mediumIdx - 1] = md;
        }
      } else if (mediumIdx > 0) mediumIdx--; // pop
    } else mediumIdx = 1u - mediumIdx; // toggle between inside and outside
    pl.bitfield &= ~WALK_MASK;
    pl.bitfield = (pl.bitfield & ~MEDIUM_MASK) | ((mediumIdx << 24) & MEDIUM_MASK);
  }
  if (eventType == EV_ABSORB) pl.bitfield |= TERMINATE_FLAG; // :483-486
  V3 gn = st.geomNormal * (isTransmission ? -1.0f : 1.0f);
  pl.origin = offset_ray_origin(st.position, gn);              // :488-489
  pl.throughput = throughput; pl.radiance = radiance;
}

// rp_main.rgen:49-69 sampleDistance: the extinction coefficient of the channel picked in proportion to throughput * albedo
inline float sample_distance(V3 albedo, V3 throughput, V3 sigma_t, float xi, V3& pdf)
{
  V3 weights = throughput * albedo;
  float sum = (weights.x + weights.y) + weights.z;
  pdf = (sum > 1e-9f) ? (weights / sum) : v3(1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 3.0f);
  return (xi < pdf.x) ? sigma_t.x : ((xi < (pdf.x + pdf.y)) ? sigma_t.y : sigma_t.z);
}
// rp_main.rgen:72-82 sampleHenyeyGreensteinCos
inline float henyey_greenstein_cos(float r, float g)
{
  if (fabsf(g) < 1e-3f) return 1.0f - 2.0f * r;
  float sq = (1.0f - g * g) / ((1.0f - g) + (2.0f * g) * r);
  return ((1.0f + g * g) - sq * sq) / (2.0f * g);
}
// rp_main.rgen:100-113 russian_roulette: true = terminate
inline bool russian_roulette(float k, float rrInvMinTermProb, V3& throughput)
{
  float mt = fmax2(throughput.x, fmax2(throughput.y, throughput.z));
  float p = fmin2(mt, rrInvMinTermProb);
  if (k > p) return true;
  throughput = throughput / p;
  return false;
}

// rp_main.miss:55-86 with the 1x1 fallback dome (Gi.cpp:2184-2199, 2232-2238): texel = u8(clear*255)/255
inline V3 quat_rotate_dir(const float q[4], V3 dir) // rp_main.miss:38-44
{
  V3 qv = v3(q[0], q[1], q[2]);
  V3 a = cross(qv, dir);
  V3 b = cross(qv, a);
  return dir + ((a * q[3]) + b) * 2.0f;
}
void miss(const Frame& F, Payload& pl)
{
  if (F.rs->mediumStackSize > 0) { // rp_main.miss:16-34, 57-66: inside a medium the segment ended in a scattering event
    uint32_t mediumIdx = payload_medium_idx(pl.bitfield, F.rs->mediumStackSize < MAX_MEDIUM_STACK ? F.rs->mediumStackSize : MAX_MEDIUM_STACK);
    if (mediumIdx > 0) {
      float distance = pl.tMaxLast * F.rs->metersPerSceneUnit;
      const Medium& m = pl.media[mediumIdx - 1];
      V3 tr = v3(expf_poly(m.sigma_t.x * -distance), expf_poly(m.sigma_t.y * -distance), expf_poly(m.sigma_t.z * -distance));
      V3 density = m.sigma_t * tr;
      float pdf = dot(pl.walkSegmentPdf, density);
      pl.throughput = pl.throughput * ((m.sigma_s * tr) / pdf);
      pl.origin = pl.origin + pl.dir * distance;
      pl.bitfield |= WALK_MISS_FLAG;
      payload_increment_walk(pl.bitfield);
      return;
    }
  }
  pl.bitfield |= TERMINATE_FLAG;
  const OrcDomeLight* dome = F.P->dome;
  if (!dome) { pl.radiance = pl.radiance + pl.throughput * F.background; return; } // fallback dome, emission multiplier 1 (Gi.cpp:2385)
  bool isPrimaryRay = (pl.bitfield & BOUNCES_MASK) == 0;
  bool useFallback = !F.rs->domeLightCameraVisible && isPrimaryRay; // :76-80
  V3 mult = v3(dome->baseEmission);
  V3 texel = F.background;
  if (!useFallback) {
    V3 d = normalize(quat_rotate_dir(dome->rotation, pl.dir)); // :83
    float u = (atan2f_poly(d.z, d.x) + 0.5f * ORC_PI) / (2.0f * ORC_PI); // :48-49
    float v = 1.0f - acosf_poly(d.y) / ORC_PI;
    F4v t = sample_bilinear_repeat(F.P->textures[dome->texture], u, v);
    texel = v3(t.x, t.y, t.z);
  }
  pl.radiance = pl.radiance + pl.throughput * (texel * mult); // :84-86
}

// colormap_viridis (colormap.glsl:3-14)
inline V3 colormap_viridis(float t)
{
  const V3 c0 = v3(0.2777273272234177f, 0.005407344544966578f, 0.3340998053353061f);
  const V3 c1 = v3(0.1050930431085774f, 1.404613529898575f, 1.384590162594685f);
  const V3 c2 = v3(-0.3308618287255563f, 0.214847559468213f, 0.09509516302823659f);
  const V3 c3 = v3(-4.634230498983486f, -5.799100973351585f, -19.33244095627987f);
  const V3 c4 = v3(6.228269936347081f, 14.17993336680509f, 56.69055260068105f);
  const V3 c5 = v3(4.776384997670288f, -13.74514537774601f, -65.35303263337234f);
  const V3 c6 = v3(-5.435455855934631f, 4.645852612178535f, 26.3124352495832f);
  return c0 + (c1 + (c2 + (c3 + (c4 + (c5 + c6 * t) * t) * t) * t) * t) * t;
}
// colormap_inferno (colormap.glsl:42-53)
inline V3 colormap_inferno(float t)
{
  const V3 c0 = v3(0.0002189403691192265f, 0.001651004631001012f, -0.01948089843709184f);
  const V3 c1 = v3(0.1065134194856116f, 0.5639564367884091f, 3.932712388889277f);
  const V3 c2 = v3(11.60249308247187f, -3.972853965665698f, -15.9423941062914f);
  const V3 c3 = v3(-41.70399613139459f, 17.43639888205313f, 44.35414519872813f);
  const V3 c4 = v3(77.162935699427f, -33.40235894210092f, -81.80730925738993f);
  const V3 c5 = v3(-71.31942824499214f, 32.62606426397723f, 73.20951985803202f);
  const V3 c6 = v3(25.13112622477341f, -12.24266895238567f, -23.07032500287172f);
  return c0 + (c1 + (c2 + (c3 + (c4 + (c5 + c6 * t) * t) * t) * t) * t) * t;
}
struct PathDebug { int lastNee = -1; uint32_t lastBounces = 0; uint32_t totalSegments = 0; }; // NEE / Bounces AOV sources (rp_main.rgen:431-435, 483-486)

void render_pixel(const Frame& F, uint32_t px, uint32_t py, const float* prevColor, float* out, OrcCounters& cnt, PathDebug* dbg = nullptr)
{
  const OrcSettings& rs = *F.rs;
  uint32_t pixelIndex = px + py * F.width; // rp_main.rgen:195
  float invSpp = 1.0f / (float)rs.spp;
  V3 pixelColor = v3(0, 0, 0);
  uint32_t maxBounces = rs.maxBounces < BOUNCES_MASK ? rs.maxBounces : BOUNCES_MASK; // :292
  for (uint32_t s = 0; s < rs.spp; s++) { // :215
    uint32_t sampleIndex = rs.sampleOffset + s;
    uint32_t rng = rng_init(pixelIndex, sampleIndex);
    float r0 = next1f(rng), r1 = next1f(rng); // :224 (always drawn)
    float sox = 0.5f, soy = 0.5f;
    if (rs.jitteredSampling) {
      if (rs.filterImportanceSampling) { float gx, gy; fis_gauss(r0, r1, gx, gy); sox = 0.5f + gx; soy = 0.5f + gy; }
      else { sox = r0; soy = r1; }
    }
    V3 Pp = (F.L + (F.camRight * ((float)px + sox)) * F.WX) + (F.camUp * ((float)py + soy)) * F.HY; // :239-242 (GLSL evaluates left to right)
    V3 origin = F.camPos;
    V3 dir = normalize(Pp - origin);
    if (rs.depthOfField && F.lensRadius > 0.0f) { // :249-263
      float z0 = next1f(rng), z1 = next1f(rng);
      V3 focal = origin + dir * F.cam->focusDistance;
      V3 ap = sample_hemisphere(z0, z1);
      origin = origin + F.camRight * (ap.x * F.lensRadius);
      origin = origin + F.camUp * (ap.y * F.lensRadius);
      dir = normalize(focal - origin);
    }
    if (dir.x == 0.0f) dir.x += ORC_FLT_MIN; // :271
    if (dir.y == 0.0f) dir.y += ORC_FLT_MIN;
    if (dir.z == 0.0f) dir.z += ORC_FLT_MIN;
    Payload pl; pl.throughput = v3(1, 1, 1); pl.bitfield = 0; pl.radiance = v3(0, 0, 0); pl.rng = rng;
    pl.origin = origin; pl.dir = dir; pl.neeToLight = v3(0, 0, 0); pl.neeContrib = v3(0, 0, 0);
    float cosCone = fmax2(1e-5f, dot(dir, F.camFwd)); // :287
    float clipNear = F.clipNear / cosCone, clipFar = F.clipFar / cosCone;
    while (true) { // :295
      uint32_t bounce = pl.bitfield & BOUNCES_MASK;
      if (bounce >= maxBounces || (pl.bitfield & TERMINATE_FLAG)) break;
      float tMin = 0.0f, tMax = ORC_FLT_MAX;
      if (rs.clippingPlanes && bounce == 0) { tMin = clipNear; tMax = clipFar; }
      const uint32_t stackSize = rs.mediumStackSize < MAX_MEDIUM_STACK ? rs.mediumStackSize : MAX_MEDIUM_STACK;
      uint32_t mediumIdx = 0;
      if (stackSize > 0) { // :317-346: distance to the next collision inside a scattering medium
        mediumIdx = payload_medium_idx(pl.bitfield, stackSize);
        if (mediumIdx > 0) {
          const Medium& m = pl.media[mediumIdx - 1];
          pl.walkSegmentPdf = v3(1, 1, 1);
          uint32_t walkLength = (pl.bitfield & WALK_MASK) >> WALK_OFFSET;
          bool hasScattering = m.sigma_s.x > 0.0f || m.sigma_s.y > 0.0f || m.sigma_s.z > 0.0f;
          if (hasScattering && walkLength <= rs.maxVolumeWalkLength) {
            V3 albedo = v3(safe_div(m.sigma_s.x, m.sigma_t.x), safe_div(m.sigma_s.y, m.sigma_t.y), safe_div(m.sigma_s.z, m.sigma_t.z));
            float x0 = next1f(pl.rng), x1 = next1f(pl.rng);
            float sg = sample_distance(albedo, pl.throughput, m.sigma_t, x0, pl.walkSegmentPdf);
            sg = sg * rs.metersPerSceneUnit;
            tMax = -logf_poly(1.0f - x1) / sg; // collision free distance
          }
        }
      }
      pl.tMaxLast = tMax;
      pl.neeContrib = v3(0, 0, 0); // :349
      Hit h;
      cnt.segments++; if (bounce < 64) cnt.bounceHistogram[bounce]++;
      if (trace_closest(*F.P, pl.origin, pl.dir, tMin, tMax, h, pl.rng)) { cnt.hits++; closest_hit(F, h, pl, h.t); }
      else miss(F, pl);
      if (rs.nextEventEstimation) { // :397-438
        float lightDist = length(pl.neeToLight);
        V3 sdir = safe_div(pl.neeToLight, lightDist);
        bool traceRay = luminance(pl.neeContrib) > 1e-6f && lightDist > 1e-9f;
        bool shadowed = true;
        if (traceRay) { cnt.shadowRays++; shadowed = trace_any(*F.P, pl.origin, sdir, 0.01f, lightDist, pl.rng); }
        if (traceRay && !shadowed) pl.radiance = pl.radiance + pl.neeContrib;
        // NEE AOV (:431-435): written at bounce 0 only, for EVERY sample (the last one wins).  The reference dispatches the shadow ray
        // unconditionally (tMin = tMax = 0 when it cannot contribute, :406-410): the empty interval misses, rp_main_shadow.miss clears
        // `shadowed`, so an untraced shadow ray shows as "not shadowed" (green).
        if (dbg && bounce == 0) dbg->lastNee = (traceRay && shadowed) ? 1 : 0;
      }
      if (length(pl.throughput) < 1e-9f) pl.bitfield |= TERMINATE_FLAG; // :441-444
      if (bounce > rs.rrBounceOffset) { // :447-459
        float k1 = next1f(pl.rng);
        if (russian_roulette(k1, rs.rrInvMinTermProb, pl.throughput)) pl.bitfield |= TERMINATE_FLAG;
      }
      if (stackSize > 0 && (pl.bitfield & WALK_MISS_FLAG) != 0) { // :462-477: continue the random walk in a new direction
        float x0 = next1f(pl.rng), x1 = next1f(pl.rng);
        float g = pl.media[mediumIdx - 1].bias; // mediumIdx as read at the top of this iteration
        float cosTheta = henyey_greenstein_cos(x0, g);
        float sinTheta = sqrtf(fmax2(0.0f, 1.0f - cosTheta * cosTheta));
        float sp, cp; sincos2pi(x1, &sp, &cp); // phi = 2 pi xi.y
        V3 t, b; orthonormal_basis(pl.dir, t, b);
        pl.dir = ((t * sinTheta) * cp + (b * sinTheta) * sp) + pl.dir * cosTheta; // :95
        pl.bitfield &= ~WALK_MISS_FLAG;
      }
      pl.bitfield++; // :480
    }
    if (dbg) { dbg->lastBounces = pl.bitfield & BOUNCES_MASK; dbg->totalSegments += dbg->lastBounces; } // :483-486
    V3 rad = pl.radiance; // :489-498
    float mv = fmax2(rad.x, fmax2(rad.y, rad.z));
    if (mv > rs.maxSampleValue) rad = rad * (rs.maxSampleValue / mv);
    V3 sc = v3(fmax2(0.0f, rad.x), fmax2(0.0f, rad.y), fmax2(0.0f, rad.z));
    pixelColor = pixelColor + sc * invSpp;
    cnt.samples++;
  }
  V3 prev = pixelColor; // :506-515
  if (rs.progressiveAccumulation && rs.sampleOffset > 0 && prevColor) prev = v3(prevColor);
  float invTotal = 1.0f / (float)(rs.sampleOffset + rs.spp); // Gi.cpp:2414
  V3 c = (prev * (float)rs.sampleOffset + pixelColor * (float)rs.spp) * invTotal;
  out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = 1.0f;
}

void make_frame(Frame& F, const Prepared& P, const OrcCamera* cam, const OrcSettings* rs, const OrcRegion* rg)
{
  F.P = &P; F.cam = cam; F.rs = rs; F.width = rg->imageWidth; F.height = rg->imageHeight;
  F.camPos = v3(cam->position);
  F.camFwd = normalize(v3(cam->forward)); F.camUp = normalize(v3(cam->up)); // Gi.cpp:2375-2376
  F.camRight = cross(F.camFwd, F.camUp);                                    // rp_main.rgen:199
  float aspect = (float)F.width / (float)F.height;
  float H = 1.0f, W = H * aspect;
  float d = H / (2.0f * tanf(cam->vfov * 0.5f));                            // :204 (host-side libm; per-frame constant)
  F.WX = W / (float)F.width; F.HY = H / (float)F.height;
  V3 C = F.camPos + F.camFwd * d;
  F.L = (C - F.camRight * W * 0.5f) - F.camUp * H * 0.5f;                   // :210 (left-to-right: (right*W)*0.5)
  F.lensRadius = (cam->fStop > 0.0f) ? cam->focalLength / (2.0f * cam->fStop) : 0.0f; // Gi.cpp:2378-2382
  uint32_t cr = pack_half2x16(cam->clipStart, cam->clipEnd);               // Gi.cpp:2419
  F.clipNear = f16_to_f32((uint16_t)(cr & 0xffffu)); F.clipFar = f16_to_f32((uint16_t)(cr >> 16));
  for (int i = 0; i < 3; i++) { // Gi.cpp:2196 glm::u8vec4(bg*255) truncates; RGBA8 unorm texel
    float v = rs->clearColor[i] * 255.0f;
    int q = (int)v; if (q < 0) q = 0; if (q > 255) q = q & 255; // u8 conversion wraps; clamp negatives to 0
    float t = (float)q / 255.0f;
    if (i == 0) F.background.x = t; else if (i == 1) F.background.y = t; else F.background.z = t;
  }
  F.ubo.sphereCount = (uint32_t)P.sphere.size(); F.ubo.distantCount = (uint32_t)P.distant.size();
  F.ubo.rectCount = (uint32_t)P.rect.size(); F.ubo.diskCount = (uint32_t)P.disk.size();
  F.ubo.total = F.ubo.sphereCount + F.ubo.distantCount + F.ubo.rectCount + F.ubo.diskCount;
  F.ubo.lightIntensityMultiplier = rs->lightIntensityMultiplier;
  F.ubo.sensorExposureScale = exp2f(cam->exposure); // rp_main.chit:126, 336 (host libm; per-frame constant)
}

// Non-colour AOVs of one pixel: clearAovs (rp_main.rgen:132-183), the bounce-0 writes of every sample in order
// (rp_main.chit:192-290) and the final normal renormalisation (rp_main.rgen:517-520).
void render_pixel_aovs(const Frame& F, uint32_t px, uint32_t py, size_t o, OrcAovs& A)
{
  const OrcSettings& rs = *F.rs;
  uint32_t pixelIndex = px + py * F.width;
  auto put3 = [&](float* buf, V3 v) { if (buf) { buf[4 * o] = v.x; buf[4 * o + 1] = v.y; buf[4 * o + 2] = v.z; } };
  auto clr3 = [&](float* buf, int id) { put3(buf, v3(A.clear[id][0], A.clear[id][1], A.clear[id][2])); };
  auto clri = [&](int32_t* buf, int id) { if (buf) memcpy(&buf[o], &A.clear[id][0], 4); };
  clr3(A.barycentrics, 3); clr3(A.texcoords, 4); clr3(A.opacity, 7); clr3(A.tangents, 8); clr3(A.bitangents, 9); clr3(A.thinWalled, 10);
  clri(A.objectId, 11); if (A.depth) A.depth[o] = A.clear[12][0]; clri(A.faceId, 13); clri(A.instanceId, 14); clr3(A.doubleSided, 15);
  if (rs.sampleOffset == 0) { clr3(A.normal, 1); clr3(A.albedo, 16); }
  float invTotal = 1.0f / (float)(rs.sampleOffset + rs.spp);
  for (uint32_t s = 0; s < rs.spp; s++) {
    uint32_t rng = rng_init(pixelIndex, rs.sampleOffset + s);
    float r0 = next1f(rng), r1 = next1f(rng);
    float sox = 0.5f, soy = 0.5f;
    if (rs.jitteredSampling) {
      if (rs.filterImportanceSampling) { float gx, gy; fis_gauss(r0, r1, gx, gy); sox = 0.5f + gx; soy = 0.5f + gy; }
      else { sox = r0; soy = r1; }
    }
    V3 Pp = (F.L + (F.camRight * ((float)px + sox)) * F.WX) + (F.camUp * ((float)py + soy)) * F.HY;
    V3 origin = F.camPos, dir = normalize(Pp - origin);
    if (rs.depthOfField && F.lensRadius > 0.0f) {
      float z0 = next1f(rng), z1 = next1f(rng);
      V3 focal = origin + dir * F.cam->focusDistance;
      V3 ap = sample_hemisphere(z0, z1);
      origin = origin + F.camRight * (ap.x * F.lensRadius);
      origin = origin + F.camUp * (ap.y * F.lensRadius);
      dir = normalize(focal - origin);
    }
    if (dir.x == 0.0f) dir.x += ORC_FLT_MIN;
    if (dir.y == 0.0f) dir.y += ORC_FLT_MIN;
    if (dir.z == 0.0f) dir.z += ORC_FLT_MIN;
    float tMin = 0.0f, tMax = ORC_FLT_MAX;
    if (rs.clippingPlanes) { float cc = fmax2(1e-5f, dot(dir, F.camFwd)); tMin = F.clipNear / cc; tMax = F.clipFar / cc; }
    Hit h;
    if (!trace_closest(*F.P, origin, dir, tMin, tMax, h, rng)) continue;
    State st; const MeshData* mesh;
    setup_shading_state(*F.P, h, dir, st, mesh);
    const Tri& T = F.P->tris[h.tri];
    const Instance& inst = F.P->instances[T.instance];
    { // chit:199-205: (1,0,0) without cutout transparency; else what the any-hit shader wrote (ahit:45-49), restated as the ACCEPTED
      // primary hit's opacity through viridis (white for 0) -- the reference keeps the last candidate's, in driver order
      V3 c = v3(1, 0, 0);
      if (T.cutout < 1.0f || T.opacityTexMat >= 0) { float op = T.opacityTexMat >= 0 ? cutout_opacity_textured(*F.P, T, h.u, h.v) : T.cutout; c = (op == 0.0f) ? v3(1, 1, 1) : colormap_viridis(op); }
      put3(A.opacity, c);
    }
    put3(A.tangents, (st.tangentU + v3(1, 1, 1)) * 0.5f);                                  // :206-208
    put3(A.bitangents, (st.tangentV + v3(1, 1, 1)) * 0.5f);                                // :209-211
    put3(A.barycentrics, v3(1.0f - h.u - h.v, h.u, h.v));                                  // :212-214
    put3(A.texcoords, v3(st.u, st.v, 0.0f));                                               // :215-217
    { const OrcMaterial& tm = F.P->materials[mesh->material];                              // :218-220
      put3(A.thinWalled, (tm.klass == ORC_MAT_OPEN_PBR && tm.p[ORC_P_THIN_WALLED] != 0.0f) ? v3(1, 0, 0) : v3(0, 1, 0)); }
    if (A.objectId) A.objectId[o] = mesh->objectId;                                        // :221-224
    if (A.depth) A.depth[o] = 2.0f * logf_poly(h.t / F.clipNear) / logf_poly(F.clipFar / F.clipNear) - 1.0f; // :225-229
    if (A.faceId) {                                                                        // :230-240 (incl. the reference's mask)
      int stride = (int)mesh->faceIdStride, invStride = 4 / stride;
      int32_t word; memcpy(&word, &mesh->faceIdData[(size_t)(T.prim / (uint32_t)invStride) * 4], 4);
      word >>= (int)((T.prim % (uint32_t)invStride) * 8);
      A.faceId[o] = word & (stride * 8 - 1);
    }
    if (A.instanceId) A.instanceId[o] = inst.instanceId;                                   // :241-244
    put3(A.doubleSided, (mesh->flags & 2u) ? v3(0, 1, 0) : v3(1, 0, 0));                   // :245-247
    if (A.normal) {                                                                        // :250-259
      V3 pos = (st.normal + v3(1, 1, 1)) * 0.5f, prev = pos;
      if (rs.progressiveAccumulation && rs.sampleOffset > 0) prev = v3(A.normal + 4 * o);
      put3(A.normal, (prev * (float)rs.sampleOffset + pos * (float)rs.spp) * invTotal);
    }
    if (A.albedo) {                                                                        // :261-289
      const OrcMaterial& bm = F.P->materials[mesh->material];
      st.cameraPosition = v3(F.cam->position); st.frame = F.rs->frame;
      OrcMaterial rm; if (material_textured(bm)) rm = resolve_material(*F.P, bm, st, dir);
      V3 al = bsdf_albedo(material_textured(bm) ? rm : bm, st, -dir), prev = al;
      if (rs.progressiveAccumulation && rs.sampleOffset > 0) prev = v3(A.albedo + 4 * o);
      put3(A.albedo, (prev * (float)rs.sampleOffset + al * (float)rs.spp) * invTotal);
    }
  }
  if (A.normal) { // rgen:517-520
    V3 n = v3(A.normal + 4 * o) * 2.0f - v3(1, 1, 1);
    put3(A.normal, (normalize(n) + v3(1, 1, 1)) * 0.5f);
  }
}

} // namespace

// ---------------------------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------------------------
extern "C" {
float orc_atan2f(float y, float x) { return atan2f_poly(y, x); }
float orc_acosf(float x) { return acosf_poly(x); }
// the raw sampler alone (what stands in for the hardware sampler when the reference's tex_lookup_float4_2d runs on the CPU, oracle/ref/ref_shim.cpp)
void orc_dbg_sample_bilinear(const OrcTexture* t, float u, float v, float* out4) { F4v r = sample_bilinear_repeat(*t, u, v); out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w; }
// The MDL runtime's remaining texture entry points: kind 0 tex_texel_float4_2d(x, y), 1 tex_resolution_2d, 2 tex_lookup_float4_3d(u, v, w, wraps), 3 tex_texel_float4_3d(x, y, z).
// q = (kind, valid (0 = the invalid texture), c0, c1, c2, wrapU, wrapV, wrapW); integer coordinates are passed as exact floats.
void orc_tex_runtime(const float* rgba, uint32_t w, uint32_t h, uint32_t d, uint32_t count, const float* queries, float* out)
{
  const OrcTexture t2{rgba, w, h}; const Tex3 t3{rgba, w, h, d};
  for (uint32_t i = 0; i < count; i++) {
    const float* q = queries + 8 * (size_t)i; float* o = out + 4 * (size_t)i;
    const int kind = (int)q[0]; const bool valid = q[1] != 0.0f;
    F4v r{0, 0, 0, 0};
    if (kind == 0) r = tex_texel_float4_2d(valid ? &t2 : nullptr, (int)q[2], (int)q[3]);
    else if (kind == 1) { int res[2]; tex_resolution_2d(valid ? &t2 : nullptr, res); r = F4v{(float)res[0], (float)res[1], 0, 0}; }
    else if (kind == 2) r = tex_lookup_float4_3d(valid ? &t3 : nullptr, q[2], q[3], q[4], (int)q[5], (int)q[6], (int)q[7]);
    else r = tex_texel_float4_3d(valid ? &t3 : nullptr, (int)q[2], (int)q[3], (int)q[4]);
    o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
  }
}
void orc_dbg_sample_trilinear(const float* rgba, uint32_t w, uint32_t h, uint32_t d, float u, float v, float ww, float* out4)
{ const Tex3 t{rgba, w, h, d}; F4v r = sample_trilinear_repeat(t, u, v, ww); out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w; }
void orc_scene_data_lookup_float4x4(const float* defaultValue, float* out) { scene_data_lookup_float4x4(defaultValue, out); }
void orc_tex_lookup(const OrcTexture* t, float u, float v, int wrapU, int wrapV, float* out4)
{
  F4v r = tex_lookup_float4_2d(*t, u, v, wrapU, wrapV);
  out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}


int orc_render(const OrcScene* scene, const OrcCamera* camera, const OrcSettings* settings,
               const OrcRegion* region, const float* prevColor, float* colorOut,
               OrcCounters* counters, int threads)
{
  if (!scene || !camera || !settings || !region || !colorOut) return 1;
  if (settings->spp == 0 || region->imageWidth == 0 || region->imageHeight == 0 || region->rowEnd > region->imageHeight || region->rowBegin > region->rowEnd) return 2;
  if (region->imageWidth > 65535u || region->imageHeight > 65535u) return 3; // imageDims packing, rp_main.h:38
  std::vector<uint32_t> list(region->rowEnd - region->rowBegin);
  for (uint32_t r = 0; r < (uint32_t)list.size(); r++) list[r] = region->rowBegin + r;
  return orc_render_rows(scene, camera, settings, region, (uint32_t)list.size(), list.data(), prevColor, colorOut, counters, threads);
}

// The same for an explicit list of image rows (test infrastructure: one prepare() / BVH build for rows that are not adjacent, e.g. two rows of one rank's
// interleaved share).  Output row r of colorOut / prevColor is image row rowList[r]; region supplies the image size only.
int orc_render_rows(const OrcScene* scene, const OrcCamera* camera, const OrcSettings* settings, const OrcRegion* region, uint32_t rowCount, const uint32_t* rowList,
                    const float* prevColor, float* colorOut, OrcCounters* counters, int threads)
{
  if (!scene || !camera || !settings || !region || !colorOut || (rowCount && !rowList)) return 1;
  if (settings->spp == 0 || region->imageWidth == 0 || region->imageHeight == 0) return 2;
  if (region->imageWidth > 65535u || region->imageHeight > 65535u) return 3;
  for (uint32_t r = 0; r < rowCount; r++) if (rowList[r] >= region->imageHeight) return 2;
  Prepared P; prepare(scene, P);
  Frame F; make_frame(F, P, camera, settings, region);
  if (threads <= 0) threads = 1;
  // work unit: 32 pixels of one row (whole rows left a 2-row request of a 4K frame on 2 of 256 cores: 13 minutes for 8 M samples)
  const uint32_t CHUNK = 32u, chunksPerRow = (F.width + CHUNK - 1u) / CHUNK;
  const uint64_t units = (uint64_t)rowCount * chunksPerRow;
  std::vector<OrcCounters> tc((size_t)threads);
  for (auto& c : tc) memset(&c, 0, sizeof(c));
  std::atomic<uint64_t> nextUnit{0};
  auto work = [&](int ti) {
    for (;;) {
      const uint64_t u = nextUnit.fetch_add(1);
      if (u >= units) break;
      const uint32_t r = (uint32_t)(u / chunksPerRow), x0 = (uint32_t)(u % chunksPerRow) * CHUNK, x1 = x0 + CHUNK < F.width ? x0 + CHUNK : F.width;
      uint32_t y = rowList[r];
      for (uint32_t x = x0; x < x1; x++) {
        size_t o = ((size_t)r * F.width + x) * 4;
        render_pixel(F, x, y, prevColor ? prevColor + o : nullptr, colorOut + o, tc[(size_t)ti]);
      }
    }
  };
  if (threads == 1) work(0);
  else { std::vector<std::thread> th; for (int i = 0; i < threads; i++) th.emplace_back(work, i); for (auto& t : th) t.join(); }
  if (counters) {
    memset(counters, 0, sizeof(*counters));
    for (auto& c : tc) {
      counters->samples += c.samples; counters->segments += c.segments; counters->shadowRays += c.shadowRays; counters->hits += c.hits;
      for (int i = 0; i < 64; i++) counters->bounceHistogram[i] += c.bounceHistogram[i];
    }
  }
  return 0;
}

// Turbo colour map, polynomial fit (the reference indexes Google's 256-entry table, Gi.cpp:338-341)
static void turbo_colormap(float x, float* rgb)
{
  const float x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x;
  rgb[0] = (((0.13572138f + 4.61539260f * x) + -42.66032258f * x2) + 132.13108234f * x3) + (-152.94239396f * x4 + 59.28637943f * x5);
  rgb[1] = (((0.09140261f + 2.19418839f * x) + 4.84296658f * x2) + -14.18503333f * x3) + (4.27729857f * x4 + 2.82956604f * x5);
  rgb[2] = (((0.10667330f + 12.64194608f * x) + -60.58204836f * x2) + 110.36276771f * x3) + (-89.90310912f * x4 + 27.34824973f * x5);
}

int orc_render_aovs(const OrcScene* scene, const OrcCamera* camera, const OrcSettings* settings, const OrcRegion* region, OrcAovs* aovs)
{
  if (!scene || !camera || !settings || !region || !aovs) return 1;
  Prepared P; prepare(scene, P);
  Frame F; make_frame(F, P, camera, settings, region);
  std::vector<uint32_t> segments(aovs->clockCycles ? (size_t)(region->rowEnd - region->rowBegin) * F.width : 0);
  for (uint32_t y = region->rowBegin; y < region->rowEnd; y++)
    for (uint32_t x = 0; x < F.width; x++) {
      const size_t o = (size_t)(y - region->rowBegin) * F.width + x;
      render_pixel_aovs(F, x, y, o, *aovs);
      if (aovs->nee || aovs->bounces || aovs->clockCycles) { // these follow the whole paths of the colour pass
        PathDebug dbg; OrcCounters cnt{}; float colour[4];
        render_pixel(F, x, y, nullptr, colour, cnt, &dbg);
        uint32_t maxBounces = settings->maxBounces < BOUNCES_MASK ? settings->maxBounces : BOUNCES_MASK;
        if (aovs->bounces) { V3 c = colormap_inferno((float)dbg.lastBounces / (float)maxBounces); aovs->bounces[4 * o] = c.x; aovs->bounces[4 * o + 1] = c.y; aovs->bounces[4 * o + 2] = c.z; }
        if (aovs->nee) {
          V3 c = dbg.lastNee < 0 ? v3(aovs->clear[2]) : (dbg.lastNee ? v3(1, 0, 0) : v3(0, 1, 0));
          aovs->nee[4 * o] = c.x; aovs->nee[4 * o + 1] = c.y; aovs->nee[4 * o + 2] = c.z;
        }
        if (aovs->clockCycles) segments[o] = dbg.totalSegments;
      }
    }
  if (aovs->clockCycles) { // _EncodeRenderBufferAsHeatmap (Gi.cpp:327-343) over the cost proxy
    float maxValue = 0.0f;
    for (uint32_t c : segments) maxValue = fmax2(maxValue, (float)c);
    for (size_t o = 0; o < segments.size(); o++) {
      float* dst = aovs->clockCycles + 4 * o;
      if (maxValue > 0.0f) {
        int idx = (int)(((float)segments[o] / maxValue) * 255.0); if (idx > 255) idx = 255;
        turbo_colormap((float)idx / 255.0f, dst);
        dst[3] = 255.0f;
      } else { dst[0] = (float)segments[o]; dst[1] = 0.0f; dst[2] = 0.0f; }
    }
  }
  return 0;
}

uint32_t orc_rng_init(uint32_t pixelIndex, uint32_t sampleIndex) { return rng_init(pixelIndex, sampleIndex); }
float orc_rng_next1f(uint32_t* state) { return next1f(*state); }
uint32_t orc_hash_pcg32(uint32_t* state) { return hash_pcg32(*state); }
uint32_t orc_encode_direction(const float v[3]) { return encode_direction(v3(v)); }
void orc_decode_direction(uint32_t e, float out[3]) { V3 d = decode_direction(e); out[0] = d.x; out[1] = d.y; out[2] = d.z; }
void orc_offset_ray_origin(const float p[3], const float n[3], float out[3]) { V3 r = offset_ray_origin(v3(p), v3(n)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_fis_gauss(float xi0, float xi1, float out[2]) { fis_gauss(xi0, xi1, out[0], out[1]); }
void orc_sincos2pi(float x, float* s, float* c) { sincos2pi(x, s, c); }
float orc_logf(float x) { return logf_poly(x); }
float orc_expf(float x) { return expf_poly(x); }
float orc_film_reflectance(float c, float nf, float n3, float d, float lam) { return film_reflectance(c, nf, n3, d, lam); }
float orc_fresnel_dielectric(float c, float eta) { return fresnel_dielectric(c, eta); }
uint32_t orc_pack_half2x16(float a, float b) { return pack_half2x16(a, b); }
void orc_unpack_half2x16(uint32_t v, float out[2]) { out[0] = f16_to_f32((uint16_t)(v & 0xffffu)); out[1] = f16_to_f32((uint16_t)(v >> 16)); }
void orc_orthonormal_basis(const float n[3], float b1[3], float b2[3]) { V3 a, b; orthonormal_basis(v3(n), a, b); b1[0] = a.x; b1[1] = a.y; b1[2] = a.z; b2[0] = b.x; b2[1] = b.y; b2[2] = b.z; }

int orc_trace_closest(const OrcScene* scene, const float o[3], const float d[3], float tMin, float tMax,
                      float* t, float* u, float* v, uint32_t* instance, uint32_t* prim)
{
  Prepared P; prepare(scene, P);
  Hit h;
  if (!trace_closest(P, v3(o), v3(d), tMin, tMax, h)) return 0;
  *t = h.t; *u = h.u; *v = h.v; *instance = P.tris[h.tri].instance; *prim = P.tris[h.tri].prim;
  return 1;
}

int orc_trace_batch(const OrcScene* scene, uint32_t count, const float* origins, const float* dirs, float tMin, float tMax,
                    float* outTUV, int32_t* outInstPrim)
{
  Prepared P; prepare(scene, P);
  int hits = 0;
  for (uint32_t i = 0; i < count; i++) {
    Hit h;
    if (trace_closest(P, v3(origins + 3 * i), v3(dirs + 3 * i), tMin, tMax, h)) {
      outTUV[3 * i] = h.t; outTUV[3 * i + 1] = h.u; outTUV[3 * i + 2] = h.v;
      outInstPrim[2 * i] = (int32_t)P.tris[h.tri].instance; outInstPrim[2 * i + 1] = (int32_t)P.tris[h.tri].prim; hits++;
    } else {
      outTUV[3 * i] = tMax; outTUV[3 * i + 1] = 0.0f; outTUV[3 * i + 2] = 0.0f; outInstPrim[2 * i] = -1; outInstPrim[2 * i + 1] = -1;
    }
  }
  return hits;
}

void orc_bsdf_debug(const OrcMaterial* mat, uint32_t count, const float* in, float* out)
{
  for (uint32_t i = 0; i < count; i++) {
    const float* p = in + 22 * (size_t)i; float* o = out + 15 * (size_t)i;
    State st; st.normal = v3(p); st.tangentU = v3(p + 3); st.tangentV = v3(p + 6); st.geomNormal = v3(p + 9);
    st.position = v3(0, 0, 0); st.u = st.v = 0.0f; st.frontFace = (p[21] < 0.5f);
    st.thinWalled = mat->klass == ORC_MAT_OPEN_PBR && mat->p[ORC_P_THIN_WALLED] != 0.0f;
    BsdfSample bs; bsdf_sample(*mat, st, v3(p + 12), p + 18, bs);
    BsdfEval ev; bsdf_evaluate(*mat, st, v3(p + 12), v3(p + 15), ev);
    o[0] = bs.k2.x; o[1] = bs.k2.y; o[2] = bs.k2.z; o[3] = bs.overPdf.x; o[4] = bs.overPdf.y; o[5] = bs.overPdf.z; o[6] = bs.pdf; o[7] = (float)bs.event;
    o[8] = ev.diffuse.x; o[9] = ev.diffuse.y; o[10] = ev.diffuse.z; o[11] = ev.glossy.x; o[12] = ev.glossy.y; o[13] = ev.glossy.z; o[14] = ev.pdf;
  }
}


// ---- restated helpers one by one, for tests/test_oracle_ref.py (oracle/_ref = the reference's own functions compiled from /root/reference)
void orc_dbg_sample_hemisphere(float x0, float x1, float* out) { V3 v = sample_hemisphere(x0, x1); out[0] = v.x; out[1] = v.y; out[2] = v.z; }
void orc_dbg_sample_sphere(float x0, float x1, const float* radius, float* out) { V3 v = sample_sphere(x0, x1, v3(radius)); out[0] = v.x; out[1] = v.y; out[2] = v.z; }
void orc_dbg_sample_disk(float x0, float x1, float rx, float ry, float* out) { sample_disk(x0, x1, rx, ry, out[0], out[1]); }
float orc_dbg_luminance(const float* c) { return luminance(v3(c)); }
float orc_dbg_safe_div(float a, float b) { return safe_div(a, b); }
void orc_dbg_colormap(int which, float t, float* out) { V3 v = which == 0 ? colormap_viridis(t) : colormap_inferno(t); out[0] = v.x; out[1] = v.y; out[2] = v.z; }
uint32_t orc_dbg_payload_medium_idx(uint32_t bitfield, uint32_t stackSize) { return payload_medium_idx(bitfield, stackSize); }
uint32_t orc_dbg_payload_increment_walk(uint32_t bitfield) { payload_increment_walk(bitfield); return bitfield; }
int orc_dbg_russian_roulette(float k, float rrInvMinTermProb, float* thr) { V3 t = v3(thr); bool term = russian_roulette(k, rrInvMinTermProb, t); thr[0] = t.x; thr[1] = t.y; thr[2] = t.z; return term ? 1 : 0; }
float orc_dbg_sample_distance(const float* albedo, const float* throughput, const float* sigma_t, float xi, float* pdf)
{ V3 p; float r = sample_distance(v3(albedo), v3(throughput), v3(sigma_t), xi, p); pdf[0] = p.x; pdf[1] = p.y; pdf[2] = p.z; return r; }
float orc_dbg_hg_cos(float r, float g) { return henyey_greenstein_cos(r, g); }
void orc_dbg_quat_rotate_dir(const float* q, const float* dir, float* out) { V3 v = quat_rotate_dir(q, v3(dir)); out[0] = v.x; out[1] = v.y; out[2] = v.z; }
float orc_dbg_apply_wrap_and_crop(float coord, int wrap, int res) { return apply_wrap_and_crop(coord, wrap, res); }
void orc_dbg_adapt_normal(const float* rayDir, const float* geomNormal, const float* normal, float* out) { V3 v = adapt_normal(v3(rayDir), v3(geomNormal), v3(normal)); out[0] = v.x; out[1] = v.y; out[2] = v.z; }

// scene_data_lookup (the restatement above) for one hit, for tests/test_oracle_ref.py: primvars as the boundary hands them over (mesh primvars
// override instancer primvars of the same name), the hit's vertex indices / barycentrics / primitive id / instance id.  Returns 1 if the name resolved.
int orc_dbg_scene_data_lookup(const OrcPrimvar* primvars, uint32_t primvarCount, const OrcPrimvar* instancerPrimvars, uint32_t instancerPrimvarCount, const char* name, int comps,
                              const uint32_t* hitIndices, float bu, float bv, uint32_t prim, int32_t instanceId, const float* cameraPosition, float frame, float* out)
{
  OrcMesh src; memset(&src, 0, sizeof(src));
  src.primvars = primvars; src.primvarCount = primvarCount; src.instancerPrimvars = instancerPrimvars; src.instancerPrimvarCount = instancerPrimvarCount;
  MeshData md; md.faces = nullptr; md.faceCount = 0; md.material = 0; md.flags = 0; md.objectId = 0; md.faceIdStride = 1; md.src = &src;
  State st; st.mesh = &md; st.prim = prim; st.instanceId = instanceId; st.bu = bu; st.bv = bv;
  for (int i = 0; i < 3; i++) st.hitIndices[i] = hitIndices[i];
  st.cameraPosition = v3(cameraPosition); st.frame = frame;
  float v[3] = {0, 0, 0};
  const bool ok = scene_data_lookup(st, name, comps, v);
  for (int c = 0; c < comps && c < 3; c++) out[c] = v[c];
  return ok ? 1 : 0;
}
// setup_shading_state for one triangle given as three GiVertex (48 B each), the mesh transform and one instance transform (4x4 row-major,
// USD convention).  out: position, normal, geom normal, tangentU, tangentV, (u, v, frontFace); fvertexOut (24 floats) = the packed vertices
// the host uploads (rp_main.h:58-64), o2wOut (12) / w2oOut (9) = the composed object-to-world rows and the inverse of its 3x3 part.
void orc_dbg_shading_state(const OrcVertex* verts, const float* meshTransform, const float* instanceTransform, const float* rayDir, float bu, float bv, float* out,
                           float* fvertexOut, float* o2wOut, float* w2oOut)
{
  Prepared P;
  static const uint32_t faces[3] = {0u, 1u, 2u};
  MeshData d; d.faces = faces; d.faceCount = 1; d.material = 0; d.flags = 0; d.objectId = 0; d.faceIdStride = 1; d.src = nullptr;
  for (int i = 0; i < 3; i++) d.verts.push_back(FVertex{v3(verts[i].pos), verts[i].bitangentSign, encode_direction(v3(verts[i].norm)), encode_direction(v3(verts[i].tangent)), verts[i].u, verts[i].v});
  P.meshes.push_back(d);
  Instance inst; inst.mesh = 0; inst.instanceId = 0;
  compose_transform(meshTransform, instanceTransform, inst.o2w);
  invert3x3(inst.o2w, inst.w2o);
  P.instances.push_back(inst);
  P.tris.push_back(Tri{v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), 0u, 0u, 1.0f, -1});
  Hit h; h.t = 1.0f; h.u = bu; h.v = bv; h.tri = 0;
  State st; const MeshData* m;
  setup_shading_state(P, h, v3(rayDir), st, m);
  const V3 o[5] = {st.position, st.normal, st.geomNormal, st.tangentU, st.tangentV};
  for (int i = 0; i < 5; i++) { out[3 * i] = o[i].x; out[3 * i + 1] = o[i].y; out[3 * i + 2] = o[i].z; }
  out[15] = st.u; out[16] = st.v; out[17] = st.frontFace ? 1.0f : 0.0f;
  for (int i = 0; i < 3; i++) {
    const FVertex& v = P.meshes[0].verts[i]; float* f = fvertexOut + 8 * i;
    f[0] = v.pos.x; f[1] = v.pos.y; f[2] = v.pos.z; f[3] = v.bsign; memcpy(&f[4], &v.n, 4); memcpy(&f[5], &v.t, 4); f[6] = v.u; f[7] = v.v;
  }
  memcpy(o2wOut, inst.o2w, sizeof(inst.o2w)); memcpy(w2oOut, inst.w2o, sizeof(inst.w2o));
}
// light arrays in the reference's 48-byte layouts (interface/rp_main.h:73-113), 12 dwords per light
void orc_dbg_sample_light(const uint32_t* counts /* sphere, distant, rect, disk */, float lightIntensityMultiplier, float sensorExposureScale, const float* sphere,
                          const float* distant, const float* rect, const float* disk, const float* k4, const float* pos, float* dirToLight, float* dist, float* power,
                          float* invPdf, uint32_t* dsPacked)
{
  Prepared P;
  auto u = [](float f) { uint32_t x; memcpy(&x, &f, 4); return x; };
  for (uint32_t i = 0; i < counts[0]; i++) { const float* l = sphere + 12 * i; P.sphere.push_back(SphereL{v3(l), u(l[3]), v3(l + 4), l[7], v3(l + 8)}); }
  for (uint32_t i = 0; i < counts[1]; i++) { const float* l = distant + 12 * i; P.distant.push_back(DistantL{v3(l), l[3], v3(l + 4), u(l[7]), l[11]}); }
  for (uint32_t i = 0; i < counts[2]; i++) { const float* l = rect + 12 * i; P.rect.push_back(RectL{v3(l), l[3], v3(l + 4), l[7], u(l[8]), u(l[9]), u(l[10])}); }
  for (uint32_t i = 0; i < counts[3]; i++) { const float* l = disk + 12 * i; P.disk.push_back(DiskL{v3(l), l[3], v3(l + 4), l[7], u(l[8]), u(l[9]), u(l[10])}); }
  Uniforms ubo{counts[0], counts[1], counts[2], counts[3], counts[0] + counts[1] + counts[2] + counts[3], lightIntensityMultiplier, sensorExposureScale};
  V3 d, p; sample_light(P, ubo, k4, v3(pos), d, *dist, p, *invPdf, *dsPacked);
  dirToLight[0] = d.x; dirToLight[1] = d.y; dirToLight[2] = d.z; power[0] = p.x; power[1] = p.y; power[2] = p.z;
}

// ---- hooks for oracle/ref/ref_loop.cpp: the reference's rp_main.rgen / .chit / .miss compiled as C++ call back here for what the
// reference gets from the Vulkan driver and the MDL code generator -- the scene as the host packs it, ray queries, the closed-form
// BSDF / EDF entry points.  Everything else (sample loop, camera, bounce loop, NEE, Russian roulette, volumes, accumulation) is the
// reference's own text there, and tests/test_oracle_ref_loop.py compares its images with orc_render's.
struct OrcHook {
  Prepared P; Frame F; OrcCamera cam; OrcSettings rs; OrcRegion rg;
  std::vector<std::vector<float>> packedVerts;                 // per mesh: 8 floats per vertex, rp_main.h:58-64
  std::vector<float> lights[4];                                // 12 floats per light, rp_main.h:73-113
};
void* orc_hook_open(const OrcScene* scene, const OrcCamera* camera, const OrcSettings* settings, const OrcRegion* region)
{
  OrcHook* H = new OrcHook(); H->cam = *camera; H->rs = *settings; H->rg = *region;
  prepare(scene, H->P); make_frame(H->F, H->P, &H->cam, &H->rs, &H->rg);
  for (const MeshData& m : H->P.meshes) {
    std::vector<float> pv(m.verts.size() * 8);
    for (size_t i = 0; i < m.verts.size(); i++) { const FVertex& v = m.verts[i]; float* f = &pv[8 * i]; f[0] = v.pos.x; f[1] = v.pos.y; f[2] = v.pos.z; f[3] = v.bsign; memcpy(&f[4], &v.n, 4); memcpy(&f[5], &v.t, 4); f[6] = v.u; f[7] = v.v; }
    H->packedVerts.push_back(std::move(pv));
  }
  auto put3 = [](float* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
  auto putu = [](float* d, uint32_t u) { memcpy(d, &u, 4); };
  for (const SphereL& l : H->P.sphere) { float f[12] = {0}; put3(f, l.pos); putu(f + 3, l.ds); put3(f + 4, l.em); f[7] = l.area; put3(f + 8, l.radius); H->lights[0].insert(H->lights[0].end(), f, f + 12); }
  for (const DistantL& l : H->P.distant) { float f[12] = {0}; put3(f, l.dir); f[3] = l.angle; put3(f + 4, l.em); putu(f + 7, l.ds); f[11] = l.invPdf; H->lights[1].insert(H->lights[1].end(), f, f + 12); }
  for (const RectL& l : H->P.rect) { float f[12] = {0}; put3(f, l.origin); f[3] = l.width; put3(f + 4, l.em); f[7] = l.height; putu(f + 8, l.t0); putu(f + 9, l.t1); putu(f + 10, l.ds); H->lights[2].insert(H->lights[2].end(), f, f + 12); }
  for (const DiskL& l : H->P.disk) { float f[12] = {0}; put3(f, l.origin); f[3] = l.rx; put3(f + 4, l.em); f[7] = l.ry; putu(f + 8, l.t0); putu(f + 9, l.t1); putu(f + 10, l.ds); H->lights[3].insert(H->lights[3].end(), f, f + 12); }
  return H;
}
void orc_hook_close(void* h) { delete (OrcHook*)h; }
// out: background rgb (fallback dome texel), lens radius, clip range packed (as uint bits), exposure
void orc_hook_frame(void* h, float* out) { OrcHook* H = (OrcHook*)h; out[0] = H->F.background.x; out[1] = H->F.background.y; out[2] = H->F.background.z; out[3] = H->F.lensRadius; uint32_t cr = pack_half2x16(H->cam.clipStart, H->cam.clipEnd); memcpy(&out[4], &cr, 4); }
int orc_hook_trace(void* h, const float* o, const float* d, float tMin, float tMax, int anyHit, uint32_t rng, float* tuv, uint32_t* instPrim)
{
  OrcHook* H = (OrcHook*)h;
  if (anyHit) return trace_any(H->P, v3(o), v3(d), tMin, tMax, rng) ? 1 : 0;
  Hit hit;
  if (!trace_closest(H->P, v3(o), v3(d), tMin, tMax, hit, rng)) return 0;
  tuv[0] = hit.t; tuv[1] = hit.u; tuv[2] = hit.v; instPrim[0] = H->P.tris[hit.tri].instance; instPrim[1] = H->P.tris[hit.tri].prim;
  return 1;
}
void orc_hook_instance(void* h, uint32_t inst, float* o2w, float* w2o, int32_t* info /* mesh, flags, material, objectId, instanceId */)
{
  OrcHook* H = (OrcHook*)h; const Instance& I = H->P.instances[inst]; const MeshData& m = H->P.meshes[I.mesh];
  memcpy(o2w, I.o2w, sizeof(I.o2w)); memcpy(w2o, I.w2o, sizeof(I.w2o));
  info[0] = (int32_t)I.mesh; info[1] = (int32_t)m.flags; info[2] = m.material; info[3] = m.objectId; info[4] = I.instanceId;
}
const float* orc_hook_mesh_vertices(void* h, uint32_t mesh) { return ((OrcHook*)h)->packedVerts[mesh].data(); }
const uint32_t* orc_hook_mesh_faces(void* h, uint32_t mesh) { return ((OrcHook*)h)->P.meshes[mesh].faces; }
void orc_hook_lights(void* h, uint32_t* counts, const float** ptrs) { OrcHook* H = (OrcHook*)h; for (int i = 0; i < 4; i++) { counts[i] = (uint32_t)(H->lights[i].size() / 12); ptrs[i] = H->lights[i].data(); } }
// out: klass, emission rgb, sigma_a rgb, sigma_s rgb, ior, anisotropy, thinWalled
void orc_hook_material(void* h, uint32_t mat, float* out)
{
  OrcHook* H = (OrcHook*)h; const OrcMaterial& m = H->P.materials[mat];
  for (int i = 0; i < 13; i++) out[i] = 0.0f;
  out[0] = (float)m.klass; out[1] = m.p[ORC_P_EMISSION]; out[2] = m.p[ORC_P_EMISSION + 1]; out[3] = m.p[ORC_P_EMISSION + 2]; out[10] = 1.0f;
  if (m.klass == ORC_MAT_OPEN_PBR) { OpbrParams o = opbr_params(m); out[4] = o.sigmaA.x; out[5] = o.sigmaA.y; out[6] = o.sigmaA.z; out[7] = o.sigmaS.x; out[8] = o.sigmaS.y; out[9] = o.sigmaS.z; out[10] = o.eta; out[11] = o.anisotropy; out[12] = o.thinWalled ? 1.0f : 0.0f; }
}
static inline void hook_state(State& st, const float* frame, float ior1, float ior2, int thin)
{ st.normal = v3(frame); st.tangentU = v3(frame + 3); st.tangentV = v3(frame + 6); st.geomNormal = v3(frame + 9); st.position = v3(0, 0, 0); st.u = st.v = 0.0f; st.frontFace = true; st.ior1 = ior1; st.ior2 = ior2; st.thinWalled = thin != 0; }
void orc_hook_bsdf_sample(void* h, uint32_t mat, const float* frame, const float* k1, const float* xi, float ior1, float ior2, int thin, float* out)
{
  State st; hook_state(st, frame, ior1, ior2, thin);
  BsdfSample bs; bsdf_sample(((OrcHook*)h)->P.materials[mat], st, v3(k1), xi, bs);
  out[0] = bs.k2.x; out[1] = bs.k2.y; out[2] = bs.k2.z; out[3] = bs.overPdf.x; out[4] = bs.overPdf.y; out[5] = bs.overPdf.z; out[6] = bs.pdf; uint32_t e = bs.event; memcpy(&out[7], &e, 4);
}
void orc_hook_bsdf_evaluate(void* h, uint32_t mat, const float* frame, const float* k1, const float* k2, float ior1, float ior2, int thin, float* out)
{
  State st; hook_state(st, frame, ior1, ior2, thin);
  BsdfEval ev; bsdf_evaluate(((OrcHook*)h)->P.materials[mat], st, v3(k1), v3(k2), ev);
  out[0] = ev.diffuse.x; out[1] = ev.diffuse.y; out[2] = ev.diffuse.z; out[3] = ev.glossy.x; out[4] = ev.glossy.y; out[5] = ev.glossy.z; out[6] = ev.pdf;
}
// dome light of the scene: out = rotation quaternion (x, y, z, w), emission multiplier; returns 0 when only the fallback dome exists
int orc_hook_dome(void* h, float* out)
{
  const OrcDomeLight* d = ((OrcHook*)h)->P.dome;
  if (!d) return 0;
  for (int i = 0; i < 4; i++) out[i] = d->rotation[i];
  for (int i = 0; i < 3; i++) out[4 + i] = d->baseEmission[i];
  return 1;
}
void orc_hook_dome_lookup(void* h, float u, float v, float* rgb)
{
  const Prepared& P = ((OrcHook*)h)->P;
  F4v t = sample_bilinear_repeat(P.textures[P.dome->texture], u, v);
  rgb[0] = t.x; rgb[1] = t.y; rgb[2] = t.z;
}
void orc_hook_bsdf_albedo(void* h, uint32_t mat, const float* frame, const float* k1, float ior1, float ior2, int thin, float* out)
{
  State st; hook_state(st, frame, ior1, ior2, thin);
  V3 a = bsdf_albedo(((OrcHook*)h)->P.materials[mat], st, v3(k1)); out[0] = a.x; out[1] = a.y; out[2] = a.z;
}
// face ids packed as the reference packs them behind the BLAS payload preamble (Gi.cpp:878-905): stride bytes per face, read as 32-bit words
const int32_t* orc_hook_mesh_face_ids(void* h, uint32_t mesh, uint32_t* stride)
{ const MeshData& m = ((OrcHook*)h)->P.meshes[mesh]; *stride = m.faceIdStride; return (const int32_t*)m.faceIdData.data(); }
void orc_hook_edf_factor(void* h, uint32_t mat, float c, float* out)
{
  const OrcMaterial& m = ((OrcHook*)h)->P.materials[mat]; V3 f = v3(1, 1, 1);
  if (m.klass == ORC_MAT_OPEN_PBR) { float cior = m.p[ORC_P_COAT_IOR], qc = (cior - 1.0f) / (cior + 1.0f); f = opbr_emission_factor(m.p[ORC_P_CLEARCOAT], v3(m.p + ORC_P_COAT_COLOR), qc * qc, c); }
  out[0] = f.x; out[1] = f.y; out[2] = f.z;
}

} // extern "C"
