// gi_device_math.h -- device-side arithmetic of the render loop (gfx950).
//
// "Arithmetic contract" (DESIGN.md): fp32, no FMA contraction (this translation unit is built with
// -ffp-contract=off), IEEE-rounded + - * / sqrt (hipcc's default correctly-rounded divide/sqrt), and two
// polynomial transcendentals (gi_sincos2pi, gi_logf).  Under that contract every path the kernels compute is
// reproducible on a CPU; tests/ hold the kernels to the oracle bit-for-bit.
//
// Restates /root/reference/src/gi/shaders/common.glsl (RNG :44-47,74-124; ONB :128-137; ray offset :143-162;
// octahedral codec :181-207; sampling maps :210-252; luminance :254-257).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gi {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 v3(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// Square root, correctly rounded like sqrtf (the oracle's), in fewer instructions for ordinary arguments.  The compiler expands sqrtf into 16 VALU instructions: scale
// by 2^32 below 2^-96, v_sqrt_f32 (1 ulp), a +- 1 ulp correction by two residuals, unscale, and a class test that returns +-0 and +inf unchanged.  For x in
// [2^-95, +inf) the scaling and the class test select nothing, and what is left -- taken here literally -- is 9 instructions after a two-instruction range test; any
// other argument (zeros, denormals, negatives, NaN, +inf) goes to sqrtf.  Same bits by construction; tests/test_gpu_parity.py runs all 2^32 arguments through both.
// Measured on one box, libraries alternating (profiles/r06zp_lean_sqrt_variants.txt): `k_path` 201.1 / 201.3 -> 199.8 / 199.6 ms per C2 launch at spp 1024 (it is
// bound by its VALU issue rate and held 12 roots in 2 407 instructions); the shade stage C3 9.3 -> 9.15 ms but C5 26.9 -> 27.75 and C4 3.3 -> 3.5 (the second copy of
// every root and its branch cost the UsdPreviewSurface kernel more than the five instructions save): the lean form is compiled into the fused kernels only.
__device__ __forceinline__ float gi_sqrt(float x)
{
#if !defined(GI_LEAN_SQRT)   // a translation unit asks for the lean form before it includes this header: the fused kernels do (gi_path.hip), the stage kernels do not
  return sqrtf(x);
#else
  if (__builtin_expect((__builtin_bit_cast(uint32_t, x) - 0x10000000u) < (0x7f800000u - 0x10000000u), 1)) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float lo = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) - 1u), hi = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) + 1u);
    const float rlo = __builtin_fmaf(-lo, s, x), rhi = __builtin_fmaf(-hi, s, x);
    s = (rlo <= 0.0f) ? lo : s;
    s = (rhi > 0.0f) ? hi : s;
    return s;
  }
  return sqrtf(x);
#endif
}
__device__ __forceinline__ float length(V3 a) { return gi_sqrt(dot(a, a)); }
__device__ __forceinline__ V3 normalize(V3 a) { float inv = 1.0f / gi_sqrt(dot(a, a)); return a * inv; }
__device__ __forceinline__ float fmax2(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float fmin2(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

#define GI_PI 3.1415926535897932384626433832795f
#define GI_FLT_MAX 3.402823466e38f
#define GI_FLT_MIN 1.175494351e-38f

// sin/cos(2*pi*x), x in [0,1]: exact octant reduction + Cephes single-precision kernels, plain mul/add.
__device__ __forceinline__ void gi_sincos2pi(float x, float* s, float* c)
{
  float y = x * 8.0f;
  int q = (int)y;
  int j = (q + 1) >> 1;
  float z = y - (float)(2 * j);
  float t = z * 0.78539816339744830962f;
  float t2 = t * t;
  float sp = ((-1.9515295891e-4f * t2 + 8.3321608736e-3f) * t2 - 1.6666654611e-1f) * t2 * t + t;
  float cp = ((2.443315711809948e-5f * t2 - 1.388731625493765e-3f) * t2 + 4.166664568298827e-2f) * t2 * t2 - 0.5f * t2 + 1.0f;
  int k = j & 3;
  float ss = (k & 1) ? cp : sp;
  float cc = (k & 1) ? sp : cp;
  *s = (k & 2) ? -ss : ss;
  *c = (k == 1 || k == 2) ? -cc : cc;
}

__device__ __forceinline__ void gi_sincosr(float a, float* s, float* c)
{
  float r = a * 0.15915494309189533577f;
  float f = r - floorf(r);
  gi_sincos2pi(f, s, c);
}

__device__ __forceinline__ float gi_logf(float x)
{
  uint32_t b = f2u(x);
  int e = 0;
  if (b < 0x00800000u) { x = x * 16777216.0f; b = f2u(x); e = -24; }
  e += (int)((b >> 23) & 0xffu) - 126;
  float m = u2f((b & 0x007fffffu) | 0x3f000000u);
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

// exp(x), x <= 0 (Beer-Lambert transmittance): Cephes expf kernel, plain mul/add
__device__ __forceinline__ float gi_expf(float x)
{
  if (x < -87.0f) return 0.0f;
  if (x > 0.0f) x = 0.0f;
  float fx = floorf(x * 1.44269504088896341f + 0.5f);
  x = x - fx * 0.693359375f;
  x = x - fx * -2.12194440e-4f;
  float z = x * x;
  float y = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z
            + x + 1.0f;
  return y * u2f((uint32_t)((int)fx + 127) << 23);
}

// atan2 / acos for the equirectangular dome lookup (rp_main.miss:46-53): Cephes atanf / asinf kernels, plain mul/add
__device__ __forceinline__ float gi_atanf(float xx)
{
  float x = fabsf(xx), y;
  if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
  else y = 0.0f;
  float z = x * x;
  y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
  return xx < 0.0f ? -y : y;
}
__device__ __forceinline__ float gi_atan2f(float y, float x)
{
  if (x == 0.0f) return y > 0.0f ? 1.5707963267948966f : (y < 0.0f ? -1.5707963267948966f : 0.0f);
  float z = gi_atanf(y / x);
  if (x < 0.0f) z = z + (y >= 0.0f ? 3.14159265358979323846f : -3.14159265358979323846f);
  return z;
}
__device__ __forceinline__ float gi_asinf(float xx)
{
  float a = fabsf(xx), x, z; bool flag = false;
  if (a > 0.5f) { z = 0.5f * (1.0f - a); x = gi_sqrt(z); flag = true; }
  else { x = a; z = x * x; }
  z = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
  if (flag) { z = z + z; z = 1.5707963267948966f - z; }
  return xx < 0.0f ? -z : z;
}
__device__ __forceinline__ float gi_acosf(float x)
{
  x = fmin2(fmax2(x, -1.0f), 1.0f);
  if (x < -0.5f) return 3.14159265358979323846f - 2.0f * gi_asinf(gi_sqrt(0.5f * (1.0f + x)));
  if (x > 0.5f) return 2.0f * gi_asinf(gi_sqrt(0.5f * (1.0f - x)));
  return 1.5707963267948966f - gi_asinf(x);
}

// half -> float (exact), for the packed diffuse/specular light multipliers (rp_main.chit:431)
__device__ __forceinline__ float gi_half_to_float(uint32_t h)
{
  uint32_t sign = (h & 0x8000u) << 16;
  uint32_t ex = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  if (ex == 0) { float v = (float)man * 5.9604644775390625e-8f; return sign ? -v : v; }
  if (ex == 31) return u2f(sign | 0x7f800000u | (man << 13));
  return u2f(sign | ((ex + 112u) << 23) | (man << 13));
}

// ---- RNG (common.glsl:74-124) ----
__device__ __forceinline__ uint32_t gi_hash_init(uint32_t x)
{
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0xd35a2d97u; x ^= x >> 15;
  return x;
}
__device__ __forceinline__ float gi_next1f(uint32_t& state)
{
  uint32_t s = state * 747796405u + 2891336453u;
  uint32_t word = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  state = (word >> 22) ^ word; // the output word replaces the LCG state (common.glsl:92-96)
  return u2f(0x3f800000u | (state >> 9)) - 1.0f;
}

// ---- common.glsl:128-137 ----
__device__ __forceinline__ void gi_orthonormal_basis(V3 n, V3& b1, V3& b2)
{
  float nsign = (n.z >= 0.0f) ? 1.0f : -1.0f;
  float a = -1.0f / (nsign + n.z);
  float b = n.x * n.y * a;
  b1 = v3(1.0f + nsign * n.x * n.x * a, nsign * b, -nsign * n.x);
  b2 = v3(b, nsign + n.y * n.y * a, -n.y);
}

// ---- common.glsl:143-162 ----
__device__ __forceinline__ float gi_offset_component(float p, float n)
{
  int io = (int)(n * 64.0f);
  int pi = (int)f2u(p);
  int moved = pi + ((p >= 0.0f) ? io : -io);
  float ip = u2f((uint32_t)moved);
  float fp = p + n * (1.0f / 65536.0f);
  return (fabsf(p) >= (1.0f / 32.0f)) ? ip : fp;
}
__device__ __forceinline__ V3 gi_offset_ray_origin(V3 p, V3 n)
{
  return v3(gi_offset_component(p.x, n.x), gi_offset_component(p.y, n.y), gi_offset_component(p.z, n.z));
}

// ---- common.glsl:181-207 ----
__device__ __forceinline__ V3 gi_decode_direction(uint32_t e)
{
  float ex = (float)(e & 0xffffu) / 65535.0f, ey = (float)(e >> 16) / 65535.0f;
  ex = ex * 2.0f - 1.0f; ey = ey * 2.0f - 1.0f;
  V3 v = v3(ex, ey, 1.0f - fabsf(ex) - fabsf(ey));
  float t = fmax2(-v.z, 0.0f);
  v.x += (v.x >= 0.0f) ? -t : t;
  v.y += (v.y >= 0.0f) ? -t : t;
  return normalize(v);
}

// colormap_inferno (colormap.glsl:42-53), Horner form with plain mul/add
__device__ __forceinline__ V3 gi_colormap_inferno(float t)
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

// colormap_viridis (colormap.glsl:3-14)
__device__ __forceinline__ V3 gi_colormap_viridis(float t)
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

__device__ __forceinline__ float gi_luminance(V3 c) { return dot(c, v3(0.2126f, 0.7152f, 0.0722f)); }
__device__ __forceinline__ float gi_safe_div(float a, float b) { return (b == 0.0f) ? 0.0f : (a / b); }
__device__ __forceinline__ V3 gi_safe_div(V3 v, float f) { return (f == 0.0f) ? v3(0.0f, 0.0f, 0.0f) : (v / f); }

// ---- common.glsl:210-252 ----
__device__ __forceinline__ V3 gi_sample_hemisphere(float x0, float x1)
{
  float a = gi_sqrt(x0);
  float s, c; gi_sincos2pi(x1, &s, &c);
  return v3(a * c, a * s, gi_sqrt(1.0f - x0));
}
__device__ __forceinline__ V3 gi_sample_sphere(float x0, float x1, V3 radius)
{
  float a = 1.0f - 2.0f * x0;
  float b = gi_sqrt(1.0f - a * a);
  float s, c; gi_sincos2pi(x1, &s, &c);
  return v3(b * c, b * s, a) * radius;
}
__device__ __forceinline__ void gi_sample_disk(float x0, float x1, float rx, float ry, float& ox, float& oy)
{
  float a = 2.0f * x0 - 1.0f, b = 2.0f * x1 - 1.0f;
  float r0, r1, phi;
  if ((a * a) > (b * b)) { r0 = rx * a; r1 = ry * a; phi = (GI_PI / 4.0f) * (b / a); }
  else { r0 = rx * b; r1 = ry * b; phi = (GI_PI / 2.0f) - (GI_PI / 4.0f) * gi_safe_div(a, b); }
  float s, c; gi_sincosr(phi, &s, &c);
  ox = r0 * c; oy = r1 * s;
}

// ---- rp_main.rgen:118-130 ----
__device__ __forceinline__ void gi_fis_gauss(float x0, float x1, float& ox, float& oy)
{
  float u1 = fmax2(1e-38f, x0);
  float r = 0.375f * gi_sqrt(-2.0f * gi_logf(u1));
  float s, c; gi_sincos2pi(x1, &s, &c);
  ox = c * r; oy = s * r;
}

} // namespace gi
