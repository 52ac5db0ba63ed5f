// glsl_compat.h -- just enough of GLSL, as C++17, to compile the pure functions of the reference's shader sources
// (/root/reference/src/gi/shaders/common.glsl, colormap.glsl, rp_main_payload.glsl and a few functions cut out of rp_main.rgen /
// mdl_interface.glsl by line range) WHERE THEY LIE, so that the oracle's restatements can be checked against the reference's own code
// (oracle/ref/build_ref.py -> oracle/_ref/libgi_ref.so; tests/test_oracle_ref.py).  Test infrastructure only; ours, not the reference's.
//
// GLSL computes in fp32 and its unsuffixed literals are floats; C++ would promote `x * 2.0` to double.  The translation unit therefore
// compiles the GLSL text with `float` redefined to the class Float below: a float that accepts double literals by rounding them to fp32
// first and only has fp32 operators, so every operation rounds exactly where the shader's would (no FMA contraction: -ffp-contract=off).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

namespace glsl {

typedef unsigned int uint;

struct Float {
  float v;
  Float() = default;
  template <typename T, typename = std::enable_if_t<std::is_arithmetic<T>::value>> Float(T x) : v((float)x) {}
  explicit operator float() const { return v; }
  explicit operator int() const { return (int)v; }
  explicit operator uint() const { return (uint)v; }
  Float operator-() const { return Float(-v); }
  Float& operator+=(Float o) { v = v + o.v; return *this; }
  Float& operator-=(Float o) { v = v - o.v; return *this; }
  Float& operator*=(Float o) { v = v * o.v; return *this; }
  Float& operator/=(Float o) { v = v / o.v; return *this; }
};
#define GLSL_FLOAT_BINOP(op) \
  inline Float operator op(Float a, Float b) { return Float(a.v op b.v); } \
  template <typename T, typename = std::enable_if_t<std::is_arithmetic<T>::value>> inline Float operator op(Float a, T b) { return Float(a.v op (float)b); } \
  template <typename T, typename = std::enable_if_t<std::is_arithmetic<T>::value>> inline Float operator op(T a, Float b) { return Float((float)a op b.v); }
GLSL_FLOAT_BINOP(+) GLSL_FLOAT_BINOP(-) GLSL_FLOAT_BINOP(*) GLSL_FLOAT_BINOP(/)
#define GLSL_FLOAT_CMP(op) \
  inline bool operator op(Float a, Float b) { return a.v op b.v; } \
  template <typename T, typename = std::enable_if_t<std::is_arithmetic<T>::value>> inline bool operator op(Float a, T b) { return a.v op (float)b; } \
  template <typename T, typename = std::enable_if_t<std::is_arithmetic<T>::value>> inline bool operator op(T a, Float b) { return (float)a op b.v; }
GLSL_FLOAT_CMP(<) GLSL_FLOAT_CMP(>) GLSL_FLOAT_CMP(<=) GLSL_FLOAT_CMP(>=) GLSL_FLOAT_CMP(==) GLSL_FLOAT_CMP(!=)

inline Float sqrt(Float a) { return Float(sqrtf(a.v)); }
inline Float cos(Float a) { return Float(cosf(a.v)); }   // libm, not the GPU's: callers compare with a tolerance
inline Float sin(Float a) { return Float(sinf(a.v)); }
inline Float log(Float a) { return Float(logf(a.v)); }
inline Float exp(Float a) { return Float(expf(a.v)); }
inline Float exp2(Float a) { return Float(exp2f(a.v)); }
inline Float tan(Float a) { return Float(tanf(a.v)); }
inline Float atan(Float y, Float x) { return Float(atan2f(y.v, x.v)); }
inline Float acos(Float a) { return Float(acosf(a.v)); }
inline Float abs(Float a) { return Float(fabsf(a.v)); }
inline Float floor(Float a) { return Float(floorf(a.v)); }
inline Float max(Float a, Float b) { return a.v < b.v ? b : a; } // GLSL: y if x < y else x
inline Float min(Float a, Float b) { return b.v < a.v ? b : a; }
inline Float clamp(Float x, Float lo, Float hi) { return min(max(x, lo), hi); }
inline Float mix(Float a, Float b, Float t) { return a * (Float(1.0f) - t) + b * t; }
inline uint max(uint a, uint b) { return a < b ? b : a; }
inline uint min(uint a, uint b) { return b < a ? b : a; }
inline uint max(int a, uint b) { return max((uint)a, b); }   // max(1, MEDIUM_STACK_SIZE) with an unsigned define
inline int max(int a, int b) { return a < b ? b : a; }
inline int min(int a, int b) { return b < a ? b : a; }
inline uint min(uint a, int b) { return min(a, (uint)b); }

// Swizzles: C++ has no `v.yx` data members over constructor-bearing types, so the build recipe rewrites the handful of swizzles the
// compiled functions use (.xy .yx .zw .xyz .r .g .b, all reads) into calls of the member functions below.
struct vec2 {
  Float x, y;
  vec2() : x(0.0f), y(0.0f) {}
  vec2(Float a) : x(a), y(a) {}
  vec2(Float a, Float b) : x(a), y(b) {}
  vec2 operator-() const { return vec2(-x, -y); }
  vec2 xy() const { return vec2(x, y); }
  vec2 yx() const { return vec2(y, x); }
};
struct bvec3 { bool x, y, z; };
inline bool any(bvec3 b) { return b.x || b.y || b.z; }
struct ivec3 {
  int x, y, z;
  ivec3() : x(0), y(0), z(0) {}
  ivec3(int a, int b, int c) : x(a), y(b), z(c) {}
  explicit ivec3(const struct vec3& v);
  ivec3 operator-() const { return ivec3(-x, -y, -z); }
};
inline ivec3 operator+(ivec3 a, ivec3 b) { return ivec3(a.x + b.x, a.y + b.y, a.z + b.z); }
struct vec3 {
  Float x, y, z;
  vec3() : x(0.0f), y(0.0f), z(0.0f) {}
  vec3(Float a) : x(a), y(a), z(a) {}
  vec3(Float a, Float b_, Float c) : x(a), y(b_), z(c) {}
  vec3(const vec2& a, Float c) : x(a.x), y(a.y), z(c) {}
  explicit vec3(bvec3 b) : x(b.x ? 1.0f : 0.0f), y(b.y ? 1.0f : 0.0f), z(b.z ? 1.0f : 0.0f) {}
  vec3 operator-() const { return vec3(-x, -y, -z); }
  vec3& operator/=(Float s) { x /= s; y /= s; z /= s; return *this; }
  vec3& operator*=(Float s) { x *= s; y *= s; z *= s; return *this; }
  vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
  vec3& operator*=(const vec3& o) { x *= o.x; y *= o.y; z *= o.z; return *this; }
  vec3 rgb() const { return *this; }
  vec3 xyz() const { return *this; }
  vec2 xy() const { return vec2(x, y); }
  vec2 yx() const { return vec2(y, x); }
  Float r() const { return x; }
  Float g() const { return y; }
  Float b() const { return z; }
};
inline ivec3::ivec3(const vec3& v) : x((int)v.x.v), y((int)v.y.v), z((int)v.z.v) {}
struct vec4 {
  Float x, y, z, w;
  vec4() : x(0.0f), y(0.0f), z(0.0f), w(0.0f) {}
  vec4(Float a) : x(a), y(a), z(a), w(a) {}
  vec4(Float a, Float b, Float c, Float d) : x(a), y(b), z(c), w(d) {}
  vec4(const vec3& v, Float d) : x(v.x), y(v.y), z(v.z), w(d) {}
  vec2 xy() const { return vec2(x, y); }
  vec2 zw() const { return vec2(z, w); }
  vec3 xyz() const { return vec3(x, y, z); }
  vec3 rgb() const { return vec3(x, y, z); }
};
struct ivec4 { int x, y, z, w; ivec4() : x(0), y(0), z(0), w(0) {} ivec4(int a, int b, int c, int d) : x(a), y(b), z(c), w(d) {} };
struct ivec2 { int x, y; ivec2() : x(0), y(0) {} ivec2(int a, int b) : x(a), y(b) {} };
// Matrices, column-major as in GLSL; products accumulate left to right (GLSL leaves the order to the driver; this is the oracle's order too,
// so comparisons through these test the STRUCTURE of a computation -- which transform, which transpose -- not the driver's rounding)
struct mat4x3 { vec3 c[4]; };                       // 4 columns of vec3 (gl_ObjectToWorldEXT / gl_WorldToObjectEXT)
struct mat3 { vec3 c[3]; mat3() {} explicit mat3(const mat4x3& m) { c[0] = m.c[0]; c[1] = m.c[1]; c[2] = m.c[2]; } };
inline vec4 make_vec4(const vec3& v, Float w) { return vec4(v.x, v.y, v.z, w); }
struct uvec2 { uint x, y; };
struct uvec3 {
  uint x, y, z;
  uvec3() : x(0), y(0), z(0) {}
  explicit uvec3(uint a) : x(a), y(a), z(a) {}
  uvec3(uint a, uint b, uint c) : x(a), y(b), z(c) {}
  uvec2 xy() const { return uvec2{x, y}; }
  uint& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  uint operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline uvec3 operator*(const uvec3& a, uint b) { return uvec3(a.x * b, a.y * b, a.z * b); }
inline uvec3 operator*(const uvec3& a, int b) { return a * (uint)b; }
inline uvec3 operator+(const uvec3& a, int b) { return uvec3(a.x + (uint)b, a.y + (uint)b, a.z + (uint)b); }
struct uvec4 {
  uint x, y, z, w;
  uvec4() : x(0), y(0), z(0), w(0) {}
  uvec4(uint a, uint b, uint c, uint d) : x(a), y(b), z(c), w(d) {}
  uvec4& operator>>=(int s) { x >>= s; y >>= s; z >>= s; w >>= s; return *this; }
  uvec4& operator|=(uint m) { x |= m; y |= m; z |= m; w |= m; return *this; }
};

#define GLSL_VEC_BINOP(V, op) \
  inline V operator op(const V& a, const V& b); \
  inline V operator op(const V& a, Float b) { return a op V(b); } \
  inline V operator op(Float a, const V& b) { return V(a) op b; } \
  template <typename T, typename = std::enable_if_t<std::is_arithmetic<T>::value>> inline V operator op(const V& a, T b) { return a op V(Float(b)); } \
  template <typename T, typename = std::enable_if_t<std::is_arithmetic<T>::value>> inline V operator op(T a, const V& b) { return V(Float(a)) op b; }
#define GLSL_VEC2_DEF(op) GLSL_VEC_BINOP(vec2, op) inline vec2 operator op(const vec2& a, const vec2& b) { return vec2(a.x op b.x, a.y op b.y); }
#define GLSL_VEC3_DEF(op) GLSL_VEC_BINOP(vec3, op) inline vec3 operator op(const vec3& a, const vec3& b) { return vec3(a.x op b.x, a.y op b.y, a.z op b.z); }
#define GLSL_VEC4_DEF(op) GLSL_VEC_BINOP(vec4, op) inline vec4 operator op(const vec4& a, const vec4& b) { return vec4(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w); }
GLSL_VEC2_DEF(+) GLSL_VEC2_DEF(-) GLSL_VEC2_DEF(*) GLSL_VEC2_DEF(/)
GLSL_VEC3_DEF(+) GLSL_VEC3_DEF(-) GLSL_VEC3_DEF(*) GLSL_VEC3_DEF(/)
GLSL_VEC4_DEF(+) GLSL_VEC4_DEF(-) GLSL_VEC4_DEF(*) GLSL_VEC4_DEF(/)

inline vec2 abs(const vec2& a) { return vec2(abs(a.x), abs(a.y)); }
inline vec3 abs(const vec3& a) { return vec3(abs(a.x), abs(a.y), abs(a.z)); }
inline vec3 exp(const vec3& a) { return vec3(exp(a.x), exp(a.y), exp(a.z)); }
inline Float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Float length(const vec3& a) { return sqrt(dot(a, a)); }
inline vec3 normalize(const vec3& a) { return a * (Float(1.0f) / sqrt(dot(a, a))); } // GLSL: x * inversesqrt(dot(x, x)); driver-defined ulp
inline vec3 reflect(const vec3& i, const vec3& n) { return i - Float(2.0f) * dot(n, i) * n; }
inline vec3 cross(const vec3& a, const vec3& b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline vec3 operator*(const mat4x3& m, const vec4& v) { return ((m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z) + m.c[3] * v.w; }
inline vec3 operator*(const vec3& v, const mat3& m) { return vec3(dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2])); }
inline bvec3 greaterThan(const vec3& a, const vec3& b) { return bvec3{a.x > b.x, a.y > b.y, a.z > b.z}; }
inline bvec3 equal(const vec3& a, const vec3& b) { return bvec3{a.x == b.x, a.y == b.y, a.z == b.z}; }
inline vec3 max(const vec3& a, const vec3& b) { return vec3(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
inline bvec3 greaterThanEqual(const vec3& a, const vec3& b) { return bvec3{a.x >= b.x, a.y >= b.y, a.z >= b.z}; }
struct bvec2 { bool x, y; };
inline bvec2 notEqual(const vec2& a, const vec2& b) { return bvec2{a.x != b.x, a.y != b.y}; }
inline bool all(bvec2 b) { return b.x && b.y; }
inline vec3 mix(const vec3& a, const vec3& b, bvec3 s) { return vec3(s.x ? b.x : a.x, s.y ? b.y : a.y, s.z ? b.z : a.z); }
inline ivec3 mix(const ivec3& a, const ivec3& b, bvec3 s) { return ivec3(s.x ? b.x : a.x, s.y ? b.y : a.y, s.z ? b.z : a.z); }
inline int f2i(Float f) { int i; memcpy(&i, &f.v, 4); return i; }
inline Float i2f(int i) { float f; memcpy(&f, &i, 4); return Float(f); }
inline ivec3 floatBitsToInt(const vec3& v) { return ivec3(f2i(v.x), f2i(v.y), f2i(v.z)); }
inline vec3 intBitsToFloat(const ivec3& v) { return vec3(i2f(v.x), i2f(v.y), i2f(v.z)); }
inline Float uintBitsToFloat(uint u) { float f; memcpy(&f, &u, 4); return Float(f); }
inline uint floatBitsToUint(Float f) { uint u; memcpy(&u, &f.v, 4); return u; }
inline vec4 uintBitsToFloat(const uvec4& u) { return vec4(uintBitsToFloat(u.x), uintBitsToFloat(u.y), uintBitsToFloat(u.z), uintBitsToFloat(u.w)); }
// unpackHalf2x16 (GLSL 4.60 section 8.4): two IEEE half floats, first component in the low bits
inline Float half_to_float(uint h)
{
  const uint sgn = (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  uint o;
  if (e == 0u) { if (m == 0u) o = sgn; else { int ee = -1; uint mm = m; do { ee++; mm <<= 1; } while (!(mm & 0x400u)); o = sgn | ((uint)(127 - 15 - ee) << 23) | ((mm & 0x3ffu) << 13); } }
  else if (e == 31u) o = sgn | 0x7f800000u | (m << 13);
  else o = sgn | ((e + 112u) << 23) | (m << 13);
  return uintBitsToFloat(o);
}
inline vec2 unpackHalf2x16(uint p) { return vec2(half_to_float(p & 0xffffu), half_to_float(p >> 16)); }
// GLSL 4.60 section 8.4: packUnorm2x16: round(clamp(c, 0, 1) * 65535.0), first component in the low bits; unpack: f / 65535.0
inline uint packUnorm2x16(const vec2& v) { const uint a = (uint)nearbyintf(clamp(v.x, 0.0f, 1.0f).v * 65535.0f), b = (uint)nearbyintf(clamp(v.y, 0.0f, 1.0f).v * 65535.0f); return a | (b << 16); }
inline vec2 unpackUnorm2x16(uint p) { return vec2(Float((float)(p & 0xffffu)) / Float(65535.0f), Float((float)(p >> 16)) / Float(65535.0f)); }

} // namespace glsl
