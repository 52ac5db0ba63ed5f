// ref_shim.cpp -- C entry points over the reference's OWN shader functions, compiled from /root/reference through oracle/ref/glsl_compat.h
// (recipe: oracle/ref/build_ref.py; output oracle/_ref/libgi_ref.so).  Test infrastructure: tests/test_oracle_ref.py compares the oracle's
// restatements with these.  The files included below from oracle/_ref/gen/ are generated from the reference sources at build time.
#include "glsl_compat.h"
extern "C" void orc_dbg_sample_bilinear(const void* orcTexture, float u, float v, float* out4); // oracle/gi_oracle.cpp
extern "C" void orc_dbg_sample_trilinear(const float* rgba, uint32_t w, uint32_t h, uint32_t d, float u, float v, float ww, float* out4);

#include <cstdint>
#undef UINT32_MAX // common.glsl declares a constant of this name

namespace ref {
using namespace glsl;
// ---- the GLSL side of interface/gtl.h (the C++ side needs glm): type names as the shaders see them
#define GI_INT int
#define GI_UINT uint
#define GI_UINT64 uint64_t
#define GI_FLOAT float
#define GI_VEC2 vec2
#define GI_VEC3 vec3
#define GI_VEC4 vec4
#define GI_UVEC2 uvec2
#define GI_UVEC4 uvec4
#define GI_INTERFACE_BEGIN(NAME)
#define GI_INTERFACE_END()
#define GI_BINDING_INDEX(NAME, IDX)
#define GTL_H // interface/rp_main.h includes interface/gtl.h: already provided above

#define float Float            // GLSL semantics: fp32 everywhere, literals included (glsl_compat.h)
#include "common.glsl"         // RNG, hashes, orthonormal basis, ray offset, octahedral codec, sampling maps, safe_div, luminance
#include "colormap.glsl"
#include "interface/rp_main.h" // UniformData, the four light structs
// what the descriptor declarations of rp_main_descriptors.glsl provide
static UniformData ubo;
static const SphereLight* sphereLights; static const DistantLight* distantLights; static const RectLight* rectLights; static const DiskLight* diskLights;
static vec3 gl_WorldRayDirectionEXT;
struct State { vec3 normal; vec3 geom_normal; vec3 position; vec3 tangent_u[1]; vec3 tangent_v[1]; vec3 text_coords[1]; }; // the members used here of mdl_types.glsl's State
// what setup_mdl_shading_state reads besides its arguments: ray-tracing built-ins and the buffer references of rp_main_descriptors.glsl
static mat4x3 gl_ObjectToWorldEXT, gl_WorldToObjectEXT;
static uint gl_InstanceCustomIndexEXT = 0, gl_PrimitiveID = 0;
static BlasPayload blas_payloads[1];
static const Face* g_faces; static const FVertex* g_vertices;
struct IndexBuffer { const Face* data; BlasPayloadBufferPreamble preamble; IndexBuffer(uint64_t) : data(g_faces), preamble() {} };
struct VertexBuffer { const FVertex* data; VertexBuffer(uint64_t) : data(g_vertices) {} };
#define TEX_WRAP_CLAMP 0
#define TEX_WRAP_REPEAT 1
#define TEX_WRAP_MIRRORED_REPEAT 2
#define TEX_WRAP_CLIP 3
namespace stack0 {
#define MEDIUM_STACK_SIZE 0
#include "rp_main_payload.glsl"
#undef MEDIUM_STACK_SIZE
}
namespace stack2 {
#define MEDIUM_STACK_SIZE 2
#include "rp_main_payload.glsl"
#include "fn_sampleDistance.h"
#include "fn_sampleHenyeyGreensteinCos.h"
#include "fn_sampleVolumeScatteringDirection.h"
#undef MEDIUM_STACK_SIZE
}
namespace stack8 {
#define MEDIUM_STACK_SIZE 8
#include "rp_main_payload.glsl"
#undef MEDIUM_STACK_SIZE
}
#include "fn_russian_roulette.h"
#include "fn_fisGauss.h"
#include "fn_quatRotateDir.h"
#include "fn_sampleLight.h"
#include "fn_apply_wrap_and_crop.h"
// tex_lookup_float4_2d (mdl_interface.glsl:127-145): the texture array, textureSize and texture() are the caller's image under the oracle's
// software sampler (bilinear, REPEAT addressing, LOD 0 -- the reference's one hardware sampler, Gi.cpp:388-392; filter weights: D5)
struct Tex2D { const void* rgba; int w, h; };
static Tex2D textures_2d[1];
struct SamplerT {}; static SamplerT tex_sampler;
struct sampler2D { const Tex2D* t; sampler2D(const Tex2D& tex, const SamplerT&) : t(&tex) {} };
#define nonuniformEXT(x) (x)
#define TEXTURE_INDEX_OFFSET 0
static ivec2 textureSize(const Tex2D& t, int) { return ivec2(t.w, t.h); }
static vec4 texture(const sampler2D& s, vec2 uv)
{
  struct { const void* rgba; uint32_t w, h; } ot = {s.t->rgba, (uint32_t)s.t->w, (uint32_t)s.t->h};
  Float o[4]; orc_dbg_sample_bilinear(&ot, uv.x.v, uv.y.v, reinterpret_cast<decltype(o[0].v)*>(o));
  return vec4(o[0], o[1], o[2], o[3]);
}
#include "fn_tex_lookup_float4_2d.h"
// the remaining texture entry points (mdl_interface.glsl:45-65, 86-105, 167-221): texelFetch is the texel itself; the 3-D sampler is the oracle's trilinear one (D5)
typedef decltype(Float().v) RawF; // (`float` is the strict-fp32 class in here)
static vec4 texelFetch(const sampler2D& s, ivec2 c, int)
{ const RawF* p = (const RawF*)s.t->rgba + 4 * ((size_t)c.y * s.t->w + (size_t)c.x); return vec4(Float(p[0]), Float(p[1]), Float(p[2]), Float(p[3])); }
struct Tex3D { const void* rgba; int w, h, d; };
static Tex3D textures_3d[1];
struct sampler3D { const Tex3D* t; sampler3D(const Tex3D& tex, const SamplerT&) : t(&tex) {} };
static ivec3 textureSize(const Tex3D& t, int) { return ivec3(t.w, t.h, t.d); }
static vec4 texture(const sampler3D& s, vec3 uvw)
{
  Float o[4]; orc_dbg_sample_trilinear((const RawF*)s.t->rgba, (uint32_t)s.t->w, (uint32_t)s.t->h, (uint32_t)s.t->d, uvw.x.v, uvw.y.v, uvw.z.v, reinterpret_cast<decltype(o[0].v)*>(o));
  return vec4(o[0], o[1], o[2], o[3]);
}
static vec4 texelFetch(const sampler3D& s, ivec3 c, int)
{ const RawF* p = (const RawF*)s.t->rgba + 4 * (((size_t)c.z * s.t->h + (size_t)c.y) * s.t->w + (size_t)c.x); return vec4(Float(p[0]), Float(p[1]), Float(p[2]), Float(p[3])); }
#include "fn_tex_texel_float4_2d.h"
#include "fn_tex_resolution_2d.h"
#include "fn_tex_lookup_float4_3d.h"
#include "fn_tex_texel_float4_3d.h"
#include "fn_mdl_adapt_normal.h"
#include "mdl_shading_state.glsl"
#undef float
} // namespace ref

using ref::vec2; using ref::vec3; using ref::vec4; using glsl::Float;
static inline vec3 V3(const float* p) { return vec3(p[0], p[1], p[2]); }
static inline void put(float* o, const vec3& v) { o[0] = v.x.v; o[1] = v.y.v; o[2] = v.z.v; }

extern "C" {
// ---- common.glsl
uint32_t ref_hash_theironborn(uint32_t x) { return ref::hash_theironborn(x); }
uint32_t ref_hash_pcg32(uint32_t* state) { return ref::hash_pcg32(*state); }
uint32_t ref_rng1d_init(uint32_t pixelIndex, uint32_t sampleIndex) { return ref::rng1d_init(pixelIndex, sampleIndex); }
float ref_rng1d_next1f(uint32_t* state) { return ref::rng1d_next1f(*state).v; }
float ref_uint_as_float(uint32_t v) { return ref::uintAsFloat(v).v; }
void ref_orthonormal_basis(const float* n, float* b1, float* b2) { vec3 a, b; ref::orthonormal_basis(V3(n), a, b); put(b1, a); put(b2, b); }
void ref_offset_ray_origin(const float* p, const float* n, float* out) { put(out, ref::offset_ray_origin(V3(p), V3(n))); }
uint32_t ref_encode_direction(const float* d) { return ref::encode_direction(V3(d)); }
void ref_decode_direction(uint32_t e, float* out) { put(out, ref::decode_direction(e)); }
void ref_sample_hemisphere(float x0, float x1, float* out) { put(out, ref::sample_hemisphere(vec2(x0, x1))); }
void ref_sample_sphere(float x0, float x1, const float* radius, float* out) { put(out, ref::sample_sphere(vec2(x0, x1), V3(radius))); }
void ref_sample_disk(float x0, float x1, float rx, float ry, float* out) { const vec2 r = ref::sample_disk(vec2(x0, x1), vec2(rx, ry)); out[0] = r.x.v; out[1] = r.y.v; }
float ref_luminance(const float* c) { return ref::luminance(V3(c)).v; }
float ref_safe_div(float a, float b) { return ref::safe_div(Float(a), Float(b)).v; }
// ---- colormap.glsl: 0 viridis, 1 inferno, 2 turbo
void ref_colormap(int which, float t, float* out) { put(out, which == 0 ? ref::colormap_viridis(t) : (which == 1 ? ref::colormap_inferno(t) : ref::colormap_turbo(t))); }
// ---- rp_main_payload.glsl, for MEDIUM_STACK_SIZE 0 / 2 / 8
#define REF_PAYLOAD(NS, N) \
  uint32_t ref_payload_get_medium_idx_##N(uint32_t bitfield) { ref::NS::ShadeRayPayload p{}; p.bitfield = bitfield; return ref::NS::shadeRayPayloadGetMediumIdx(p); } \
  uint32_t ref_payload_set_medium_idx_##N(uint32_t bitfield, uint32_t idx) { ref::NS::ShadeRayPayload p{}; p.bitfield = bitfield; ref::NS::shadeRayPayloadSetMediumIdx(p, idx); return p.bitfield; }
REF_PAYLOAD(stack0, 0) REF_PAYLOAD(stack2, 2) REF_PAYLOAD(stack8, 8)
uint32_t ref_payload_increment_walk(uint32_t bitfield) { ref::stack2::ShadeRayPayload p{}; p.bitfield = bitfield; ref::stack2::shadeRayPayloadIncrementWalk(p); return p.bitfield; }
uint32_t ref_payload_get_walk(uint32_t bitfield) { ref::stack2::ShadeRayPayload p{}; p.bitfield = bitfield; return ref::stack2::shadeRayPayloadGetWalk(p); }
// ---- rp_main.rgen
void ref_fis_gauss(float x0, float x1, float* out) { const vec2 r = ref::fisGauss(vec2(x0, x1)); out[0] = r.x.v; out[1] = r.y.v; }
int ref_russian_roulette(float k, float rrInvMinTermProb, float* throughput) { ref::ubo.rrInvMinTermProb = rrInvMinTermProb; vec3 t = V3(throughput); const bool term = ref::russian_roulette(Float(k), t); put(throughput, t); return term ? 1 : 0; }
float ref_sample_distance(const float* albedo, const float* throughput, const float* sigma_t, float xi, float* pdf) { vec3 p; const Float r = ref::stack2::sampleDistance(V3(albedo), V3(throughput), V3(sigma_t), Float(xi), p); put(pdf, p); return r.v; }
float ref_sample_hg_cos(float r, float g) { return ref::stack2::sampleHenyeyGreensteinCos(Float(r), Float(g)).v; }
void ref_sample_volume_direction(float x0, float x1, float bias, float* dir) { vec3 d = V3(dir); ref::stack2::sampleVolumeScatteringDirection(vec2(x0, x1), Float(bias), d); put(dir, d); }
// ---- rp_main.miss
void ref_quat_rotate_dir(const float* q, const float* dir, float* out) { put(out, ref::quatRotateDir(vec4(q[0], q[1], q[2], q[3]), V3(dir))); }
// ---- mdl_interface.glsl
float ref_apply_wrap_and_crop(float coord, int wrap, float crop0, float crop1, int res) { return ref::apply_wrap_and_crop(Float(coord), wrap, vec2(crop0, crop1), res).v; }
// tex: 0 = the invalid texture, 1 = the image given
void ref_tex_lookup_float4_2d(const float* rgba, int w, int h, int tex, float u, float v, int wrapU, int wrapV, float* out)
{
  ref::textures_2d[0] = ref::Tex2D{rgba, w, h};
  vec4 r = ref::tex_lookup_float4_2d(tex, vec2(u, v), wrapU, wrapV, vec2(0.0f, 1.0f), vec2(0.0f, 1.0f), Float(0.0f));
  out[0] = r.x.v; out[1] = r.y.v; out[2] = r.z.v; out[3] = r.w.v;
}
// the remaining texture entry points; same query layout as the oracle's orc_tex_runtime (kind, valid, c0, c1, c2, wrapU, wrapV, wrapW)
void ref_tex_runtime(const float* rgba, int w, int h, int d, uint32_t count, const float* queries, float* out)
{
  ref::textures_2d[0] = ref::Tex2D{rgba, w, h}; ref::textures_3d[0] = ref::Tex3D{rgba, w, h, d};
  for (uint32_t i = 0; i < count; i++) {
    const float* q = queries + 8 * (size_t)i; float* o = out + 4 * (size_t)i;
    const int kind = (int)q[0], tex = q[1] != 0.0f ? 1 : 0;
    vec4 r(Float(0.0f), Float(0.0f), Float(0.0f), Float(0.0f));
    if (kind == 0) r = ref::tex_texel_float4_2d(tex, ref::ivec2((int)q[2], (int)q[3]), ref::ivec2(0, 0), Float(0.0f));
    else if (kind == 1) { const ref::ivec2 res = ref::tex_resolution_2d(tex, ref::ivec2(0, 0), Float(0.0f)); r = vec4(Float((float)res.x), Float((float)res.y), Float(0.0f), Float(0.0f)); }
    else if (kind == 2) r = ref::tex_lookup_float4_3d(tex, vec3(q[2], q[3], q[4]), (int)q[5], (int)q[6], (int)q[7], vec2(0.0f, 1.0f), vec2(0.0f, 1.0f), vec2(0.0f, 1.0f), Float(0.0f));
    else r = ref::tex_texel_float4_3d(tex, ref::ivec3((int)q[2], (int)q[3], (int)q[4]), Float(0.0f));
    o[0] = r.x.v; o[1] = r.y.v; o[2] = r.z.v; o[3] = r.w.v;
  }
}
void ref_adapt_normal(const float* rayDir, const float* geomNormal, const float* shadingNormal, const float* normal, float* out)
{ ref::gl_WorldRayDirectionEXT = V3(rayDir); ref::State s; s.normal = V3(shadingNormal); s.geom_normal = V3(geomNormal); put(out, ref::mdl_adapt_normal(s, V3(normal))); }
// ---- rp_main.chit: sampleLight over caller-provided light arrays (layouts: interface/rp_main.h; 48 bytes each)
struct RefLightSetup { uint32_t sphereCount, distantCount, rectCount, diskCount; float lightIntensityMultiplier, sensorExposure; const void* sphere; const void* distant; const void* rect; const void* disk; };
void ref_sample_light(const RefLightSetup* L, const float* k4, const float* surfacePos, float* dirToLight, float* dist, float* power, float* invPdf, uint32_t* diffuseSpecularPacked)
{
  ref::ubo.sphereLightCount = L->sphereCount; ref::ubo.distantLightCount = L->distantCount; ref::ubo.rectLightCount = L->rectCount; ref::ubo.diskLightCount = L->diskCount;
  ref::ubo.totalLightCount = L->sphereCount + L->distantCount + L->rectCount + L->diskCount;
  ref::ubo.lightIntensityMultiplier = L->lightIntensityMultiplier; ref::ubo.sensorExposure = L->sensorExposure;
  ref::sphereLights = (const ref::SphereLight*)L->sphere; ref::distantLights = (const ref::DistantLight*)L->distant;
  ref::rectLights = (const ref::RectLight*)L->rect; ref::diskLights = (const ref::DiskLight*)L->disk;
  vec3 d, p; Float di, ip; glsl::uint ds = 0;
  ref::sampleLight(vec4(k4[0], k4[1], k4[2], k4[3]), V3(surfacePos), d, di, p, ip, ds);
  put(dirToLight, d); *dist = di.v; put(power, p); *invPdf = ip.v; *diffuseSpecularPacked = ds;
}
// ---- mdl_shading_state.glsl: setup_mdl_shading_state for ONE triangle.  fvertex: 3 x 8 floats in the reference's packed vertex layout
// (rp_main.h:58-64: pos, bitangent sign | encoded normal, encoded tangent (uint bits), u, v); o2w: 3x4 rows; w2o: 3x3 rows.
// out: position, normal, geom normal, tangent_u, tangent_v, (u, v, frontFace)
void ref_setup_shading_state(const float* fvertex, const float* o2w, const float* w2o, const float* rayDir, float bu, float bv, float* out)
{
  static const ref::Face face{0u, 1u, 2u};
  ref::g_faces = &face; ref::g_vertices = (const ref::FVertex*)fvertex;
  ref::blas_payloads[0].bufferAddress = 0; ref::blas_payloads[0].vertexOffset = 0; ref::blas_payloads[0].bitfield = 0;
  for (int c = 0; c < 4; c++) ref::gl_ObjectToWorldEXT.c[c] = vec3(o2w[c], o2w[4 + c], o2w[8 + c]);
  for (int c = 0; c < 3; c++) ref::gl_WorldToObjectEXT.c[c] = vec3(w2o[c], w2o[3 + c], w2o[6 + c]);
  ref::gl_WorldToObjectEXT.c[3] = vec3(0.0f);
  ref::gl_WorldRayDirectionEXT = V3(rayDir);
  ref::State st; bool front = false;
  ref::setup_mdl_shading_state(vec2(bu, bv), st, front);
  put(out, st.position); put(out + 3, st.normal); put(out + 6, st.geom_normal); put(out + 9, st.tangent_u[0]); put(out + 12, st.tangent_v[0]);
  out[15] = st.text_coords[0].x.v; out[16] = st.text_coords[0].y.v; out[17] = front ? 1.0f : 0.0f;
}
int ref_light_struct_sizes(int which) { return which == 0 ? (int)sizeof(ref::SphereLight) : which == 1 ? (int)sizeof(ref::DistantLight) : which == 2 ? (int)sizeof(ref::RectLight) : (int)sizeof(ref::DiskLight); }
}
