// ref_loop.cpp -- the reference's rp_main.rgen / rp_main.chit / rp_main.miss / rp_main_shadow.miss, compiled as C++ from /root/reference
// (oracle/ref/build_ref.py writes the language-forced rewrites into oracle/_ref/gen_loop/) and RUN on the CPU: the sample loop, camera
// rays, depth of field, clip planes, the bounce loop, next-event estimation, Russian roulette, the volume random walk, emission, the
// medium stack, radiance clamping, accumulation and the progressive blend are the reference's own text.  What the reference gets from the
// Vulkan driver and from the MDL code generator comes from the oracle through the orc_hook_* entry points: the scene as the host packs
// it, ray queries (traceRayEXT -> the oracle's traversal incl. its cutout rule, deviation D1), and the closed-form BSDF / EDF functions.
// tests/test_oracle_ref_loop.py compares the images with orc_render's.  Test infrastructure only.
//
// One object per set of feature macros (the reference generates a shader per render-settings combination): -DREF_VARIANT=<name> and the
// macros of build_ref.py's LOOP_VARIANTS.
#include "glsl_compat.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#undef UINT32_MAX

extern "C" { // oracle/gi_oracle.cpp
void orc_hook_frame(void* h, float* out);
int orc_hook_trace(void* h, const float* o, const float* d, float tMin, float tMax, int anyHit, uint32_t rng, float* tuv, uint32_t* instPrim);
void orc_hook_instance(void* h, uint32_t inst, float* o2w, float* w2o, int32_t* info);
const float* orc_hook_mesh_vertices(void* h, uint32_t mesh);
const uint32_t* orc_hook_mesh_faces(void* h, uint32_t mesh);
void orc_hook_lights(void* h, uint32_t* counts, const float** ptrs);
void orc_hook_material(void* h, uint32_t mat, float* out);
void orc_hook_bsdf_sample(void* h, uint32_t mat, const float* frame, const float* k1, const float* xi, float ior1, float ior2, int thin, float* out);
void orc_hook_bsdf_evaluate(void* h, uint32_t mat, const float* frame, const float* k1, const float* k2, float ior1, float ior2, int thin, float* out);
void orc_hook_edf_factor(void* h, uint32_t mat, float c, float* out);
int orc_hook_dome(void* h, float* rotationEmission);
void orc_hook_dome_lookup(void* h, float u, float v, float* rgb);
void orc_hook_bsdf_albedo(void* h, uint32_t mat, const float* frame, const float* k1, float ior1, float ior2, int thin, float* out);
const int32_t* orc_hook_mesh_face_ids(void* h, uint32_t mesh, uint32_t* stride);
}

#define REF_CAT2(a, b) a##b
#define REF_CAT(a, b) REF_CAT2(a, b)
#define REF_NS REF_CAT(refloop_, REF_VARIANT)

namespace REF_NS {
using namespace glsl;
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

// ---- per-material macros of the generated hit shaders: the superset (the stubs below answer per material at run time)
#define IS_EMISSIVE
#define IS_THIN_WALLED
// colour + the AOVs whose rules live in rgen / chit (normal, NEE, bounces); the "aovs" variant asks for all but ClockCycles (clockARB, D6)
#ifndef AOV_MASK
#define AOV_MASK ((1 << 0) | (1 << 1) | (1 << 2) | (1 << 5))
#endif

#define float Float
// ---- ray-tracing built-ins and resources the shaders name
struct AccelerationStructure {};
static AccelerationStructure sceneAS;
static uvec3 gl_LaunchIDEXT;
static vec3 gl_WorldRayDirectionEXT;
static float gl_HitTEXT, gl_RayTmaxEXT;
static uint gl_InstanceCustomIndexEXT, gl_PrimitiveID, gl_InstanceID;
static mat4x3 gl_ObjectToWorldEXT, gl_WorldToObjectEXT;
static vec2 baryCoord;
static const uint gl_RayFlagsTerminateOnFirstHitEXT = 4u, gl_RayFlagsSkipClosestHitShaderEXT = 8u;
static void traceRayEXT(AccelerationStructure&, uint rayFlags, uint cullMask, uint sbtRecordOffset, uint sbtRecordStride, uint missIndex, vec3 origin, float tMin, vec3 direction,
                        float tMax, int payload);

#include "rp_main.rgen" // -> aovs.glsl, rp_main_payload.glsl (common.glsl), rp_main_descriptors.glsl (interface/rp_main.h), colormap.glsl; defines rgen_main()

// ---- what the descriptors' buffer-reference blocks are to the shaders (rp_main_descriptors.glsl:75-85)
static void* g_hook;
static const BlasPayload* blas_payloads;
static const Face* g_faces; static const FVertex* g_vertices;
static BlasPayloadBufferPreamble g_preamble; static const int* g_faceIdWords;
struct IndexBuffer { const Face* data; BlasPayloadBufferPreamble preamble; IndexBuffer(uint64_t) : data(g_faces), preamble(g_preamble) {} };
struct RawIntBuffer { const int* data; RawIntBuffer(uint64_t) : data(g_faceIdWords) {} }; // faceIdsInfo's offset is 0 here
struct VertexBuffer { const FVertex* data; VertexBuffer(uint64_t) : data(g_vertices) {} };
static uint g_material; static bool g_thinWalled; static float g_matInfo[13];

#ifndef SCENE_DATA_COUNT
#define SCENE_DATA_COUNT 0
#endif
#if SCENE_DATA_COUNT > 0
#include "mdl_renderer_state.glsl"
#endif
#include "mdl_types.glsl"
#if SCENE_DATA_COUNT > 0
// buffer references of the scene-data readers (mdl_interface.glsl:258-262): the "device address" is a host pointer here
struct BufferRefInt { const int* data; BufferRefInt(uint64_t a) : data(reinterpret_cast<const int*>(a)) {} };
struct BufferRefFloat { const Float* data; BufferRefFloat(uint64_t a) : data(reinterpret_cast<const Float*>(a)) {} };
struct BufferRefVec2 { const vec2* data; BufferRefVec2(uint64_t a) : data(reinterpret_cast<const vec2*>(a)) {} };
struct BufferRefVec4 { const vec4* data; BufferRefVec4(uint64_t a) : data(reinterpret_cast<const vec4*>(a)) {} };
#include "fn_scene_data.h"
#endif

// ---- the entry points the MDL back end generates per material (GlslShaderGen.cpp:181-193), answered by the oracle's closed forms
static inline void to3(const vec3& v, float* o);
static bool mdl_thin_walled(State) { return g_thinWalled; }
static vec3 mdl_volume_absorption_coefficient(State) { return vec3(g_matInfo[4], g_matInfo[5], g_matInfo[6]); }
static vec3 mdl_volume_scattering_coefficient(State) { return vec3(g_matInfo[7], g_matInfo[8], g_matInfo[9]); }
static vec3 mdl_ior(State) { return vec3(g_matInfo[10]); }
#define MEDIUM_DIRECTIONAL_BIAS g_matInfo[11]
static void mdl_edf_emission_init(State&) {}
static void mdl_edf_emission_evaluate(Edf_evaluate_data& d, State st);
static vec3 mdl_edf_emission_intensity(State) { return vec3(g_matInfo[1], g_matInfo[2], g_matInfo[3]); }
static void mdl_bsdf_scattering_init(State&) {}
static void mdl_bsdf_scattering_sample(Bsdf_sample_data& d, State st);
static void mdl_bsdf_scattering_evaluate(Bsdf_evaluate_data& d, State st);
static void mdl_bsdf_scattering_auxiliary(Bsdf_auxiliary_data& d, State st);

#include "mdl_shading_state.glsl"
#include "rp_main.chit"
// the two dome-light texture slots (Gi.cpp:2184-2238, 2330-2337): [0] the 1x1 fallback texel (the colour AOV's clear value), [1] the dome light's
// equirectangular image, or the fallback again when the scene has none.  textureLod = the oracle's software sampler (bilinear, REPEAT; D5);
// the lookup coordinates are rp_main.miss's own (sampleDomeLight: atan / acos of the rotated direction)
static vec3 g_background; static bool g_hasDome;
struct Tex2D { uint idx; }; static Tex2D textures_2d[2] = {{0u}, {1u}};
struct SamplerT {}; static SamplerT tex_sampler;
struct sampler2D { uint idx; sampler2D(const Tex2D& t, const SamplerT&) : idx(t.idx) {} };
#define nonuniformEXT(x) (x)
static vec4 textureLod(const sampler2D& s, vec2 uv, uint)
{
  if (s.idx == 0u || !g_hasDome) return vec4(g_background.x, g_background.y, g_background.z, 1.0f);
  Float rgb[3]; orc_hook_dome_lookup(g_hook, uv.x.v, uv.y.v, reinterpret_cast<decltype(rgb[0].v)*>(rgb)); // (Float is a float)
  return vec4(rgb[0], rgb[1], rgb[2], 1.0f);
}
#include "rp_main.miss"
namespace shadow_miss {
#include "rp_main_shadow.miss"
}
#undef float

static inline void to3(const vec3& v, float* o) { o[0] = v.x.v; o[1] = v.y.v; o[2] = v.z.v; }
static void frame_of(const State& st, float* f) { to3(st.normal, f); to3(st.tangent_u[0], f + 3); to3(st.tangent_v[0], f + 6); to3(st.geom_normal, f + 9); }
static void mdl_edf_emission_evaluate(Edf_evaluate_data& d, State st)
{
  d.cos = dot(d.k1, st.normal);
  d.pdf = d.cos > 0.0f ? Float(1.0f) : Float(0.0f); // uniform EDF: "pdf > 0 iff cos > 0"; edf * intensity = emission colour (x the OpenPBR coat factor)
  float f[3]; orc_hook_edf_factor(g_hook, g_material, d.cos.v, f);
  d.edf = vec3(f[0], f[1], f[2]);
}
static void mdl_bsdf_scattering_sample(Bsdf_sample_data& d, State st)
{
  float fr[12], k1[3], xi[4] = {d.xi.x.v, d.xi.y.v, d.xi.z.v, d.xi.w.v}, out[8];
  frame_of(st, fr); to3(d.k1, k1);
  orc_hook_bsdf_sample(g_hook, g_material, fr, k1, xi, d.ior1.x.v, d.ior2.x.v, g_thinWalled ? 1 : 0, out);
  d.k2 = vec3(out[0], out[1], out[2]); d.bsdf_over_pdf = vec3(out[3], out[4], out[5]); d.pdf = out[6];
  uint32_t e; memcpy(&e, &out[7], 4); d.event_type = (int)e; d.handle = 0;
}
static void mdl_bsdf_scattering_evaluate(Bsdf_evaluate_data& d, State st)
{
  float fr[12], k1[3], k2[3], out[7];
  frame_of(st, fr); to3(d.k1, k1); to3(d.k2, k2);
  orc_hook_bsdf_evaluate(g_hook, g_material, fr, k1, k2, d.ior1.x.v, d.ior2.x.v, g_thinWalled ? 1 : 0, out);
  d.bsdf_diffuse = vec3(out[0], out[1], out[2]); d.bsdf_glossy = vec3(out[3], out[4], out[5]); d.pdf = out[6];
}
static void mdl_bsdf_scattering_auxiliary(Bsdf_auxiliary_data& d, State st)
{
  float fr[12], k1[3], out[3];
  frame_of(st, fr); to3(d.k1, k1);
  orc_hook_bsdf_albedo(g_hook, g_material, fr, k1, d.ior1.x.v, d.ior2.x.v, g_thinWalled ? 1 : 0, out);
  d.albedo_diffuse = vec3(out[0], out[1], out[2]); d.albedo_glossy = vec3(0.0f); // the Albedo AOV only uses the sum (rp_main.chit:279)
}

// traceRayEXT: the oracle's traversal stands in for the acceleration structure; hit -> rp_main.chit, miss -> rp_main.miss (payload 0),
// rp_main_shadow.miss (payload 1; the shadow hit group has no closest-hit shader)
static void traceRayEXT(AccelerationStructure&, uint rayFlags, uint, uint, uint, uint, vec3 origin, Float tMin, vec3 direction, Float tMax, int payload)
{
  float o[3], d[3], tuv[3]; uint32_t ip[2];
  to3(origin, o); to3(direction, d);
  gl_WorldRayDirectionEXT = direction; gl_RayTmaxEXT = tMax;
  if (payload == 1) {
    const bool hit = tMax.v > tMin.v && orc_hook_trace(g_hook, o, d, tMin.v, tMax.v, 1, shadowRayPayload.rng_state, tuv, ip) != 0; // an empty interval hits nothing
    if (!hit) shadow_miss::shadow_miss_main();
    return;
  }
  if (!orc_hook_trace(g_hook, o, d, tMin.v, tMax.v, 0, rayPayload.rng_state, tuv, ip)) { miss_main(); return; }
  float o2w[12], w2o[9]; int32_t info[5];
  orc_hook_instance(g_hook, ip[0], o2w, w2o, info);
  for (int c = 0; c < 4; c++) gl_ObjectToWorldEXT.c[c] = vec3(o2w[c], o2w[4 + c], o2w[8 + c]);
  for (int c = 0; c < 3; c++) gl_WorldToObjectEXT.c[c] = vec3(w2o[c], w2o[3 + c], w2o[6 + c]);
  gl_WorldToObjectEXT.c[3] = vec3(0.0f);
  gl_HitTEXT = tuv[0]; baryCoord = vec2(tuv[1], tuv[2]);
  gl_PrimitiveID = ip[1]; gl_InstanceID = ip[0]; gl_InstanceCustomIndexEXT = 0;
  g_faces = (const Face*)orc_hook_mesh_faces(g_hook, (uint32_t)info[0]); g_vertices = (const FVertex*)orc_hook_mesh_vertices(g_hook, (uint32_t)info[0]);
  static BlasPayload bp[1];
  bp[0].bufferAddress = 0; bp[0].vertexOffset = 0; bp[0].bitfield = (uint)info[1]; // BLAS_PAYLOAD_BITFLAG_FLIP_FACING | _DOUBLE_SIDED (rp_main.h:115-116)
  blas_payloads = bp;
  g_material = (uint)info[2];
  { // BlasPayloadBufferPreamble (Gi.cpp:886-905) and the instance-id buffer, for the ObjectId / FaceId / InstanceId AOVs
    uint32_t stride = 1; g_faceIdWords = (const int*)orc_hook_mesh_face_ids(g_hook, (uint32_t)info[0], &stride);
    g_preamble.objectId = info[3]; g_preamble.faceIdsInfo = (stride << FACE_ID_STRIDE_OFFSET) | 0u;
    static std::vector<int> ids; if (ids.size() <= ip[0]) ids.resize(ip[0] + 1); ids[ip[0]] = info[4]; InstanceIds = ids.data();
  }
  orc_hook_material(g_hook, g_material, reinterpret_cast<float*>(g_matInfo)); // (Float is a float)
  g_thinWalled = g_matInfo[12].v != 0.0f;
  chit_main();
}
} // namespace REF_NS

// ---- C entry point: renders rows [rowBegin, rowEnd) of the frame; colour / normal / NEE / bounces AOVs as float4 per pixel of the band
struct RefLoopParams {
  float camPos[3], camFwd[3], camUp[3]; float vfov, focusDistance, exposure, frame, time;
  uint32_t width, height, rowBegin, rowEnd, spp, sampleOffset, maxBounces, rrBounceOffset, maxVolumeWalkLength;
  float maxSampleValue, rrInvMinTermProb, lightIntensityMultiplier, metersPerSceneUnit;
  float clearColor[4], clearNormal[4], clearNee[4], clearBounces[4];
};
#define REF_ENTRY REF_CAT(ref_loop_render_, REF_VARIANT)
// extra (may be null): 17 pointers indexed by AOV id (aovs.glsl:5-21), each null or float4 per pixel of the band (integer AOVs: the int's bits in .x);
// clearExtra: their clear values; prevExtra: previous contents of the accumulating ones (normal = id 1 through `normal`, albedo = 16)
extern "C" int REF_ENTRY(void* hook, const RefLoopParams* p, const float* prevColor, float* color, float* normal, float* nee, float* bounces, float* const* extra,
                         const float* clearExtra, const float* prevAlbedo)
{
  using namespace REF_NS;
  g_hook = hook;
  float fr[5]; orc_hook_frame(hook, fr);
  g_background = vec3(fr[0], fr[1], fr[2]);
  // UniformData as Gi.cpp:2373-2426 fills it
  ubo = UniformData();
  uint32_t counts[4]; const float* ptrs[4]; orc_hook_lights(hook, counts, ptrs);
  ubo.sphereLightCount = counts[0]; ubo.distantLightCount = counts[1]; ubo.rectLightCount = counts[2]; ubo.diskLightCount = counts[3];
  ubo.totalLightCount = counts[0] + counts[1] + counts[2] + counts[3];
  static const float zeroLight[12] = {0}; // an empty light store still holds one zero-filled element (GgpuSyncBuffer: SyncBuffer.cpp:90), which sampleLight reads
  for (int i = 0; i < 4; i++) if (counts[i] == 0u) ptrs[i] = zeroLight;
  sphereLights = (SphereLight*)ptrs[0]; distantLights = (DistantLight*)ptrs[1]; rectLights = (RectLight*)ptrs[2]; diskLights = (DiskLight*)ptrs[3];
  ubo.metersPerSceneUnit = p->metersPerSceneUnit; ubo.maxVolumeWalkLength = p->maxVolumeWalkLength;
  ubo.cameraPosition = vec3(p->camPos[0], p->camPos[1], p->camPos[2]);
  ubo.imageDims = (p->height << 16) | p->width;
  ubo.cameraForward = normalize(vec3(p->camFwd[0], p->camFwd[1], p->camFwd[2])); ubo.cameraUp = normalize(vec3(p->camUp[0], p->camUp[1], p->camUp[2]));
  ubo.focusDistance = p->focusDistance; ubo.cameraVFoV = p->vfov; ubo.sampleOffset = p->sampleOffset; ubo.lensRadius = fr[3]; ubo.spp = p->spp;
  ubo.invTotalSampleCount = 1.0f / float(p->sampleOffset + p->spp); ubo.maxSampleValue = p->maxSampleValue;
  ubo.maxBouncesAndRrBounceOffset = (p->maxBounces << 16) | (p->rrBounceOffset & 0xffffu);
  ubo.rrInvMinTermProb = p->rrInvMinTermProb; ubo.lightIntensityMultiplier = p->lightIntensityMultiplier;
  uint32_t cr; memcpy(&cr, &fr[4], 4); ubo.clipRangePacked = cr;
  ubo.sensorExposure = p->exposure; ubo.frame = p->frame; ubo.time = p->time;
  ubo.domeLightRotation = vec4(0.0f, 0.0f, 0.0f, 1.0f); ubo.domeLightEmissionMultiplier = vec3(1.0f); // no dome light: the uniform fallback texture (Gi.cpp:2232-2238, 2384-2385)
  float dm[7]; g_hasDome = orc_hook_dome(hook, dm) != 0;
  if (g_hasDome) { ubo.domeLightRotation = vec4(dm[0], dm[1], dm[2], dm[3]); ubo.domeLightEmissionMultiplier = vec3(dm[4], dm[5], dm[6]); } // Gi.cpp:2384-2396
  const size_t n = (size_t)p->width * p->height;
  std::vector<vec4> colorBuf(n), clearF(17); std::vector<ivec4> clearI(17); std::vector<vec3> normalBuf(n), neeBuf(n), bouncesBuf(n);
  clearF[0] = vec4(p->clearColor[0], p->clearColor[1], p->clearColor[2], p->clearColor[3]);
  clearF[1] = vec4(p->clearNormal[0], p->clearNormal[1], p->clearNormal[2], p->clearNormal[3]);
  clearF[2] = vec4(p->clearNee[0], p->clearNee[1], p->clearNee[2], p->clearNee[3]);
  clearF[5] = vec4(p->clearBounces[0], p->clearBounces[1], p->clearBounces[2], p->clearBounces[3]);
  std::vector<vec3> v3Buf[17]; std::vector<int> iBuf[17]; std::vector<Float> depthBuf(n);
  for (int id = 3; id < 17; id++) {
    if (clearExtra) { clearF[id] = vec4(clearExtra[4 * id], clearExtra[4 * id + 1], clearExtra[4 * id + 2], clearExtra[4 * id + 3]); int ci; memcpy(&ci, &clearExtra[4 * id], 4); clearI[id] = ivec4{ci, 0, 0, 0}; }
    v3Buf[id].resize(n); iBuf[id].resize(n);
  }
  ClearValuesF = clearF.data(); ClearValuesI = clearI.data();
  ColorAov = colorBuf.data(); NormalsAov = normalBuf.data(); NeeAov = neeBuf.data(); BouncesAov = bouncesBuf.data();
#if (AOV_MASK & 0x1ff98) == 0x1ff98 // the "aovs" variant: the descriptors declare an AOV only when its bit is set
  BarycentricsAov = v3Buf[3].data(); TexcoordsAov = v3Buf[4].data(); OpacityAov = v3Buf[7].data(); TangentsAov = v3Buf[8].data(); BitangentsAov = v3Buf[9].data();
  ThinWalledAov = v3Buf[10].data(); ObjectIdAov = iBuf[11].data(); DepthAov = depthBuf.data(); FaceIdAov = iBuf[13].data(); InstanceIdAov = iBuf[14].data();
  DoubleSidedAov = v3Buf[15].data(); AlbedoAov = v3Buf[16].data();
#else
  if (extra) return 1;
#endif
  for (uint32_t y = p->rowBegin; y < p->rowEnd; y++)
    for (uint32_t x = 0; x < p->width; x++) {
      const size_t pi = (size_t)y * p->width + x, o = ((size_t)(y - p->rowBegin) * p->width + x) * 4;
      if (prevColor) colorBuf[pi] = vec4(prevColor[o], prevColor[o + 1], prevColor[o + 2], prevColor[o + 3]);
      neeBuf[pi] = clearF[2].rgb(); bouncesBuf[pi] = clearF[5].rgb();
      if (prevAlbedo) v3Buf[16][pi] = vec3(prevAlbedo[o], prevAlbedo[o + 1], prevAlbedo[o + 2]);
      if (p->sampleOffset > 0 && normal) normalBuf[pi] = vec3(normal[o], normal[o + 1], normal[o + 2]); // in/out for the accumulating normal AOV
      gl_LaunchIDEXT = uvec3{x, y, 0u};
      rgen_main();
      const vec4 c = colorBuf[pi];
      color[o] = c.x.v; color[o + 1] = c.y.v; color[o + 2] = c.z.v; color[o + 3] = c.w.v;
      if (normal) { normal[o] = normalBuf[pi].x.v; normal[o + 1] = normalBuf[pi].y.v; normal[o + 2] = normalBuf[pi].z.v; normal[o + 3] = 0.0f; }
      if (nee) { nee[o] = neeBuf[pi].x.v; nee[o + 1] = neeBuf[pi].y.v; nee[o + 2] = neeBuf[pi].z.v; nee[o + 3] = 0.0f; }
      if (bounces) { bounces[o] = bouncesBuf[pi].x.v; bounces[o + 1] = bouncesBuf[pi].y.v; bounces[o + 2] = bouncesBuf[pi].z.v; bounces[o + 3] = 0.0f; }
      if (extra)
        for (int id = 3; id < 17; id++) {
          float* e = extra[id]; if (!e) continue;
          if (id == 11 || id == 13 || id == 14) { memcpy(&e[o], &iBuf[id][pi], 4); e[o + 1] = e[o + 2] = e[o + 3] = 0.0f; }
          else if (id == 12) { e[o] = depthBuf[pi].v; e[o + 1] = e[o + 2] = e[o + 3] = 0.0f; }
          else { e[o] = v3Buf[id][pi].x.v; e[o + 1] = v3Buf[id][pi].y.v; e[o + 2] = v3Buf[id][pi].z.v; e[o + 3] = 0.0f; }
        }
    }
  return 0;
}

#if SCENE_DATA_COUNT > 0
// One scene-data read through the reference's readers.  kind: 1 float, 2 float2, 3 float3, 4 float4, 11..14 int .. int4.  infos: the six
// sceneDataInfos words of the BLAS payload preamble (Gi.cpp:955-1018), buffer: the payload buffer they index into (offsets in units of 32 B).
extern "C" void ref_scene_data_lookup(int kind, const uint32_t* infos, const void* buffer, const uint32_t* hitIndices, const float* bary, uint32_t primitiveId, int32_t instanceId,
                                      int sceneDataId, int uniformLookup, const float* defaults, const float* cameraPosition, float frame, float* out)
{
  using namespace REF_NS;
  State st; st.renderer_state.hitIndices = uvec3(hitIndices[0], hitIndices[1], hitIndices[2]); st.renderer_state.hitBarycentrics = vec2(bary[0], bary[1]);
  st.renderer_state.sceneDataBufferAddress = reinterpret_cast<uint64_t>(buffer);
  for (int i = 0; i < SCENE_DATA_COUNT; i++) st.renderer_state.sceneDataInfos[i] = infos[i];
  gl_PrimitiveID = primitiveId; gl_InstanceID = 0; static int ids[1]; ids[0] = instanceId; InstanceIds = ids;
  ubo.cameraPosition = vec3(cameraPosition[0], cameraPosition[1], cameraPosition[2]); ubo.frame = frame;
  const bool ul = uniformLookup != 0;
  int iv[4]; for (int c = 0; c < 4; c++) memcpy(&iv[c], &defaults[c], 4);
  switch (kind) {
  case 1: out[0] = scene_data_lookup_float(st, sceneDataId, defaults[0], ul).v; break;
  case 2: { vec2 r = scene_data_lookup_float2(st, sceneDataId, vec2(defaults[0], defaults[1]), ul); out[0] = r.x.v; out[1] = r.y.v; } break;
  case 3: { vec3 r = scene_data_lookup_float3(st, sceneDataId, vec3(defaults[0], defaults[1], defaults[2]), ul); out[0] = r.x.v; out[1] = r.y.v; out[2] = r.z.v; } break;
  case 4: { vec4 r = scene_data_lookup_float4(st, sceneDataId, vec4(defaults[0], defaults[1], defaults[2], defaults[3]), ul); out[0] = r.x.v; out[1] = r.y.v; out[2] = r.z.v; out[3] = r.w.v; } break;
  case 11: { int r = scene_data_lookup_int(st, sceneDataId, iv[0], ul); memcpy(&out[0], &r, 4); } break;
  case 12: { ivec2 r = scene_data_lookup_int2(st, sceneDataId, ivec2(iv[0], iv[1]), ul); memcpy(&out[0], &r.x, 4); memcpy(&out[1], &r.y, 4); } break;
  case 13: { ivec3 r = scene_data_lookup_int3(st, sceneDataId, ivec3(iv[0], iv[1], iv[2]), ul); memcpy(&out[0], &r.x, 4); memcpy(&out[1], &r.y, 4); memcpy(&out[2], &r.z, 4); } break;
  case 14: { ivec4 r = scene_data_lookup_int4(st, sceneDataId, ivec4(iv[0], iv[1], iv[2], iv[3]), ul); memcpy(&out[0], &r.x, 4); memcpy(&out[1], &r.y, 4); memcpy(&out[2], &r.z, 4); memcpy(&out[3], &r.w, 4); } break;
  default: break;
  }
}
#endif

