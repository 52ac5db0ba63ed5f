/*
 * gi_oracle.h -- CPU ORACLE for the gatling `gi/` render loop.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a from-scratch CPU restatement of the reference's device hot path
 * (/root/reference/src/gi/shaders/rp_main.rgen, rp_main.chit, rp_main.miss,
 * rp_main_shadow.miss, common.glsl, mdl_shading_state.glsl, rp_main_payload.glsl,
 * interface/rp_main.h) plus the host packing that defines its inputs
 * (src/gi/impl/Gi.cpp:287-300, 641-658, 826-1157, 1188-1202, 2373-2426, 2573-2976).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * call this library, and only as the checker.  The product (gatling_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" for BSDF arithmetic and traversal tie-breaking:
 * the reference's BSDF code lives in the MDL SDK 2024.1.4 + MaterialX (neither is in
 * /root/reference), the reference needs a Vulkan-RT device + OpenUSD to run, and every
 * reference image under src/hdGatling/testenv is a git-LFS stub.
 * What IS pinned here: RNG, octahedral codec, ray offset, FIS, sampling maps, payload bit
 * fields, Russian roulette, volume sampling, light sampling, dome rotation, texture wrap,
 * normal adaption, colour maps -- checked against the reference's OWN functions, compiled
 * from /root/reference/src/gi/shaders as C++ (oracle/ref/, oracle/_ref/libgi_ref.so;
 * tests/test_oracle_ref.py: bit-exact, a few ulp where sin / cos / log are involved) -- and
 * the whole render loop: rp_main.rgen / .chit / .miss / _shadow.miss compiled the same way and
 * RUN (oracle/ref/ref_loop.cpp; this oracle only answers their ray queries and MDL entry
 * points through the orc_hook_* functions), images compared with orc_render's in
 * tests/test_oracle_ref_loop.py: bit-identical on 90-100 % of the pixels, 4e-5 on the rest.
 *
 * Arithmetic contract shared with the HIP kernels (DESIGN.md section "Arithmetic contract"):
 * fp32 only, no FMA contraction, IEEE +,-,*,/,sqrt, and the two polynomial
 * transcendentals orc_sincos2pi / orc_logf below.  With that, HIP == oracle bit-for-bit.
 */
#ifndef GI_ORACLE_H
#define GI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* == GiVertex, Gi.h:110-118 */
typedef struct OrcVertex {
  float pos[3];
  float u;
  float norm[3];
  float v;
  float tangent[3];
  float bitangentSign;
} OrcVertex;

/* Material classes (closed-form BSDFs; replaces MDL codegen). */
enum {
  ORC_MAT_DIFFUSE = 0,             /* C1 "PR1 ref model": Lambert + uniform emission */
  ORC_MAT_USD_PREVIEW_SURFACE = 1, /* diffuse + GGX specular + clearcoat             */
  ORC_MAT_OPEN_PBR = 2
};

/* Parameter block indices (float p[64]); shared by all classes where meaningful. */
enum {
  ORC_P_BASE_COLOR = 0,        /* 3: diffuseColor / base_color                    */
  ORC_P_EMISSION = 3,          /* 3: emissiveColor / emission_color*luminance     */
  ORC_P_USE_SPECULAR_WORKFLOW = 6,
  ORC_P_SPECULAR_COLOR = 7,    /* 3 */
  ORC_P_METALLIC = 10,
  ORC_P_ROUGHNESS = 11,
  ORC_P_CLEARCOAT = 12,
  ORC_P_CLEARCOAT_ROUGHNESS = 13,
  ORC_P_OPACITY = 14,
  ORC_P_OPACITY_THRESHOLD = 15,
  ORC_P_IOR = 16,
  ORC_P_BASE_WEIGHT = 17,
  ORC_P_SPECULAR_WEIGHT = 18,
  ORC_P_COAT_COLOR = 19,       /* 3 */
  ORC_P_COAT_IOR = 22,
  ORC_P_TRANSMISSION_WEIGHT = 23,
  ORC_P_TRANSMISSION_COLOR = 24, /* 3 */
  ORC_P_DIFFUSE_ROUGHNESS = 27,
  ORC_P_TRANSMISSION_DEPTH = 28,
  ORC_P_TRANSMISSION_SCATTER = 29, /* 3 */
  ORC_P_TRANSMISSION_SCATTER_ANISOTROPY = 47, /* (32..46 hold the device's derived constants) */
  ORC_P_COAT_DARKENING = 48,   /* open_pbr_surface.mtlx:64 */
  ORC_P_FUZZ_WEIGHT = 49, ORC_P_FUZZ_COLOR = 50, ORC_P_FUZZ_ROUGHNESS = 53, /* :57-59: the fuzz (sheen) layer (:569-581) */
  ORC_P_SUBSURFACE_WEIGHT = 55, ORC_P_SUBSURFACE_COLOR = 56, ORC_P_SUBSURFACE_ANISOTROPY = 59, /* :43-52; thin-walled subsurface (:140-196) */
  ORC_P_SUBSURFACE_RADIUS = 32, ORC_P_SUBSURFACE_RADIUS_SCALE = 33, /* :47-50; the volumetric form (:182-192, 207-218), rendered with a medium stack */
  ORC_P_THIN_WALLED = 54,      /* geometry_thin_walled (:88) */
  ORC_P_THIN_FILM_IOR = 6, ORC_P_THIN_FILM_WEIGHT = 62, ORC_P_THIN_FILM_THICKNESS = 63, /* OpenPBR thin_film_* (:71-76); slot 6 is useSpecularWorkflow for UsdPreviewSurface */
  ORC_P_SPECULAR_ANISOTROPY = 60, ORC_P_COAT_ANISOTROPY = 61, /* specular_roughness_anisotropy (:27), coat_roughness_anisotropy (:65) */
  ORC_P_COAT_ROTATION = 36,    /* geometry_coat_tangent (:91, 561) as a document binds it: Tworld turned by this many turns towards the bitangent (rotate3d about the normal) */
  ORC_P_SPECULAR_ROTATION = 37, /* geometry_tangent (:89; 385 ... 457) in the same form: the tangent of the dielectric and conductor lobes, turned */
  ORC_P_COUNT = 64
};

/* Texture runtime (mdl_interface.glsl:8-38, 127-145; mdl_types.glsl:117-120).  Texels are linear float RGBA, row 0 first
 * (v = 0 side), as the harness decoded them; lookups are bilinear with REPEAT addressing (the reference's one sampler,
 * Gi.cpp:388-392, CgpuVk.cpp:1985-1990) after apply_wrap_and_crop. */
enum { ORC_TEX_WRAP_CLAMP = 0, ORC_TEX_WRAP_REPEAT = 1, ORC_TEX_WRAP_MIRRORED_REPEAT = 2, ORC_TEX_WRAP_CLIP = 3 };
typedef struct OrcTexture { const float* rgba; uint32_t width, height; } OrcTexture;
/* material inputs that can be driven by a texture (UsdUVTexture semantics: value = texel * scale + bias) */
enum { ORC_TEX_BASE_COLOR = 0, ORC_TEX_EMISSION = 1, ORC_TEX_ROUGHNESS = 2, ORC_TEX_METALLIC = 3, ORC_TEX_NORMAL = 4, ORC_TEX_OPACITY = 5 /* cutout opacity, read by the any-hit test */,
       ORC_TEX_COAT_NORMAL = 6 /* OpenPBR geometry_coat_normal (open_pbr_surface.mtlx:87, 560): a tangent-space normal map for the coat lobe's own shading frame */,
       ORC_TEX_TRANSMISSION_WEIGHT = 7, ORC_TEX_TRANSMISSION_COLOR = 8 /* OpenPBR transmission_weight / transmission_color (open_pbr_surface.mtlx:29, 31) at the hit; the MEDIUM a path enters
       keeps the material's constant colour (its absorption is derived once per material) */, ORC_TEX_SLOT_COUNT = 9 };
typedef struct OrcTexBinding {
  int32_t texture; /* index into OrcScene.textures; < 0 = input not textured */
  int32_t wrapS, wrapT;
  int32_t channel; /* scalar inputs: which channel of the (scaled, biased) texel */
  float scale[4], bias[4];
  int32_t hasTransform; /* UsdTransform2d upstream of the lookup's st (UsdPreviewSurface specification): s' = (xf[0] s + xf[1] t) + xf[2], t' = (xf[3] s + xf[4] t) + xf[5] */
  float xf[6];
} OrcTexBinding;

/* Scene data (primvars; mdl_interface.glsl:260-479, Gi.cpp:905-1019): a material input may read a named primvar of the mesh
 * (UsdPrimvarReader): float3 inputs use scene_data_lookup_float3, scalar inputs scene_data_lookup_float.  A texture on the same
 * input wins. */
enum { ORC_PRIMVAR_FLOAT = 0, ORC_PRIMVAR_VEC2, ORC_PRIMVAR_VEC3, ORC_PRIMVAR_VEC4, ORC_PRIMVAR_INT, ORC_PRIMVAR_INT2, ORC_PRIMVAR_INT3, ORC_PRIMVAR_INT4 }; /* Gi.h:76-79 */
enum { ORC_INTERP_CONSTANT = 0, ORC_INTERP_INSTANCE, ORC_INTERP_UNIFORM, ORC_INTERP_VERTEX }; /* Gi.h:81-84 */
typedef struct OrcPrimvar { char name[64]; int32_t type; int32_t interpolation; const float* data; uint32_t floatCount; } OrcPrimvar;

typedef struct OrcMaterial {
  uint32_t klass;
  uint32_t flags;
  float p[ORC_P_COUNT];
  OrcTexBinding tex[ORC_TEX_SLOT_COUNT];
  char primvarInput[ORC_TEX_SLOT_COUNT][64]; /* "" = input not driven by scene data */
} OrcMaterial;

typedef struct OrcMesh {
  const OrcVertex* vertices;
  uint32_t vertexCount;
  const uint32_t* faces; /* 3 indices per face */
  uint32_t faceCount;
  int32_t id;
  int32_t isDoubleSided;
  int32_t isLeftHanded;
  int32_t visible;
  float transform[16];            /* USD row-major, row-vector convention (Gi.cpp:641-650) */
  const float* instanceTransforms; /* instanceCount x 16, same convention (Gi.cpp:652-658)  */
  uint32_t instanceCount;
  int32_t material; /* index into OrcScene.materials */
  const int32_t* faceIds;     /* faceCount entries or NULL (GiMeshDesc.faceIds, Gi.h:127) */
  uint32_t maxFaceId;         /* GiMeshDesc.maxFaceId: chooses the 1/2/4-byte face-id stride (Gi.cpp:878-885) */
  const int32_t* instanceIds; /* instanceCount entries or NULL (Gi.cpp:660-670) */
  const OrcPrimvar* primvars; uint32_t primvarCount;                   /* GiMeshDesc.primvars */
  const OrcPrimvar* instancerPrimvars; uint32_t instancerPrimvarCount; /* giSetMeshInstancerPrimvars; mesh primvars override them (Gi.cpp:913-929) */
} OrcMesh;

/* Light descriptions at *setter* level (Gi.cpp:2573-2976); derived fields are computed by the oracle. */
typedef struct OrcSphereLight { float pos[3]; float baseEmission[3]; float radius[3]; float diffuse, specular; } OrcSphereLight;
typedef struct OrcDistantLight { float direction[3]; float baseEmission[3]; float angle; float diffuse, specular; } OrcDistantLight;
typedef struct OrcRectLight { float origin[3]; float t0[3]; float t1[3]; float baseEmission[3]; float width, height; float diffuse, specular; } OrcRectLight;
typedef struct OrcDiskLight { float origin[3]; float t0[3]; float t1[3]; float baseEmission[3]; float radiusX, radiusY; float diffuse, specular; } OrcDiskLight;

struct OrcDomeLight;
typedef struct OrcScene {
  const OrcMesh* meshes; uint32_t meshCount;
  const OrcMaterial* materials; uint32_t materialCount;
  const OrcSphereLight* sphereLights; uint32_t sphereLightCount;
  const OrcDistantLight* distantLights; uint32_t distantLightCount;
  const OrcRectLight* rectLights; uint32_t rectLightCount;
  const OrcDiskLight* diskLights; uint32_t diskLightCount;
  const OrcTexture* textures; uint32_t textureCount;
  const struct OrcDomeLight* dome; /* NULL: only the fallback dome (colour clear value) */
} OrcScene;

/* Dome light (Gi.cpp:2943-2976; rp_main.miss:38-86): equirectangular texture, rotation quaternion (x,y,z,w), emission multiplier */
typedef struct OrcDomeLight { int32_t texture; float rotation[4]; float baseEmission[3]; } OrcDomeLight;

/* == GiCameraDesc, Gi.h:96-108 */
typedef struct OrcCamera {
  float position[3];
  float forward[3];
  float up[3];
  float vfov;
  float fStop;
  float focusDistance;
  float focalLength;
  float clipStart;
  float clipEnd;
  float exposure;
} OrcCamera;

/* == GiRenderSettings, Gi.h:139-159 (bools as int32) */
typedef struct OrcSettings {
  int32_t clippingPlanes;
  int32_t depthOfField;
  int32_t domeLightCameraVisible;
  int32_t filterImportanceSampling;
  int32_t jitteredSampling;
  int32_t nextEventEstimation;
  int32_t progressiveAccumulation;
  uint32_t maxBounces;
  uint32_t rrBounceOffset;
  uint32_t spp;
  uint32_t sampleOffset; /* scene->sampleOffset (Gi.cpp:2411) */
  float lightIntensityMultiplier;
  float maxSampleValue;
  float rrInvMinTermProb;
  float metersPerSceneUnit;
  uint32_t mediumStackSize;     /* GiRenderSettings.mediumStackSize (Gi.h:151): 0 = inside/outside toggle only; <= 8 here */
  uint32_t maxVolumeWalkLength; /* Gi.h:150 */
  float clearColor[4]; /* Color AOV clear value == fallback dome colour (Gi.cpp:2184-2199) */
  float frame;         /* GiRenderSettings.frame (Gi.h:144) -> ubo.frame, the value of the FRAME scene data (mdl_interface.glsl:390-395) */
} OrcSettings;

typedef struct OrcRegion {
  uint32_t imageWidth, imageHeight; /* full image (RNG uses global pixel index) */
  uint32_t rowBegin, rowEnd;        /* rows [rowBegin,rowEnd) are rendered      */
} OrcRegion;

/* Non-colour AOVs (rp_main.rgen:132-183, 517-520; rp_main.chit:192-290).  Buffers are (rowEnd-rowBegin)*width elements: vec3
 * AOVs as 4 floats per pixel (std430 vec3[] stride; .w is never written), ids/depth as one int32/float.  NULL = not bound.
 * clear[id] is the binding's 16-byte clear value.  NEE (rp_main.rgen:431-435: outcome of the pixel's last traced shadow ray) and
 * Bounces (:483-486: inferno colour of the last sample's bounce count) need the full paths; ClockCycles is not produced. */
typedef struct OrcAovs {
  float* normal; float* barycentrics; float* texcoords; float* opacity; float* tangents; float* bitangents; float* thinWalled;
  float* doubleSided; float* albedo; float* depth;
  int32_t* objectId; int32_t* faceId; int32_t* instanceId;
  float clear[17][4];
  float* nee; float* bounces;
  float* clockCycles; /* cost proxy: ray segments of all samples of the pixel, heat-mapped (Turbo) against the region's maximum */
} OrcAovs;
float orc_atan2f(float y, float x);
float orc_acosf(float x);
void orc_tex_lookup(const OrcTexture* t, float u, float v, int wrapU, int wrapV, float* out4);
int orc_render_aovs(const OrcScene* scene, const OrcCamera* camera, const OrcSettings* settings, const OrcRegion* region, OrcAovs* aovs);

typedef struct OrcCounters {
  uint64_t samples;
  uint64_t segments;        /* closest-hit traces */
  uint64_t shadowRays;      /* shadow traces actually traced (traceRay == true) */
  uint64_t hits;
  uint64_t bounceHistogram[64]; /* [b] = #paths that traced a segment at bounce b */
} OrcCounters;

/* Renders rows [rowBegin,rowEnd) into colorOut (RGBA32F, (rowEnd-rowBegin)*width*4 floats,
 * row 0 = bottom, pixelIndex = x + y*width as in rp_main.rgen:195).  prevColor may be NULL
 * (needed only when sampleOffset>0 && progressiveAccumulation).  threads<=0 -> 1 thread.
 * Returns 0 on success. */
int orc_render(const OrcScene* scene, const OrcCamera* camera, const OrcSettings* settings,
               const OrcRegion* region, const float* prevColor, float* colorOut,
               OrcCounters* counters, int threads);
/* the same for an explicit list of image rows (output row r = image row rowList[r]; region gives the image size only) */
int orc_render_rows(const OrcScene* scene, const OrcCamera* camera, const OrcSettings* settings, const OrcRegion* region, uint32_t rowCount, const uint32_t* rowList,
                    const float* prevColor, float* colorOut, OrcCounters* counters, int threads);

/* ---- known-answer helpers (restated common.glsl / Gi.cpp pieces) ---- */
uint32_t orc_rng_init(uint32_t pixelIndex, uint32_t sampleIndex);     /* common.glsl:121-124 */
float    orc_rng_next1f(uint32_t* state);                            /* common.glsl:92-96   */
uint32_t orc_hash_pcg32(uint32_t* state);                            /* common.glsl:85-90   */
uint32_t orc_encode_direction(const float v[3]);                     /* Gi.cpp:287-300      */
void     orc_decode_direction(uint32_t e, float out[3]);             /* common.glsl:198-207 */
void     orc_offset_ray_origin(const float p[3], const float n[3], float out[3]); /* common.glsl:143-162 */
void     orc_fis_gauss(float xi0, float xi1, float out[2]);          /* rp_main.rgen:118-130 */
void     orc_sincos2pi(float x, float* s, float* c);
float    orc_expf(float x); /* x <= 0 */
float    orc_logf(float x);
float    orc_film_reflectance(float cosTheta, float filmIor, float substrateIor, float thicknessNm, float wavelengthNm); /* thin film: Airy reflectance, unpolarised */
float    orc_fresnel_dielectric(float cosTheta, float eta);
uint32_t orc_pack_half2x16(float a, float b);
void     orc_unpack_half2x16(uint32_t v, float out[2]);
void     orc_orthonormal_basis(const float n[3], float b1[3], float b2[3]); /* common.glsl:128-137 */
/* closest hit of a single ray against the scene (brute force); returns 1 on hit. */
int      orc_trace_closest(const OrcScene* scene, const float o[3], const float d[3], float tMin, float tMax,
                           float* t, float* u, float* v, uint32_t* instance, uint32_t* prim);

/* batch version: the scene is prepared once.  outTUV 3*count, outInstPrim 2*count (-1 on miss). Returns #hits. */
int      orc_trace_batch(const OrcScene* scene, uint32_t count, const float* origins, const float* dirs, float tMin, float tMax,
                         float* outTUV, int32_t* outInstPrim);
/* closed-form BSDF entry points on explicit shading frames.  in: 22 floats per item = normal, tangentU, tangentV,
 * geomNormal, k1, k2 (for evaluate), xi[4].  out: 15 floats = k2, bsdf_over_pdf, pdf, event, eval diffuse, eval glossy,
 * eval pdf. */
void     orc_bsdf_debug(const OrcMaterial* mat, uint32_t count, const float* in, float* out);

/* the MDL runtime's remaining texture entry points (mdl_interface.glsl:45-65, 86-105, 167-221), 8 floats per query -> 4 per result (see gi_oracle.cpp) */
void orc_tex_runtime(const float* rgba, uint32_t w, uint32_t h, uint32_t d, uint32_t count, const float* queries, float* out);
void orc_dbg_sample_trilinear(const float* rgba, uint32_t w, uint32_t h, uint32_t d, float u, float v, float ww, float* out4);
void orc_scene_data_lookup_float4x4(const float* defaultValue, float* out);
#ifdef __cplusplus
}
#endif
#endif
