// gi_types.h -- POD layouts shared by the host code (gi_host.h and the gi_*.cpp units, bvh8.cpp) and the HIP kernels.
// All structs are plain data in HBM; sizes are asserted.  See DESIGN.md "Data layout in HBM".
#pragma once

#include <stdint.h>

namespace gi {

// 8-wide quantised BVH node, 80 bytes (5 x 16 B loads).  Layout after Ylitie/Karras/Laine 2017
// ("Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs"), implemented from the paper.
struct Node8 {
  float p[3];        // quantisation origin (node AABB min)
  uint8_t e[3];      // per-axis exponent: scale_i = 2^(e_i - 127) (float exponent field)
  uint8_t imask;     // bit s set: slot s holds an internal node
  uint32_t childBase; // index of the first internal child node (internal children are contiguous, slot order)
  uint32_t triBase;   // index of the first triangle referenced by this node's leaf slots
  uint8_t meta[8];    // 0: empty; internal: (1<<5)|(24+slot); leaf: (unary count<<5)|triangle offset
  uint8_t qlo[3][8];  // quantised child AABB min, [axis][slot]
  uint8_t qhi[3][8];  // quantised child AABB max
};
static_assert(sizeof(Node8) == 80, "Node8 must be 80 bytes");

// Triangle record, 64 bytes.  Traversal reads the first 48 (3 x 16 B): world-space vertex + edges + tie-break id;
// the shade stage reads the last 32: everything needed to fetch the hit's geometry in ONE dependent step
// (instance, material, absolute vertex indices) instead of chasing instance -> mesh -> face -> vertices.
struct TriRec {
  float v0[3];
  float e1[3];
  float e2[3];
  uint32_t origId;    // global triangle id in scene order (tie-break key, DESIGN.md "Traversal contract")
  uint32_t instance;  // index into InstanceRec[]
  // material index (bits 0-23) | material class (bits 24-27) | bit 28: cutout opacity
  // < 1 | mesh flags << 30 (bit0 flipFacing, bit1 doubleSided; rp_main.h:115-116)
  uint32_t matFlags;
  uint32_t vi[3];     // absolute indices into the scene vertex array
  uint32_t prim;      // gl_PrimitiveID within the mesh
};
static_assert(sizeof(TriRec) == 64, "TriRec must be 64 bytes");

// Vertex record, 48 bytes (3 x 16 B).  Same content as rp::FVertex (rp_main.h:58-64) but with the octahedral
// unorm2x16 normal / tangent already DECODED (decode_direction, common.glsl:198-207, evaluated once on the host with
// the same fp32 operations the shader would execute per hit): the quantisation the reference applies is preserved
// bit-for-bit, the 6 decodes per hit are not repeated.
struct FVertex {
  float pos[3];
  float bsign;
  float normal[3];
  float u;
  float tangent[3];
  float v;
};
static_assert(sizeof(FVertex) == 48, "FVertex must be 48 bytes");

// Shading record of one MESH triangle (scenes beyond LDS, round 3): three 48-byte vertices at arbitrary indices are 3 - 4.5 cache lines per hit, this record is
// 160 bytes = two.  Until round 6 it was ONE 128-byte line with normals and tangents in the reference's octahedral unorm2x16 encoding (rp::FVertex,
// rp_main.h:58-64), decoded per hit; now they are stored DECODED, as FVertex holds them -- decoded once on the host with the operations of decode_direction
// (common.glsl:198-207), so both forms give the same bits -- because the shade stage is bound by its instruction count, not by its lines (DESIGN.md section 4,
// r05ea): six decodes were ~400 of the ~3 500 VALU instructions a wave of hits executes (two IEEE
// divisions, a square root and a third division each).  Shared by all instances of the mesh;
// TriRec::vi[0] holds its index.
struct TriShade {
  float p[3][3];       // object-space corner positions
  float n[3][3];       // decoded (unit) normals of the corners
  float t[3][3];       // decoded tangents
  float uv[3][2];      // texture coordinates
  float bsign[3];      // bitangent signs
  uint32_t vi[3];      // absolute vertex indices (scene-data lookups by vertex)
  uint32_t pad;
};
static_assert(sizeof(TriShade) == 160, "TriShade is ten 16-byte pieces");

// Replaces gl_ObjectToWorldEXT / gl_WorldToObjectEXT + BlasPayload (rp_main.h:118-123), 96 bytes
struct InstanceRec {
  float o2w[12]; // rows of the 3x4 object->world matrix
  float w2o[9];  // inverse of its 3x3 part, row-major
  uint32_t mesh;
  int32_t instanceId;
  uint32_t pad; // the mesh's object id (GiMeshDesc.id -> BlasPayloadBufferPreamble.objectId, rp_main.h:142-147)
};
static_assert(sizeof(InstanceRec) == 96, "InstanceRec must be 96 bytes");

constexpr uint32_t MAT_PARAM_COUNT = 64;
// derived per-material constants, filled by the host into the spare tail of MaterialRec::p (same fp32 formulas the
// oracle evaluates per hit)
// class 1 (UsdPreviewSurface): albedo, F0, alpha, coat, coatAlpha.  class 2 (OpenPBR): albedo = base_color*base_weight,
// F0 slot = metal edge tint (specular_color*specular_weight), alpha, coat, coatAlpha, coatF0, modulated eta, sigma_a
// MP_FEATURES (OpenPBR records only): the device copy of the thin-walled switch p[54] carries a small bit set as a float -- which optional lobes the material
// has -- so that k_shade loads the fuzz / anisotropy inputs only for materials that use them (the stage is bound by its scattered requests, not by arithmetic)
constexpr uint32_t MP_FEATURES = 54, MATF_THIN_WALLED = 1u, MATF_FUZZ = 2u, MATF_ANISOTROPY = 4u, MATF_THIN_FILM = 8u,
                   // not thin-walled, subsurface_weight > 0: the volumetric subsurface lobe
                   // (live in renders with a medium stack), coefficients in MaterialRec::sss
                   MATF_SSS_VOLUME = 16u,
                   // geometry_coat_tangent: an anisotropic coat whose tangent is turned (cosine / sine in MaterialRec::sss[6..7])
                   MATF_COAT_ROTATION = 32u,
                   // geometry_tangent: anisotropic base lobes whose tangent is turned: k_shade turns the shading frame's tangents in place before the
                   // BSDF runs (cosine / sine in sss[8..9]; sss[10..11] = the coat's turn relative to that frame, for coats without a frame of their own)
                   MATF_SPEC_ROTATION = 64u;
enum : uint32_t { MP_ALBEDO = 32, MP_F0 = 35, MP_ALPHA = 38, MP_COAT = 39, MP_COAT_ALPHA = 40, MP_COAT_F0 = 41, MP_ETA = 42, MP_SIGMA_A = 43,
    MP_CUTOUT = 46 /* mdl_cutout_opacity, 1 = opaque */ };
// Textured material inputs (UsdUVTexture semantics: value = texel * scale + bias at the hit's st).  Replaces the MDL
// renderer runtime's tex_lookup_* path (mdl_interface.glsl:127-145) for the inputs the closed-form materials expose.
enum : uint32_t { TEX_BASE_COLOR = 0, TEX_EMISSION = 1, TEX_ROUGHNESS = 2, TEX_METALLIC = 3, TEX_NORMAL = 4,
    TEX_OPACITY = 5 /* read by the any-hit test, not by k_shade */,
                  TEX_COAT_NORMAL = 6 /* OpenPBR geometry_coat_normal: the coat lobe's own shading frame */,
                  /* OpenPBR transmission_weight (scalar) / transmission_color (rgb: the surface tint of a medium-less, depth-0 transmission) */
                  TEX_TRANSMISSION_WEIGHT = 7, TEX_TRANSMISSION_COLOR = 8, TEX_SLOT_COUNT = 9 };
// the slots k_shade resolves per hit, in resolve order (TEX_OPACITY belongs to the any-hit test, TEX_COAT_NORMAL is resolved before the base normal)
constexpr uint32_t shade_slot(uint32_t k) { return k < TEX_OPACITY ? k : k + 2u; }
constexpr uint32_t SHADE_SLOT_COUNT = 7;
enum : uint32_t { TEX_WRAP_CLAMP = 0, TEX_WRAP_REPEAT = 1, TEX_WRAP_MIRRORED_REPEAT = 2, TEX_WRAP_CLIP = 3 }; // mdl_types.glsl:117-120
struct TexBindingRec {
  uint32_t tex;   // texture index + 1; 0 = input not textured
  uint32_t mode;  // wrapS | wrapT << 8 | channel << 16 | TEX_MODE_* flags
  float scale[4], bias[4];
  // TEX_MODE_XFORM: texture-coordinate transform of the lookup, s' = (xf[0] s + xf[1] t) +
  // xf[2], t' = (xf[3] s + xf[4] t) + xf[5] (UsdTransform2d upstream of a UsdUVTexture's `st`)
  float xf[6];
};
constexpr uint32_t MAT_FLAG_OPACITY_TEX = 1u << 30; // MaterialRec::flags: the cutout opacity is textured (the any-hit test looks it up at the candidate's st)
constexpr uint32_t MAT_FLAG_TEXTURED = 1u << 31; // MaterialRec::flags: some input is textured or primvar-driven (k_shade resolves the inputs per hit)
constexpr uint32_t TEX_MODE_PRIMVAR = 1u << 24;   // TexBindingRec::mode: (no texture) the input reads the mesh's scene data for this slot
// ... the scene-data name is "CAMERA_POSITION": ubo.cameraPosition (mdl_interface.glsl:329-334, Frontend.cpp:251)
constexpr uint32_t TEX_MODE_CAMERA_POSITION = 1u << 25;
constexpr uint32_t TEX_MODE_XFORM = 1u << 27;           // TexBindingRec::xf is not the identity
constexpr uint32_t TEX_MODE_FRAME = 1u << 26;           // ... the scene-data name is "FRAME": ubo.frame (mdl_interface.glsl:390-395, Frontend.cpp:252)
// MeshRec::sdInfo: integer primvar, nearest-vertex interpolation (scene_data_lookup_int, mdl_interface.glsl:426-457)
constexpr uint32_t SD_INFO_INT = 1u << 5;
// Scene data (primvars) of a mesh for the material inputs of ITS material (replaces BlasPayloadBufferPreamble::sceneDataInfos,
// rp_main.h:125-148, Gi.cpp:905-1019): per input slot the float offset into SceneView::sceneData and
// info = valid | (stride - 1) << 1 | interpolation << 3 (GiPrimvarInterpolation: constant, instance, uniform, vertex).
struct MeshRec { uint32_t vertexOffset; uint32_t pad; uint32_t sdOffset[TEX_SLOT_COUNT]; uint32_t sdInfo[TEX_SLOT_COUNT]; };
struct MaterialRec {
  uint32_t klass;
  uint32_t flags;
  float p[MAT_PARAM_COUNT];
  TexBindingRec tex[TEX_SLOT_COUNT];
  // volumetric subsurface medium (OpenPBR, derived on the host from subsurface_color / _radius / _radius_scale): sigma_s[3], sigma_t[3]; then the
  // cosine and the sine of the coat tangent's turn (MATF_COAT_ROTATION), of the base lobes' (MATF_SPEC_ROTATION), of the coat's relative to the latter
  float sss[12];
};
static_assert(sizeof(MaterialRec) == 888, "MaterialRec must be 888 bytes");
// A texture: linear float RGBA texels, row 0 first.  (8-bit sources are decoded to linear float by the caller; a
// compressed unorm8/half store is a later memory optimisation, the lookup arithmetic would not change.)
struct TextureRec { const float* texels; uint32_t width, height; };

// == rp::SphereLight / DistantLight / RectLight / DiskLight (rp_main.h:73-113), 48 bytes each
struct SphereLightRec { float pos[3]; uint32_t ds; float em[3]; float area; float radius[3]; float pad; };
struct DistantLightRec { float dir[3]; float angle; float em[3]; uint32_t ds; float pad[3]; float invPdf; };
struct RectLightRec { float origin[3]; float width; float em[3]; float height; uint32_t t0, t1, ds; float pad; };
struct DiskLightRec { float origin[3]; float rx; float em[3]; float ry; uint32_t t0, t1, ds; float pad; };
static_assert(sizeof(SphereLightRec) == 48 && sizeof(DistantLightRec) == 48 && sizeof(RectLightRec) == 48 && sizeof(DiskLightRec) == 48, "lights are 48 bytes");
// Derived per rect / disk light, beside its 48-byte record (round 6): the two tangents DECODED (host: decode_direction's operations, as for FVertex) and the
// light's normal cross(t1, t0) -- sample_light decoded both codes and took the cross product for every light sample of every hit (two IEEE divisions, a square
// root and a third division per decode).  Same bits: the host runs the device's operations in the device's order, without contraction.
struct LightFrame { float t0[3], pad0, t1[3], pad1, n[3], pad2; };
static_assert(sizeof(LightFrame) == 48, "LightFrame is three 16-byte pieces");

// Per-frame constants (replaces rp::UniformData, rp_main.h:25-56; derived camera terms are computed once on
// the host exactly as rp_main.rgen:199-212 does per pixel).
struct FrameUniforms {
  float camPos[3]; float WX;
  float camFwd[3]; float HY;
  float camUp[3]; float lensRadius;
  float camRight[3]; float focusDistance;
  float L[3]; float clipNear;
  float background[3]; float clipFar;
  float invSpp, sppF, sampleOffsetF, invTotalSampleCount;
  float maxSampleValue, rrInvMinTermProb, lightIntensityMultiplier, exposureScale;
  float metersPerSceneUnit, padf[3];
  uint32_t spp, sampleOffset, maxBounces, rrBounceOffset;
  uint32_t imageWidth, imageHeight, rowBegin, pixelCount; // pixelCount = pixels of this tile
  uint32_t batchFirstSample, batchSamples, workTotal, poolSlots; // this batch: samples [first, first+count) of every tile pixel
  uint32_t flags; // FLAG_*
  uint32_t sphereCount, distantCount, rectCount, diskCount, totalLightCount;
  uint32_t mediumStackSize, maxVolumeWalkLength; // GiRenderSettings (Gi.h:150-151); stack size 0 = inside/outside toggle only
  uint32_t rowStride, padStride;                 // the tile's rows are rowBegin + k * rowStride (multi-GPU row interleaving)
  float sceneLo[3], padLo, sceneHi[3], padHi;    // FLAG_BOUNDS_RETIRE: the root node's dequantised bounds, padded (host: sceneBoundsForRetire)
};
enum : uint32_t {
  FLAG_JITTER = 1u, FLAG_FIS = 2u, FLAG_DOF = 4u, FLAG_CLIP = 8u, FLAG_NEE = 16u, FLAG_PROGRESSIVE = 32u,
  FLAG_PIXEL_MAJOR = 64u, // work order of the wavefront pipeline (gi_queues.h work_item)
  // wavefront pipeline: k_raygen does not write the Slot of a new camera path; its (rng, work item) travel beside the ray record and the
  // slot is written when the first segment HITS (k_route / k_trace); a camera ray that leaves the scene retires without ever touching a slot
  FLAG_DEFER_SLOT = 128u,
  // the shadow walks of bounce i run on a second stream beside the closest-hit walks of bounce i + 1 (gi_render.cpp "two streams"): k_raygen runs AFTER
  // the iteration's k_trace / k_route and zeroes only what k_shade and the shadow launch append to; k_zero_closest zeroes the rest before k_trace
  FLAG_TWO_STREAM = 512u,
  // k_route / k_trace bin the hits of a specialised shade class with its full class (a thin batch -- one sample per
  // pixel and call -- pays more for a further k_shade launch per iteration than the variant saves: gi_render.cpp)
  FLAG_MERGE_SHADE_VARIANTS = 1024u,
  // with FLAG_DEFER_SLOT on the k_trace_dyn path: a camera ray whose slab interval against the scene bounds is empty is never
  // queued -- k_raygen retires its sample (the arithmetic of retire_fresh_miss) and hands the slot straight to the next k_raygen
  FLAG_BOUNDS_RETIRE = 256u,
};

// Device-side scene view handed to the kernels.
// Two-level layout records, each ONE cache line (r03: the traversal kernels are bound by the number of distinct lines a lane requests -- tools/ta_calib.hip --
// so a candidate costs two lines, its mesh triangle and its instance, instead of the six the first version touched: index record, three vertices, instance
// matrix, instance info):
// object-space corners (the mesh's vertex positions, bit for bit) + gl_PrimitiveID, 64 bytes
struct BlasTri { float p0[3], p1[3], p2[3]; uint32_t prim; uint32_t pad[6]; };
struct InstTrav {                                                               // what a walk needs of an instance, 128 bytes
  float o2w[12];                       // rows of the object->world matrix (candidates are rebuilt in world space with the host's xformPoint arithmetic)
  float w2o[9];                        // inverse of its 3x3 part (the ray enters the BLAS in object space)
  uint32_t blasRoot, triBase, matFlags; // BLAS root node, scene-order id of the instance's first triangle (flat numbering), TriRec::matFlags of its mesh
  float slack;                         // object-space magnitude the transformed ray's rounding error scales with
  uint32_t pad[7];
};
static_assert(sizeof(BlasTri) == 64 && sizeof(InstTrav) == 128, "two-level records are one cache line each");
struct alignas(16) F4 { float x, y, z, w; };
struct SceneView {
  const Node8* nodes;      // packed, 80 bytes apart (one node per 128-byte line measured 1-3 % slower, r03c)
  const TriRec* tris;
  const InstanceRec* instances;
  const FVertex* verts;
  // LDS-resident scenes (shadePacked == 0): the world-space geometric normal of
  // every flattened triangle, made on the host with setup_shading_state's operations
  const F4* triGeomNormal;
  // != 0: TriRec::vi[0] indexes triShade (scenes beyond LDS); 0: TriRec::vi are vertex indices (LDS-resident scenes, fused kernels)
  const TriShade* triShade; uint32_t shadePacked;
  const MaterialRec* materials;
  const SphereLightRec* sphereLights;
  const DistantLightRec* distantLights;
  const RectLightRec* rectLights;
  const DiskLightRec* diskLights;
  const LightFrame* rectFrames; const LightFrame* diskFrames; // decoded tangents + normal of rectLights[i] / diskLights[i]
  const int32_t* triFaceId; // per triangle (BVH order): the value the FaceId AOV shows (rp_main.chit:230-240)
  uint32_t nodeCount;
  uint32_t triCount;
  uint32_t bvhDepth; // levels of the BVH8 below the root = the most traversal-stack entries a ray can need
  uint32_t hasCutouts; // some triangle has cutout opacity < 1: traversal runs the any-hit test (needs the path rng)
  const TextureRec* textures;
  const MeshRec* meshes;     // indexed by InstanceRec::mesh
  const float* sceneData;    // all primvar arrays the bound materials read
  // dome light (rp_main.miss:38-86); domeTexture = index + 1 of the equirectangular image, 0 = fallback dome only
  uint32_t domeTexture;
  uint32_t domeCameraVisible; // GiRenderSettings.domeLightCameraVisible: primary rays see the dome image
  float domeRotation[4];
  float domeEmission[3];
  float background[3];        // the fallback dome texel: colour clear value as RGBA8 unorm (Gi.cpp:2194-2199)
  float cameraPosition[3];    // ubo.cameraPosition / ubo.frame for the CAMERA_POSITION / FRAME scene-data names
  float frame;
  uint32_t mediumStackSize;   // > 0: rays that end inside a medium scatter instead of leaving the scene (rp_main.miss:57-66)
  // Two-level layout for instanced scenes (k_trace_dyn2): a TLAS over instance bounds whose leaf references name instances, one BLAS
  // per mesh in OBJECT space shared by all its instances.  Candidates are still tested as WORLD-space triangles (rebuilt from the
  // object-space vertices with the instance transform, the host's own arithmetic), so results equal the flat layout's bit for bit.
  const Node8* tlasNodes; const uint32_t* tlasItems;   // leaf reference k of a TLAS node = instance tlasItems[triBase + offset]
  const Node8* blasNodes; const BlasTri* blasTris;     // all BLASes concatenated; Node8::childBase / triBase are absolute
  const InstTrav* instTrav;                            // per instance
  const uint32_t* flatOfOrig;                          // scene-order triangle id -> index into `tris` (BVH order)
  uint32_t twoLevel;                                   // 0: not built
};


// Device buffers of the bound non-colour AOVs (null = not bound) and their clear values by GiAovId (Gi.h:36-56).
struct AovTargets {
  F4* normal; F4* barycentrics; F4* texcoords; F4* opacity; F4* tangents; F4* bitangents; F4* thinWalled; F4* doubleSided; F4* albedo;
  float* depth; int32_t* objectId; int32_t* faceId; int32_t* instanceId;
  float clear[17][4];
};

// Path state that persists across stages: ONE 64-byte record per slot of the persistent path pool.  A slot carries one
// (pixel, sample) work item at a time; when its path ends k_raygen hands it the next item (work id = running base +
// position in the regen queue: no atomics), so the pool stays full until the work runs out, whatever the tile size.
// Everything that merely flows from one stage to the next (rays, hits, shadow rays) lives in the queues as records
// written/read in queue order (coalesced); only this record is gathered/scattered by slot index, and 64 B is one
// fabric request.  The default 64 Mi-slot pool is 4 GiB.
struct alignas(64) Slot {
  F4 thr;  // throughput.xyz, asfloat(bitfield)   (rp_main_payload.glsl:24-33)
  F4 rad;  // radiance.xyz, asfloat(rng state)
  F4 id;   // asfloat(pixel index inside the tile), asfloat(sample index inside the batch), asfloat(1 = a sample is in flight), -
  F4 pad;
};
static_assert(sizeof(Slot) == 64, "Slot must be 64 bytes");

// Medium stack of a path (MEDIUM_STACK_SIZE > 0; rp_main_payload.glsl:11-17, 37-40): per slot `mediaStride` floats =
// stack entries of 8 floats (ior, bias, sigma_s[3], sigma_t[3]) followed by walkSegmentPdf (3 floats + pad).
// the payload's medium index has four bits (rp_main_payload.glsl:4-5): 15 is the deepest stack the reference can address
constexpr uint32_t MEDIUM_FLOATS = 8, MAX_MEDIUM_STACK = 15;
constexpr uint32_t VOLUME_MISS = 0xfffffffeu; // "triangle" id of a hit record that is a scattering event inside a medium
// Debug AOVs that follow whole paths (only when bound): Bounces = inferno colour of the bounce count of the pixel's LAST
// sample (rp_main.rgen:483-486, written by k_raygen when that sample retires); NEE = outcome of the shadow test at bounce 0 of the
// pixel's last sample (rp_main.rgen:431-435; untraced = not shadowed): per tile pixel the maximum of (sample + 1) << 1 | shadowed over
// all bounce-0 outcomes (k_shade, k_trace<any>, primary misses), which k_resolve_nee turns into red / green.  neeKey is null unless
// the AOV is bound and next-event estimation is on.
struct PathState {
  Slot* slots; float* media; uint32_t mediaStride;
  unsigned long long* neeKey; uint32_t neeSampleBase; F4* bouncesAov;
  // ClockCycles AOV (cost proxy): per tile pixel, the ray segments of all its samples so far (k_raygen adds a path's count when it retires)
  uint32_t* pathSegments;
};

// Work queues.  Every queue is split into NSHARD segments (segment s holds records [s*cap, s*cap + count[q][s])):
// producers append to the segment of their block (blockIdx % NSHARD, i.e. one per XCD in dispatch order), so the append
// counters are NSHARD different words.  One device-scope atomic word sustains only ~88 updates/us on MI355X
// (MI355X_MICROARCH.md "dequeue"); an unsharded per-wave append made every stage atomic-bound.
//   TRACE_A/B : slot, a = (origin, tMin), b = (direction, tMax)          -- the ray record (36 B)
//   REGEN_A/B : slot | REGEN_MISSED                                       -- paths that ended (or left the scene)
//   HIT+class : slot, a = (t, u, v, triangle), b = (direction, -)         -- the hit record (36 B), one queue per material class
//   SHADOW    : slot, a = (origin, distance), b = (direction, -), c = (neeContrib, -)
// The A/B pairs alternate per iteration so that no counter has to be reset between a queue's consumer and its next
// producers (k_raygen zeroes the counters of the following iteration, see zero_next_counters).
// HIT is one queue per material class (the sort key between trace and shade): Q_HIT + klass
// ... more precisely per SHADE class: the material classes 0 .. 2 (diffuse, UsdPreviewSurface, OpenPBR with every lobe) and the specialised variants of a
// class.  The shade class of a triangle's material rides in bits 24-27 of TriRec::matFlags (and from there in the top four bits of a hit word);
// MaterialRec::klass stays the BSDF model.  SHADE_CLASS_OPBR_BASE: OpenPBR materials whose
// optional lobes are all absent (gi_shading.h "BASE variant", gi_build.cpp shadeClassOf).
constexpr uint32_t MAT_CLASS_COUNT = 4, SHADE_CLASS_OPBR_BASE = 3;
enum : uint32_t { Q_TRACE_A = 0, Q_TRACE_B = 1, Q_REGEN_A = 2, Q_REGEN_B = 3, Q_SHADOW = 4, Q_HIT = 5, Q_COUNT = Q_HIT + MAT_CLASS_COUNT };
constexpr uint32_t NSHARD = 8;
constexpr uint32_t NCURSOR = 8;
struct FreshRec { uint32_t rng, work; }; // beside a camera ray's record (FLAG_DEFER_SLOT): rng state after the camera draws, work item id of the batch
struct QueueSet {
  FreshRec* fresh[2];      // TRACE_A / TRACE_B only; valid where the queue's slot word carries TRACE_FRESH
  uint32_t* slot[Q_COUNT]; // each NSHARD * cap entries
  F4* a[Q_COUNT];          // null where the queue has no such field
  F4* b[Q_COUNT];
  F4* c[Q_COUNT];
  uint32_t cap;            // per-shard capacity
};
// Each append counter sits alone in its own 128-byte line: device-scope atomics are serialised per LINE at the memory
// side, so counters sharing a line would share the ~88 updates/us budget.
struct alignas(128) PaddedCounter { uint32_t v; uint32_t pad[31]; };
struct Counters {
  PaddedCounter count[Q_COUNT][NSHARD];
  PaddedCounter workBase[2]; // work items handed out before iteration parity p (k_raygen reads [p], writes [p^1])
  // k_trace_dyn: the launch's rays [0, n) are cut into NCURSOR equal ranges, each with its own cursor on its own 128-byte line (a
  // device-scope atomic on one line completes ~88 times per microsecond; 8 lines, 8x that); [0]: closest-hit queue, [1]: shadow queue
  PaddedCounter cursor[2][NCURSOR];
  unsigned long long segments, shadowRays, nodesVisited, trisTested, shadowNodesVisited, shadowTrisTested;
  // shadow walks in the two visiting orders (k_trace_dyn<any>: [0] near-to-far, [1] slot order): rays launched, and node visits summed per wave into one of 16
  // lines (the host picks the order a scene's shadow rays visit fewer nodes in, gi_render.cpp shadowOrder)
  unsigned long long shadowOrderRays[2];
  PaddedCounter shadowOrderSteps[2][16];
  uint32_t overflow; // set by block_append when a shard would run past its capacity (host sizing bug): giCRender fails loudly
  // k_path, counting builds only (GI_C_SCENE_OPTION_COUNT_TRAVERSAL): shader-clock cycles per phase summed over waves, lanes doing useful work per phase summed
  // over trips, trips -- [0] regeneration, [1] closest-hit traversal, [2] shading, [3] shadow ray + finish (GATLING_OPTIONS=phase_stats=1 prints them)
  unsigned long long phaseCycles[4], phaseLanes[4], phaseTrips;
  // k_trace_dyn's closest-hit launches, counting builds only: [0] steps (loop trips of all waves), lanes per step that [1] hold a ray, [2] walk (run the node
  // test), [3] wait for the triangle ring (drained walk, pairs pending); [4] triangle
  // batches, [5] pairs in them, [6] steps in which some lane was refilled, [7] lanes refilled
  unsigned long long dynStats[8];
};

} // namespace gi
