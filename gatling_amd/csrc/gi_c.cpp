// gi_c.cpp -- host side of the MI355X-native gi core: scene containers, dirty flags, host packing, BVH build,
// uploads, the wavefront bounce loop, render buffers.  Implements include/gi_c.h.
//
// Restates the host logic of /root/reference/src/gi/impl/Gi.cpp behind the same API shape:
//   giCreateMesh/giSetMesh* (:620-782)          -> MeshData + dirty flags
//   _giBuildGeometryStructures/_giCreateBvh (:784-1315) -> flatten instances, pack FVertex, build BVH8, upload
//   giRender (:1989-2524)                        -> dirty handling, uniforms (:2373-2426), bounce loop, D2H
//   light setters (:2573-2976)                   -> CPU mirrors of the 48-byte device structs, dense stores
//   render buffers (:2978-3006)
// GPU plumbing (src/cgpu, src/ggpu in the reference) is the HIP runtime: hipMalloc / hipMemcpyAsync / streams.

#include "../../include/gi_c.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bvh8.h"
#include "gi_kernels.h"
#include "gi_image.h"
#include "gi_options.h"
#include "gi_types.h"

using namespace gi;

// ---------------------------------------------------------------------------------------------------------------
// global state (one giCInitialize per process, like Gi.cpp:244-259)
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr bool WORK_ORDER_PIXEL_MAJOR_DEFAULT = true;  // (GATLING_OPTIONS work_order) FLAG_PIXEL_MAJOR, gi_queues.h work_item

thread_local std::string t_lastError;
void setError(const std::string& e) { t_lastError = e; fprintf(stderr, "[gatling_gi] error: %s\n", e.c_str()); }

#define HIP_TRY(expr)                                                                                      \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) { setError(std::string(#expr) + ": " + hipGetErrorString(_e)); return GI_C_ERROR; } \
  } while (0)

// One entry per HIP device the library renders on (giCInitializeDevices / $GATLING_DEVICES; giCInitialize: one).  devs[0] is the PRIMARY device:
// render buffers, textures and every single-device entry point live there; the others hold replicas of the scene and render row shares.
struct DevCtx { int device = 0; int cuCount = 256; hipStream_t stream = nullptr; hipStream_t stream2 = nullptr; /* the shadow launches of two-stream batches (renderOnDevice "two streams") */
                int peer = 1; /* 1: the primary device and this one address each other's memory (peer access enabled both ways, or the same physical device); 0: no peer access --
                                 the device's row shares travel through pinned host memory; -1: hipDeviceCanAccessPeer / EnablePeerAccess failed with an error */ };
// One host thread per further device, created with the first multi-device render and kept until giCTerminate (a frame's share is handed to it as a job; until
// r04 every frame created and joined its own std::threads).  The thread binds its HIP device once.
struct DeviceWorker {
  std::thread th; std::mutex m; std::condition_variable cv;
  std::function<void()> job; bool busy = false, stop = false;
  void start() { th = std::thread([this] { std::unique_lock<std::mutex> lk(m); for (;;) { cv.wait(lk, [this] { return busy || stop; }); if (stop) return; lk.unlock(); job(); lk.lock(); busy = false; cv.notify_all(); } }); }
  void post(std::function<void()> fn) { { std::lock_guard<std::mutex> lk(m); job = std::move(fn); busy = true; } cv.notify_all(); }
  void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return !busy; }); }
  void shutdown() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_all(); if (th.joinable()) th.join(); }
  ~DeviceWorker() { shutdown(); } // (a process that exits without giCTerminate must not meet a joinable std::thread in a static destructor)
};
struct Context {
  bool initialized = false;
  int device = 0;               // == devs[0].device
  int cuCount = 256;            // == devs[0].cuCount
  hipStream_t stream = nullptr; // == devs[0].stream
  std::vector<DevCtx> devs;
  std::vector<std::unique_ptr<DeviceWorker>> workers; // [slot - 1], made on demand (renderOnDevices)
  std::mutex workerMutex;   // one multi-device frame at a time owns the workers (two scenes may render concurrently)
  std::mutex resourceMutex; // GPU resource destruction from sync threads (Gi.cpp:679-683)
} g_ctx;

double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr int GI_C_OUT_OF_MEMORY_INTERNAL = -77; // DeviceBuffer::alloc: hipErrorOutOfMemory (never returned through the C ABI)
template <typename T>
struct DeviceBuffer {
  T* ptr = nullptr;
  size_t count = 0;
  int alloc(size_t n)
  {
    if (n <= count && ptr) return GI_C_OK;
    release();
    if (n == 0) n = 1;
    const hipError_t e = hipMalloc((void**)&ptr, n * sizeof(T));
    if (e != hipSuccess) {
      ptr = nullptr;
      // out of memory is an answer the render loop acts on (more batches, a smaller pool: renderOnDevice), not yet an error
      if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); t_lastError = "hipMalloc: out of memory"; return GI_C_OUT_OF_MEMORY_INTERNAL; }
      setError(std::string("hipMalloc: ") + hipGetErrorString(e)); return GI_C_ERROR;
    }
    count = n;
    return GI_C_OK;
  }
  size_t bytes() const { return ptr ? count * sizeof(T) : 0; }
  int upload(const std::vector<T>& v, hipStream_t s)
  {
    if (const int rc = alloc(v.size())) { if (rc == GI_C_OUT_OF_MEMORY_INTERNAL) setError("hipMalloc: out of device memory (" + std::to_string(v.size() * sizeof(T)) + " bytes)"); return GI_C_ERROR; }
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return GI_C_OK;
  }
  void release() { if (ptr) { (void)hipFree(ptr); ptr = nullptr; count = 0; } }
};

// glm::packHalf2x16 (round to nearest even)
uint16_t f32ToF16(float f)
{
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u, absx = x & 0x7fffffffu;
  if (absx >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u : 0u));
  if (absx >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (absx < 0x33000001u) return (uint16_t)sign;
  int exp = (int)(absx >> 23) - 127;
  uint32_t man = (absx & 0x7fffffu) | 0x800000u;
  if (exp < -14) {
    int shift = -14 - exp + 13;
    uint32_t r = man >> shift, rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    return (uint16_t)(sign | r);
  }
  uint32_t r = ((uint32_t)(exp + 15) << 10) | ((man >> 13) & 0x3ffu), rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
  return (uint16_t)(sign | r);
}
float f16ToF32(uint16_t h)
{
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, out;
  if (exp == 0) { float v = (float)man * 5.9604644775390625e-8f; return sign ? -v : v; }
  if (exp == 31) out = sign | 0x7f800000u | (man << 13); else out = sign | ((exp + 112u) << 23) | (man << 13);
  float f; memcpy(&f, &out, 4); return f;
}
uint32_t packHalf2x16(float a, float b) { return (uint32_t)f32ToF16(a) | ((uint32_t)f32ToF16(b) << 16); }

// _EncodeDirection, Gi.cpp:287-300 (glm::packUnorm2x16 rounds)
uint32_t encodeDirection(const float* vin)
{
  float x = vin[0], y = vin[1], z = vin[2];
  float inv = 1.0f / sqrtf((x * x + y * y) + z * z);
  x *= inv; y *= inv; z *= inv;
  float s = fabsf(x) + fabsf(y) + fabsf(z);
  x /= s; y /= s; z /= s;
  float px = x >= 0.0f ? 1.0f : -1.0f, py = y >= 0.0f ? 1.0f : -1.0f, ex, ey;
  if (z < 0.0f) { ex = (1.0f - fabsf(y)) * px; ey = (1.0f - fabsf(x)) * py; } else { ex = x; ey = y; }
  ex = ex * 0.5f + 0.5f; ey = ey * 0.5f + 0.5f;
  ex = std::min(std::max(ex, 0.0f), 1.0f); ey = std::min(std::max(ey, 0.0f), 1.0f);
  return (uint32_t)nearbyintf(ex * 65535.0f) | ((uint32_t)nearbyintf(ey * 65535.0f) << 16);
}

// decode_direction (common.glsl:198-207) evaluated once per vertex on the host, operation for operation what
// gi_decode_direction / the oracle execute (IEEE fp32, no contraction), so results stay bit-identical.
void decodeDirection(uint32_t e, float out[3])
{
  float ex = (float)(e & 0xffffu) / 65535.0f, ey = (float)(e >> 16) / 65535.0f;
  ex = ex * 2.0f - 1.0f; ey = ey * 2.0f - 1.0f;
  float x = ex, y = ey, z = 1.0f - fabsf(ex) - fabsf(ey);
  float t = (-z > 0.0f) ? -z : 0.0f;
  x += (x >= 0.0f) ? -t : t;
  y += (y >= 0.0f) ? -t : t;
  float inv = 1.0f / sqrtf((x * x + y * y) + z * z);
  out[0] = x * inv; out[1] = y * inv; out[2] = z * inv;
}

// Turbo colour map (A. Mikhailov's polynomial fit of Google's Turbo look-up table; the reference indexes the 256-entry table,
// Gi.cpp:338-341).  Same association as the oracle's turbo_colormap.
void turboColormap(float x, float* rgb)
{
  const float x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x;
  rgb[0] = (((0.13572138f + 4.61539260f * x) + -42.66032258f * x2) + 132.13108234f * x3) + (-152.94239396f * x4 + 59.28637943f * x5);
  rgb[1] = (((0.09140261f + 2.19418839f * x) + 4.84296658f * x2) + -14.18503333f * x3) + (4.27729857f * x4 + 2.82956604f * x5);
  rgb[2] = (((0.10667330f + 12.64194608f * x) + -60.58204836f * x2) + 110.36276771f * x3) + (-89.90310912f * x4 + 27.34824973f * x5);
}

// per-material constants of the closed-form BSDFs (DESIGN.md "Materials"); same fp32 formulas as the oracle's ups_params
float cutoutOpacity(const MaterialRec& m) // same rule as the oracle's cutout_opacity
{
  float op = m.p[GI_C_P_OPACITY], th = m.p[GI_C_P_OPACITY_THRESHOLD];
  float cl = op > 0.0f ? op : 0.0f; cl = cl < 1.0f ? cl : 1.0f;
  if (m.klass == GI_C_MAT_OPEN_PBR) return cl;
  if (th > 0.0f) return (op >= th) ? 1.0f : 0.0f;
  return cl;
}

void deriveMaterialConstants(MaterialRec& m)
{
  const float cutout = cutoutOpacity(m);
  struct SetCutout { MaterialRec& m; float v; ~SetCutout() { m.p[MP_CUTOUT] = v; } } setCutout{m, cutout};
  const float* p = m.p;
  if (m.klass == GI_C_MAT_OPEN_PBR) { // same fp32 formulas as the oracle's opbr_params (open_pbr_surface.mtlx:306-373)
    float bw = p[GI_C_P_BASE_WEIGHT], sw = p[GI_C_P_SPECULAR_WEIGHT];
    float r = p[GI_C_P_ROUGHNESS], cr = p[GI_C_P_CLEARCOAT_ROUGHNESS], coat = p[GI_C_P_CLEARCOAT];
    float cior = p[GI_C_P_COAT_IOR], ior = p[GI_C_P_IOR];
    float qc = (cior - 1.0f) / (cior + 1.0f);
    float ratio = ior / cior, inv = cior / ior;
    float etaCoated = (ratio > 1.0f) ? ratio : inv;
    float etaS = etaCoated * coat + ior * (1.0f - coat);
    float q = (etaS - 1.0f) / (etaS + 1.0f);
    float f0 = sw * (q * q); f0 = f0 > 0.0f ? f0 : 0.0f; f0 = f0 < 0.99999f ? f0 : 0.99999f;
    float eps = ((etaS - 1.0f) > 0.0f ? 1.0f : ((etaS - 1.0f) < 0.0f ? -1.0f : 0.0f)) * sqrtf(f0);
    float depth = p[GI_C_P_TRANSMISSION_DEPTH];
    float out[MAT_PARAM_COUNT]; memcpy(out, p, sizeof(out));
    for (int i = 0; i < 3; i++) {
      out[MP_ALBEDO + i] = p[GI_C_P_BASE_COLOR + i] * bw;
      out[MP_F0 + i] = p[GI_C_P_SPECULAR_COLOR + i] * sw;
    }
    { // dielectric base VDF (open_pbr_surface.mtlx:220-298): absorption = extinction - scattering, shifted to be non-negative
      float ab[3];
      for (int i = 0; i < 3; i++) {
        float tc = p[GI_C_P_TRANSMISSION_COLOR + i]; tc = tc > 1e-6f ? tc : 1e-6f;
        const float ext = (depth > 0.0f) ? -logf(tc) / depth : 0.0f;
        const float sc = (depth > 0.0f) ? p[GI_C_P_TRANSMISSION_SCATTER + i] / depth : 0.0f;
        ab[i] = ext - sc;
      }
      float mn = ab[0] < ab[1] ? ab[0] : ab[1]; mn = mn < ab[2] ? mn : ab[2];
      for (int i = 0; i < 3; i++) out[MP_SIGMA_A + i] = (depth > 0.0f) ? ((0.0f > mn) ? ab[i] - mn : ab[i]) : 0.0f;
    }
    { // coat roughening (open_pbr_surface.mtlx:101-131), same fp32 operations as opbr_effective_roughness
      const float c4 = (cr * cr) * (cr * cr), r4 = (r * r) * (r * r);
      float t = 2.0f * c4 + r4; t = t < 1.0f ? t : 1.0f;
      const float ra = sqrtf(sqrtf(t));
      r = ra * coat + r * (1.0f - coat);
    }
    out[MP_ALPHA] = (r * r > 0.001f) ? r * r : 0.001f; out[MP_COAT] = coat; out[MP_COAT_ALPHA] = (cr * cr > 0.001f) ? cr * cr : 0.001f;
    out[MP_COAT_F0] = qc * qc; out[MP_ETA] = (1.0f + eps) / (1.0f - eps);
    // which optional lobes the material has, in the slot of the thin-walled switch: materials without them do not load their inputs per hit (gi_types.h MP_FEATURES)
    { // volumetric subsurface medium (oracle opbr_params "Volumetric subsurface", the same fp32 operations): extinction 1 / (radius * scale), albedo 1 - s^2
      for (int i = 0; i < 3; i++) {
        float c = p[GI_C_P_SUBSURFACE_COLOR + i]; c = c > 0.0f ? c : 0.0f; c = c < 1.0f ? c : 1.0f;
        const float sq = sqrtf((9.59217f + 41.6808f * c) + (17.7126f * c) * c);
        const float sv = (4.09712f + 4.20863f * c) - sq;
        float alb = 1.0f - sv * sv; alb = alb > 0.0f ? alb : 0.0f; alb = alb < 1.0f ? alb : 1.0f;
        float rr = p[GI_C_P_SUBSURFACE_RADIUS] * p[GI_C_P_SUBSURFACE_RADIUS_SCALE + i]; rr = rr > 1e-6f ? rr : 1e-6f;
        m.sss[3 + i] = 1.0f / rr; m.sss[i] = alb * m.sss[3 + i];
      }
      m.sss[6] = m.sss[7] = 0.0f;
    }
    out[MP_FEATURES] = (float)((p[GI_C_P_THIN_WALLED] != 0.0f ? MATF_THIN_WALLED : 0u) | (p[GI_C_P_FUZZ_WEIGHT] > 0.0f ? MATF_FUZZ : 0u) |
                               ((p[GI_C_P_THIN_WALLED] == 0.0f && p[GI_C_P_SUBSURFACE_WEIGHT] > 0.0f) ? MATF_SSS_VOLUME : 0u) |
                               ((p[GI_C_P_SPECULAR_ANISOTROPY] > 0.0f || p[GI_C_P_COAT_ANISOTROPY] > 0.0f) ? MATF_ANISOTROPY : 0u) |
                               (p[GI_C_P_THIN_FILM_WEIGHT] > 0.0f ? MATF_THIN_FILM : 0u));
    memcpy(m.p, out, sizeof(out));
    return;
  }
  float r = p[GI_C_P_ROUGHNESS], cr = p[GI_C_P_CLEARCOAT_ROUGHNESS];
  float alpha = (r * r > 0.001f) ? r * r : 0.001f, coatAlpha = (cr * cr > 0.001f) ? cr * cr : 0.001f;
  float albedo[3], F0[3];
  if (p[GI_C_P_USE_SPECULAR_WORKFLOW] != 0.0f) {
    for (int i = 0; i < 3; i++) { F0[i] = p[GI_C_P_SPECULAR_COLOR + i]; albedo[i] = p[GI_C_P_BASE_COLOR + i]; }
  } else {
    float ior = p[GI_C_P_IOR], metal = p[GI_C_P_METALLIC];
    float q = (1.0f - ior) / (1.0f + ior), f0 = q * q;
    for (int i = 0; i < 3; i++) { F0[i] = f0 * (1.0f - metal) + p[GI_C_P_BASE_COLOR + i] * metal; albedo[i] = p[GI_C_P_BASE_COLOR + i] * (1.0f - metal); }
  }
  for (int i = 0; i < 3; i++) { m.p[MP_ALBEDO + i] = albedo[i]; m.p[MP_F0 + i] = F0[i]; }
  m.p[MP_ALPHA] = alpha; m.p[MP_COAT] = p[GI_C_P_CLEARCOAT]; m.p[MP_COAT_ALPHA] = coatAlpha;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------
// handle types
// ---------------------------------------------------------------------------------------------------------------
// DIRTY_XFORM: only transforms of meshes that are part of the built scene changed -- the incremental path (updateTransforms) handles it unless a full
// rebuild is due anyway.  The reference keeps each mesh's BLAS and rebuilds the TLAS (Gi.cpp:1180-1202).
enum DirtyFlags : uint32_t { DIRTY_BVH = 1u, DIRTY_FRAMEBUFFER = 2u, DIRTY_LIGHTS = 4u, DIRTY_MATERIALS = 8u, DIRTY_ALL = 0xfu, DIRTY_XFORM = 16u };

struct GiCTexture { GiCScene* scene; uint32_t width, height; std::vector<float> rgba; std::string cacheKey; uint32_t refs = 1; };
struct GiCPrimvar { std::string name; int32_t type, interpolation; std::vector<float> data; };
struct GiCMaterial { GiCScene* scene; std::string name; GiCMaterialDesc desc; GiCTextureBinding tex[GI_C_TEX_SLOT_COUNT] = {}; std::string primvarInput[GI_C_TEX_SLOT_COUNT];
                     float texXf[GI_C_TEX_SLOT_COUNT][6] = {}; bool hasTexXf[GI_C_TEX_SLOT_COUNT] = {}; /* giCSetMaterialTextureTransform */ };

struct GiCMesh {
  GiCScene* scene;
  std::string name;
  std::vector<GiCVertex> vertices;
  std::vector<GiCFace> faces;
  std::vector<int32_t> faceIds;
  int32_t id = 0;
  bool doubleSided = false, flipFacing = false, visible = true;
  uint32_t maxFaceId = 0;
  float transform[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::vector<float> instanceTransforms; // 16 per instance; empty until giCSetMeshInstanceTransforms (as in Gi.cpp:620-638)
  std::vector<int32_t> instanceIds;
  std::vector<GiCPrimvar> primvars, instancerPrimvars;
  GiCMaterial* material = nullptr;
  bool xformDirty = false;  // transform / instance transforms changed since the last build or update ...
  std::vector<uint8_t> instDirty; // ... and which instances (empty: all of them)
  uint32_t builtInstances = 0xffffffffu; // instance count the built scene holds for this mesh (0xffffffff: not part of it)
};

// swap-remove dense store (GgpuDenseDataStore, src/ggpu/impl/DenseDataStore.cpp:35-93): the arrays stay dense so
// the *LightCount uniforms are the live counts.
template <typename Rec, typename Handle>
struct DenseStore {
  std::vector<Rec> recs;
  std::vector<Handle*> owners;
  uint32_t add(Handle* h, const Rec& r) { recs.push_back(r); owners.push_back(h); return (uint32_t)recs.size() - 1u; }
  void remove(uint32_t idx);
};

struct GiCSphereLight { GiCScene* scene; uint32_t index; };
struct GiCDistantLight { GiCScene* scene; uint32_t index; };
struct GiCRectLight { GiCScene* scene; uint32_t index; };
struct GiCDiskLight { GiCScene* scene; uint32_t index; };
struct GiCDomeLight { GiCScene* scene; std::string filePath; GiCTexture* texture = nullptr; bool ownsTexture = false; float rotation[4] = {0, 0, 0, 1}; float baseEmission[3] = {1, 1, 1}; float diffuse = 1.0f, specular = 1.0f; };

template <typename Rec, typename Handle>
void DenseStore<Rec, Handle>::remove(uint32_t idx)
{
  uint32_t last = (uint32_t)recs.size() - 1u;
  if (idx != last) { recs[idx] = recs[last]; owners[idx] = owners[last]; owners[idx]->index = idx; }
  recs.pop_back(); owners.pop_back();
}

struct GiCRenderBuffer {
  uint32_t width, height, stride;
  size_t size;
  void* deviceMem = nullptr; // on the primary device
  void* hostMem = nullptr; // pinned (hipHostMalloc): the reference maps a HostVisible|HostCached buffer (Gi.cpp:2019-2031)
  bool deviceOnly = false;
  bool scratch = false; // internal stand-in that lives in the rendering device's own scratch memory (no replicas)
  std::vector<void*> replicaMem; // [slot - 1]: the same buffer on the other devices (multi-device renders), allocated on first use
  void* stageMem = nullptr; // pinned, rb->size: where the row shares of devices WITHOUT peer access to the primary pass through (allocated on first use)
};

// Everything a scene keeps in ONE device's memory: the scene arrays, the path pool, the queues, the per-render scratch.  GiCScene IS the primary
// device's (inheritance keeps the single-device code reading `s->dNodes`); multi-device renders add one replica per further device.
struct SceneDevice {
  uint32_t slot = 0; // index into g_ctx.devs
  DeviceBuffer<MeshRec> dMeshes; DeviceBuffer<float> dSceneData;
  std::vector<DeviceBuffer<float>*> dTexels; DeviceBuffer<TextureRec> dTextures; // device copies (rebuilt with the materials)
  DeviceBuffer<Node8> dNodes; DeviceBuffer<TriRec> dTris; DeviceBuffer<InstanceRec> dInstances;
  DeviceBuffer<FVertex> dVerts; DeviceBuffer<MaterialRec> dMaterials; DeviceBuffer<int32_t> dTriFaceId; DeviceBuffer<TriShade> dTriShade;
  DeviceBuffer<SphereLightRec> dSphere; DeviceBuffer<DistantLightRec> dDistant; DeviceBuffer<RectLightRec> dRect; DeviceBuffer<DiskLightRec> dDisk;
  DeviceBuffer<Node8> dTlasNodes, dBlasNodes; DeviceBuffer<uint32_t> dTlasItems, dFlatOfOrig; DeviceBuffer<BlasTri> dBlasTris; DeviceBuffer<InstTrav> dInstTrav;
  // path state
  DeviceBuffer<Slot> slots;
  DeviceBuffer<float> media; // per-slot medium stack + walkSegmentPdf (mediumStackSize > 0)
  DeviceBuffer<F4> scratchColor; DeviceBuffer<unsigned long long> neeKey; // NEE / Bounces AOVs bound without / with the colour AOV
  DeviceBuffer<uint32_t> pathSegments;                                    // ClockCycles AOV (cost proxy)
  DeviceBuffer<F4> sampleBuf; // per-sample colours of the current batch (rgb, -): [pixel][sample] under the pixel-major work order of the stage kernels, [sample][pixel] otherwise (gi_queues.h sample_record)
  DeviceBuffer<F4> accum;        // per-pixel running sum across batches
  DeviceBuffer<uint32_t> qSlot[Q_COUNT]; // NSHARD segments of queueCap records each
  DeviceBuffer<F4> qA[Q_COUNT], qB[Q_COUNT], qC[Q_COUNT];
  DeviceBuffer<FreshRec> qFresh[2]; // beside TRACE_A / TRACE_B: (rng, work item) of camera rays whose Slot is written only when they hit (FLAG_DEFER_SLOT)
  uint32_t queueCap = 0;
  DeviceBuffer<Counters> dCounters;
  Counters* hCounters = nullptr; // pinned
  uint64_t memTotalMb = 0;       // the device's memory (hipMemGetInfo, asked once): sizes the default sample-buffer budget
  static constexpr uint32_t POLL_RING = 4, POLL_LAG = 2; // drain test of the bounce loop: iteration it reads the queue sizes of iteration it - POLL_LAG (giCRenderImpl)
  PaddedCounter* hPoll = nullptr; hipEvent_t pollEvent[POLL_RING] = {}; // pinned ring of queue-size snapshots + their completion events
  hipEvent_t evShade = nullptr, evShadow = nullptr; // two-stream batches: k_shade(i) done (the second stream's shadow launch waits for it) / shadow launch (i) done (k_raygen(i + 1) waits for it)
  GiCRenderStats stats{};
  std::vector<hipEvent_t> eventPool;
  void releaseAll();
};

// The scene as host arrays (built once per scene change, kept for incremental transform updates) ...
struct TwoLevelHost { std::vector<Node8> tlasNodes, blasNodes; std::vector<uint32_t> tlasItems; std::vector<BlasTri> blasTris; std::vector<InstTrav> instTrav; };
struct MeshBuild { const GiCMesh* m; uint32_t vertexOffset, matFlags, instFirst, instCount, triFirst; uint32_t meshIdx; std::vector<int32_t> faceIdAov; uint32_t shadeBase = 0; };
// One flattened mesh instance of a PARTITIONED scene (after the first transform edit): its own subtree in its own node range, its triangles in its own
// (scene-order) range, joined by a top tree over the subtree roots (bvh8.h buildTopBvh8).  Moving it rebuilds these ranges and the top tree only.
struct InstPart { uint32_t meshBuild, instInMesh; uint32_t triFirst, nf; uint32_t nodeOff, nodeCount, nodeCap, depth; float box[6]; };
struct SceneHost {
  std::vector<FVertex> verts; std::vector<InstanceRec> instances; std::vector<MaterialRec> mats; std::vector<MeshRec> meshRecs; std::vector<float> sceneData;
  Bvh8 bvh; std::vector<int32_t> triFaceId; std::vector<uint32_t> flatOfOrig; TwoLevelHost two;
  std::vector<MeshBuild> meshBuilds;
  std::vector<TriShade> triShade; bool shadePacked = false; // one-line shading records per mesh triangle (scenes beyond LDS): TriRec::vi[0] indexes them
  bool partitioned = false; std::vector<InstPart> parts; uint32_t topCap = 0; // partitioned layout: nodes [0, topCap) = top tree, then the parts' ranges
};

struct GiCScene : SceneDevice {
  std::mutex mutex;
  uint32_t dirty = DIRTY_ALL;
  std::vector<GiCMesh*> meshes;       // creation order (deterministic triangle ids; the reference uses an unordered_set)
  std::vector<GiCMaterial*> materials;
  std::vector<GiCTexture*> textures;  // creation order
  DenseStore<SphereLightRec, GiCSphereLight> sphereLights;
  DenseStore<DistantLightRec, GiCDistantLight> distantLights;
  DenseStore<RectLightRec, GiCRectLight> rectLights;
  DenseStore<DiskLightRec, GiCDiskLight> diskLights;
  uint32_t sampleOffset = 0;
  bool haveOldParams = false;
  GiCCameraDesc oldCamera{};
  GiCRenderSettings oldSettings{};
  uint8_t oldClear[GI_C_MAX_AOV_COMP_SIZE] = {0};
  uint32_t oldRowBegin = 0, oldRowEnd = 0, oldRowStride = 1;
  GiCDomeLight* oldDome = nullptr;
  float oldDomeEmission[3] = {0, 0, 0};
  // the scene as built (the same on every device)
  uint32_t nodeCount = 0, triCount = 0, bvhDepth = 0;
  float bounds[6] = {0, 0, 0, 0, 0, 0}; bool boundsValid = false; // the flat tree's root bounds (nodeBounds + a relative pad), for FLAG_BOUNDS_RETIRE
  bool twoLevel = false; int optTwoLevel = -1; // 1: build and use the two-level layout (scenes beyond LDS); otherwise the flat one
  bool hasCutouts = false;
  bool shadePacked = false; // the built scene carries TriShade records (beyond LDS)
  uint32_t classMask = 0; // material classes that own at least one triangle (one k_shade launch per class)
  uint32_t classTextured = 0; // classes with at least one textured material in use (k_shade<class, TEXTURED>)
  uint32_t shadeClassMask = 0, shadeClassTextured = 0; // the same per SHADE class (gi_types.h: the HIT queues of the wavefront pipeline; classMask / classTextured pick the fused kernels)
  std::unique_ptr<SceneHost> host; // the scene as host arrays, kept for incremental transform updates
  std::vector<std::unique_ptr<SceneDevice>> replicas; // devices 1 .. N-1 (created with the first build when the library runs on several devices)
  // options + stats
  bool countTraversal = false, kernelTimers = false;
  uint32_t kernelTimerStride = 1;
  uint64_t optPoolSlots = 0, optSampleBufferMb = 0; // 0 = default
  int32_t optFusedPath = -1; // -1 = default: LDS-resident scenes run the fused persistent kernel k_path; 1 = k_path_bw (wave-local wavefront) when NEE is off; 2 = k_path; 0 = always the wavefront stage kernels
  int32_t optTraceDyn = -1; // -1 = default; 0 = block-synchronous k_trace everywhere; N = k_trace_dyn refill threshold
  // Visiting order of shadow walks (k_trace_dyn<any>; any order gives the same image): -1 = not chosen yet -- launches alternate between near-to-far (0) and slot order
  // (1) and the frame's node-visit counts are added up below; once both orders have walked enough rays the cheaper one is kept until the tree is rebuilt.
  std::atomic<int32_t> shadowOrder{-1}; /* read by every device worker at the start of its render, written by the primary at the end of its own */ uint64_t shadowOrderRays[2] = {0, 0}, shadowOrderSteps[2] = {0, 0};
  int32_t optDevices = 0;   // 0 = every device the library was initialised on; N = at most N of them
  uint32_t lastRenderDevices = 0; // devices the previous giCRender used: progressive accumulation blends against each device's own buffer, so a change restarts it
};

void SceneDevice::releaseAll()
{
  dNodes.release(); dTris.release(); dInstances.release(); dVerts.release(); dTriFaceId.release(); dTriShade.release();
  dTlasNodes.release(); dBlasNodes.release(); dTlasItems.release(); dFlatOfOrig.release(); dBlasTris.release(); dInstTrav.release();
  for (auto* b : dTexels) { b->release(); delete b; }
  dTexels.clear(); dTextures.release(); dMeshes.release(); dSceneData.release();
  dMaterials.release(); dSphere.release(); dDistant.release(); dRect.release(); dDisk.release();
  slots.release(); media.release(); scratchColor.release(); neeKey.release(); pathSegments.release(); sampleBuf.release(); accum.release();
  for (uint32_t q = 0; q < Q_COUNT; q++) { qSlot[q].release(); qA[q].release(); qB[q].release(); qC[q].release(); }
  qFresh[0].release(); qFresh[1].release();
  dCounters.release();
  if (hCounters) { (void)hipHostFree(hCounters); hCounters = nullptr; }
  if (evShade) { (void)hipEventDestroy(evShade); evShade = nullptr; }
  if (evShadow) { (void)hipEventDestroy(evShadow); evShadow = nullptr; }
  if (hPoll) { (void)hipHostFree(hPoll); hPoll = nullptr; for (hipEvent_t& e : pollEvent) { (void)hipEventDestroy(e); e = nullptr; } }
  for (hipEvent_t e : eventPool) (void)hipEventDestroy(e);
  eventPool.clear();
}

// ---------------------------------------------------------------------------------------------------------------
// init
// ---------------------------------------------------------------------------------------------------------------
extern "C" {

const char* giCGetLastError(void) { return t_lastError.c_str(); }

// The device list: giCInitializeDevices' argument, else $GATLING_DEVICES ("0,1,2,3" or "all"), else the one ordinal of giCInitialize.  The same ordinal may be
// listed twice (two contexts on one GPU): that is how the multi-device path is tested on a one-GPU box.
static int initDevices(const std::vector<int>& ordinals)
{
  if (g_ctx.initialized) return GI_C_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) { setError("no HIP device available: this library has no CPU fallback"); return GI_C_ERROR; }
  if (ordinals.empty()) { setError("empty device list"); return GI_C_ERROR; }
  for (int d : ordinals) if (d < 0 || d >= n) { setError("device ordinal out of range"); return GI_C_ERROR; }
  std::vector<DevCtx> devs;
  for (int d : ordinals) {
    DevCtx c; c.device = d;
    HIP_TRY(hipSetDevice(d));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, d));
    c.cuCount = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c.stream2, hipStreamNonBlocking));
    devs.push_back(c);
  }
  // the row shares travel to the primary device over xGMI: peer access both ways.  The outcome is kept (giCGetDevicePeerAccess) and decides how a device's share is
  // gathered: without peer access a hipMemcpyDefault between two devices silently stages through pageable host memory -- the library then does it itself, through a pinned buffer
  for (size_t i = 1; i < devs.size(); i++) {
    if (devs[i].device == devs[0].device) continue; // another context on the same GPU (tests): its memory is directly addressable
    int can01 = 0, can10 = 0;
    const hipError_t q0 = hipDeviceCanAccessPeer(&can01, devs[0].device, devs[i].device), q1 = hipDeviceCanAccessPeer(&can10, devs[i].device, devs[0].device);
    if (q0 != hipSuccess || q1 != hipSuccess) { devs[i].peer = -1; (void)hipGetLastError(); continue; }
    if (!can01 || !can10) { devs[i].peer = 0; continue; }
    (void)hipSetDevice(devs[0].device); const hipError_t e0 = hipDeviceEnablePeerAccess(devs[i].device, 0);
    (void)hipSetDevice(devs[i].device); const hipError_t e1 = hipDeviceEnablePeerAccess(devs[0].device, 0);
    auto fine = [](hipError_t e) { return e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled; };
    devs[i].peer = (fine(e0) && fine(e1)) ? 1 : -1;
    if (devs[i].peer != 1) fprintf(stderr, "[gatling_gi] peer access between devices %d and %d could not be enabled (%s / %s): row shares of device %d go through pinned host memory\n",
                                   devs[0].device, devs[i].device, hipGetErrorString(e0), hipGetErrorString(e1), devs[i].device);
    (void)hipGetLastError();
  }
  HIP_TRY(hipSetDevice(devs[0].device));
  g_ctx.devs = devs;
  g_ctx.device = devs[0].device; g_ctx.cuCount = devs[0].cuCount; g_ctx.stream = devs[0].stream;
  g_ctx.initialized = true;
  return GI_C_OK;
}

int giCInitialize(int deviceOrdinal)
{
  if (g_ctx.initialized) return GI_C_OK;
  std::vector<int> ordinals{deviceOrdinal};
  if (const char* e = getenv("GATLING_DEVICES")) { // "0,1,2,3", or "all"
    ordinals.clear();
    if (!strcmp(e, "all")) { int n = 0; if (hipGetDeviceCount(&n) == hipSuccess) for (int d = 0; d < n; d++) ordinals.push_back(d); }
    else for (const char* p = e; *p;) { char* end = nullptr; const long v = strtol(p, &end, 10); if (end == p) { p++; continue; } ordinals.push_back((int)v); p = end; }
    if (ordinals.empty()) ordinals.push_back(deviceOrdinal);
  }
  return initDevices(ordinals);
}

int giCInitializeDevices(const int32_t* deviceOrdinals, uint32_t count)
{
  if (!deviceOrdinals || count == 0) { setError("giCInitializeDevices: empty device list"); return GI_C_ERROR; }
  return initDevices(std::vector<int>(deviceOrdinals, deviceOrdinals + count));
}

uint32_t giCGetApiVersion(void) { return GI_C_API_VERSION; }
uint32_t giCGetDeviceCount(void) { return g_ctx.initialized ? (uint32_t)g_ctx.devs.size() : 0u; }
int32_t giCGetDevicePeerAccess(uint32_t index) { return (g_ctx.initialized && index < g_ctx.devs.size()) ? (int32_t)g_ctx.devs[index].peer : -1; }

void giCTerminate(void)
{
  if (!g_ctx.initialized) return;
  for (auto& w : g_ctx.workers) if (w) w->shutdown();
  g_ctx.workers.clear();
  for (DevCtx& c : g_ctx.devs) {
    (void)hipSetDevice(c.device);
    (void)hipStreamSynchronize(c.stream);
    (void)hipStreamDestroy(c.stream);
    if (c.stream2) { (void)hipStreamSynchronize(c.stream2); (void)hipStreamDestroy(c.stream2); }
  }
  g_ctx.devs.clear();
  g_ctx.stream = nullptr;
  g_ctx.initialized = false;
}

// ---------------------------------------------------------------------------------------------------------------
// scene / material / mesh
// ---------------------------------------------------------------------------------------------------------------
GiCScene* giCCreateScene(void)
{
  if (!g_ctx.initialized) { setError("giCCreateScene before giCInitialize"); return nullptr; }
  return new GiCScene();
}

void giCDestroyScene(GiCScene* s)
{
  if (!s) return;
  std::lock_guard<std::mutex> g(g_ctx.resourceMutex);
  for (auto& r : s->replicas) {
    (void)hipSetDevice(g_ctx.devs[r->slot].device);
    (void)hipStreamSynchronize(g_ctx.devs[r->slot].stream);
    r->releaseAll();
  }
  (void)hipSetDevice(g_ctx.device);
  (void)hipStreamSynchronize(g_ctx.stream);
  s->releaseAll();
  delete s;
}

GiCMaterial* giCCreateMaterial(GiCScene* scene, const char* name, const GiCMaterialDesc* desc)
{
  if (!scene || !desc) { setError("giCCreateMaterial: null argument"); return nullptr; }
  if (desc->klass > GI_C_MAT_OPEN_PBR) { setError("giCCreateMaterial: unsupported material class"); return nullptr; }
  GiCMaterial* m = new GiCMaterial{scene, name ? name : "", *desc};
  // subsurface_radius / subsurface_radius_scale joined the block in round 4 (slots 32..35, ignored before): a caller built against the older header leaves them 0, which
  // would mean an extinction of 1e6 per scene unit.  An all-zero radius AND scale reads as "unset": OpenPBR's defaults (open_pbr_surface.mtlx:47-49: 1; 1, 0.5, 0.25)
  if (desc->klass == GI_C_MAT_OPEN_PBR) {
    float* p = m->desc.p;
    if (p[GI_C_P_SUBSURFACE_RADIUS] == 0.0f && p[GI_C_P_SUBSURFACE_RADIUS_SCALE] == 0.0f && p[GI_C_P_SUBSURFACE_RADIUS_SCALE + 1] == 0.0f && p[GI_C_P_SUBSURFACE_RADIUS_SCALE + 2] == 0.0f) {
      p[GI_C_P_SUBSURFACE_RADIUS] = 1.0f; p[GI_C_P_SUBSURFACE_RADIUS_SCALE] = 1.0f; p[GI_C_P_SUBSURFACE_RADIUS_SCALE + 1] = 0.5f; p[GI_C_P_SUBSURFACE_RADIUS_SCALE + 2] = 0.25f;
    }
  }
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->materials.push_back(m);
  scene->dirty |= DIRTY_MATERIALS | DIRTY_BVH | DIRTY_FRAMEBUFFER;
  return m;
}

void giCDestroyMaterial(GiCMaterial* mat)
{
  if (!mat) return;
  GiCScene* s = mat->scene;
  {
    std::lock_guard<std::mutex> g(s->mutex);
    s->materials.erase(std::remove(s->materials.begin(), s->materials.end(), mat), s->materials.end());
    for (GiCMesh* m : s->meshes) if (m->material == mat) m->material = nullptr;
    s->dirty |= DIRTY_MATERIALS | DIRTY_BVH | DIRTY_FRAMEBUFFER;
  }
  delete mat;
}

// ---------------------------------------------------------------------------------------------------------------
// textures [ext]: decoded pixels in, device copies made with the next scene build (TextureManager.cpp:100-275 minus imgio)
// ---------------------------------------------------------------------------------------------------------------
static GiCTexture* createTextureImpl(GiCScene* scene, const GiCTextureDesc* desc);
GiCTexture* giCCreateTexture(GiCScene* scene, const GiCTextureDesc* desc)
{
  try { return createTextureImpl(scene, desc); }
  catch (const std::exception& e) { setError(std::string("giCCreateTexture: ") + e.what()); return nullptr; }
}
static GiCTexture* createTextureImpl(GiCScene* scene, const GiCTextureDesc* desc)
{
  if (!scene || !desc || !desc->rgba || desc->width == 0 || desc->height == 0) { setError("giCCreateTexture: bad arguments"); return nullptr; }
  std::unique_ptr<GiCTexture> t(new GiCTexture{scene, desc->width, desc->height, std::vector<float>(desc->rgba, desc->rgba + (size_t)desc->width * desc->height * 4)});
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->textures.push_back(t.get());
  scene->dirty |= DIRTY_MATERIALS | DIRTY_FRAMEBUFFER;
  return t.release();
}

// ---- image files: asset reader + loader hook in front of the in-library decoders --------------------------------------------------------------------------------
// (TextureManager.cpp:39-52: every image goes open -> size -> data -> ImgioLoadImage -> close through the registered GiAssetReader)
static std::mutex g_imageHookMutex;
static GiCAssetReader g_assetReader{};   // .open == nullptr: none registered
static GiCImageLoader g_imageLoader{};   // .load == nullptr: none registered
void giCRegisterAssetReader(const GiCAssetReader* r)
{
  std::lock_guard<std::mutex> g(g_imageHookMutex);
  if (r && r->open && r->size && r->data && r->close) g_assetReader = *r; else g_assetReader = GiCAssetReader{};
}
void giCSetImageLoader(const GiCImageLoader* l)
{
  std::lock_guard<std::mutex> g(g_imageHookMutex);
  if (l && l->load) g_imageLoader = *l; else g_imageLoader = GiCImageLoader{};
}
static float halfBitsToFloat(uint16_t h) // IEEE binary16 -> binary32, exact
{
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0u) {
    if (man == 0u) bits = sign;
    else { int e = -1; uint32_t m = man; do { e++; m <<= 1; } while (!(m & 0x400u)); bits = sign | (uint32_t)(127 - 15 - e) << 23 | (m & 0x3ffu) << 13; }
  } else if (exp == 31u) bits = sign | 0x7f800000u | man << 13;
  else bits = sign | (exp + 112u) << 23 | man << 13;
  float f; memcpy(&f, &bits, 4); return f;
}
// the image behind `path` as linear float RGBA in imgio's orientation
static bool loadImage(const char* path, bool srgbToLinear, bool keepHdr, uint32_t& w, uint32_t& h, std::vector<float>& px)
{
  GiCAssetReader reader; GiCImageLoader loader;
  { std::lock_guard<std::mutex> g(g_imageHookMutex); reader = g_assetReader; loader = g_imageLoader; }
  std::vector<uint8_t> fileBytes; const uint8_t* bytes = nullptr; size_t size = 0; void* asset = nullptr;
  if (reader.open) {
    asset = reader.open(reader.user, path);
    if (!asset) return false;
    size = (size_t)reader.size(reader.user, asset);
    bytes = static_cast<const uint8_t*>(reader.data(reader.user, asset));
  } else {
    if (!readFileBytes(path, fileBytes)) return false;
    bytes = fileBytes.data(); size = fileBytes.size();
  }
  bool ok = false;
  GiCDecodedImage img{};
  if (bytes && loader.load && loader.load(loader.user, path, bytes, (uint64_t)size, keepHdr ? 1 : 0, &img) == 1) {
    const size_t n = (size_t)img.width * img.height;
    // (a hook's answer is untrusted: a bogus width x height must not become a bad_alloc that leaves through the extern "C" callers, and release / close run whatever
    // happens -- ADVICE r05.  2^28 texels = 4 GiB of fp32 RGBA is the cap; the reference's largest texture is bounded by maxImageDimension2D, 16 384^2 = 2^28)
    if (img.pixels && n > 0 && n <= ((size_t)1 << 28) && img.format >= GI_C_IMAGE_RGBA8_UNORM && img.format <= GI_C_IMAGE_RGBA32_FLOAT) try {
      w = img.width; h = img.height; px.assign(n * 4, 1.0f);
      for (size_t i = 0; i < n; i++) {
        float* o = &px[i * 4];
        switch (img.format) {
          case GI_C_IMAGE_RGBA8_UNORM: { const uint8_t* p = static_cast<const uint8_t*>(img.pixels) + i * 4; for (int c = 0; c < 3; c++) o[c] = srgbToLinear ? srgb8ToLinear(p[c]) : (float)p[c] / 255.0f; o[3] = (float)p[3] / 255.0f; break; }
          case GI_C_IMAGE_RGB16_FLOAT: { const uint16_t* p = static_cast<const uint16_t*>(img.pixels) + i * 3; for (int c = 0; c < 3; c++) o[c] = halfBitsToFloat(p[c]); break; }
          case GI_C_IMAGE_RGBA16_FLOAT: { const uint16_t* p = static_cast<const uint16_t*>(img.pixels) + i * 4; for (int c = 0; c < 4; c++) o[c] = halfBitsToFloat(p[c]); break; }
          case GI_C_IMAGE_R32_FLOAT: { const float v = static_cast<const float*>(img.pixels)[i]; o[0] = o[1] = o[2] = v; break; }
          default: memcpy(o, static_cast<const float*>(img.pixels) + i * 4, 16); break;
        }
      }
      ok = true;
    } catch (const std::exception&) { ok = false; px.clear(); }
    if (loader.release) loader.release(loader.user, &img);
  }
  if (!ok && bytes) ok = decodeImageBytes(bytes, size, srgbToLinear, w, h, px);
  if (asset) reader.close(reader.user, asset);
  return ok;
}

// File textures are shared: a path that is already loaded (and still alive) yields the same texture with one more reference,
// as GiTextureManager's weak-pointer cache does (TextureManager.cpp:100-150); giCDestroyTexture drops one reference.
GiCTexture* giCCreateTextureFromFile(GiCScene* scene, const char* filePath, int32_t srgbToLinear)
{
  if (!scene || !filePath) { setError("giCCreateTextureFromFile: bad arguments"); return nullptr; }
  const std::string key = std::string(srgbToLinear ? "s:" : "l:") + filePath;
  {
    std::lock_guard<std::mutex> g(scene->mutex);
    for (GiCTexture* t : scene->textures) if (t->cacheKey == key) { t->refs++; return t; }
  }
  uint32_t w = 0, h = 0; std::vector<float> px;
  if (!loadImage(filePath, srgbToLinear != 0, /*keepHdr=*/false, w, h, px)) { setError("giCCreateTextureFromFile: cannot open or decode the image (in-library: .png, baseline .jpg, .hdr, .pfm; other formats through giCSetImageLoader)"); return nullptr; }
  GiCTextureDesc td{w, h, px.data()};
  GiCTexture* t = giCCreateTexture(scene, &td);
  if (t) { std::lock_guard<std::mutex> g(scene->mutex); t->cacheKey = key; }
  return t;
}

int giCDebugDecodeImage(const char* filePath, int32_t srgbToLinear, uint32_t* width, uint32_t* height, float* rgba, uint64_t rgbaFloats)
{
  uint32_t w = 0, h = 0; std::vector<float> px;
  if (!filePath || !loadImage(filePath, srgbToLinear != 0, /*keepHdr=*/false, w, h, px)) return 0;
  if (width) *width = w;
  if (height) *height = h;
  if (rgba && rgbaFloats >= px.size()) memcpy(rgba, px.data(), px.size() * sizeof(float));
  return 1;
}

void giCDestroyTexture(GiCTexture* tex)
{
  if (!tex) return;
  GiCScene* s = tex->scene;
  {
    std::lock_guard<std::mutex> g(s->mutex);
    if (--tex->refs != 0u) return; // shared file texture still in use
    s->textures.erase(std::remove(s->textures.begin(), s->textures.end(), tex), s->textures.end());
    for (GiCMaterial* m : s->materials) for (auto& b : m->tex) if (b.texture == tex) b.texture = nullptr;
    s->dirty |= DIRTY_MATERIALS | DIRTY_BVH | DIRTY_FRAMEBUFFER;
  }
  delete tex;
}

int giCSetMaterialTexture(GiCMaterial* mat, int32_t input, const GiCTextureBinding* binding)
{
  if (!mat || input < 0 || input >= GI_C_TEX_SLOT_COUNT) { setError("giCSetMaterialTexture: bad arguments"); return GI_C_ERROR; }
  if (binding && binding->texture && binding->texture->scene != mat->scene) { setError("giCSetMaterialTexture: texture belongs to another scene"); return GI_C_ERROR; }
  if (binding && (binding->wrapS < 0 || binding->wrapS > 3 || binding->wrapT < 0 || binding->wrapT > 3)) { setError("giCSetMaterialTexture: bad wrap mode"); return GI_C_ERROR; }
  std::lock_guard<std::mutex> g(mat->scene->mutex);
  if (binding) mat->tex[input] = *binding; else mat->tex[input] = GiCTextureBinding{};
  mat->scene->dirty |= DIRTY_MATERIALS | DIRTY_BVH | DIRTY_FRAMEBUFFER;
  return GI_C_OK;
}

int giCSetMaterialTextureTransform(GiCMaterial* mat, int32_t input, const float* xf)
{
  if (!mat || input < 0 || input >= GI_C_TEX_SLOT_COUNT) { setError("giCSetMaterialTextureTransform: bad arguments"); return GI_C_ERROR; }
  std::lock_guard<std::mutex> g(mat->scene->mutex);
  mat->hasTexXf[input] = xf != nullptr;
  if (xf) memcpy(mat->texXf[input], xf, sizeof(float) * 6);
  mat->scene->dirty |= DIRTY_MATERIALS | DIRTY_BVH | DIRTY_FRAMEBUFFER;
  return GI_C_OK;
}

static GiCMesh* createMeshImpl(GiCScene* scene, const GiCMeshDesc* d);
GiCMesh* giCCreateMesh(GiCScene* scene, const GiCMeshDesc* d)
{
  try { return createMeshImpl(scene, d); }
  catch (const std::exception& e) { setError(std::string("giCCreateMesh: ") + e.what()); return nullptr; }
}
static GiCMesh* createMeshImpl(GiCScene* scene, const GiCMeshDesc* d)
{
  if (!scene || !d) { setError("giCCreateMesh: null argument"); return nullptr; }
  if ((d->faceCount && !d->faces) || (d->vertexCount && !d->vertices)) { setError("giCCreateMesh: null arrays"); return nullptr; }
  for (uint32_t i = 0; i < d->faceCount; i++)
    for (int k = 0; k < 3; k++)
      if (d->faces[i].v_i[k] >= d->vertexCount) { setError("giCCreateMesh: face index out of range"); return nullptr; }
  std::unique_ptr<GiCMesh> m(new GiCMesh());
  m->scene = scene; m->name = d->name ? d->name : "";
  m->vertices.assign(d->vertices, d->vertices + d->vertexCount); // copies, like giProcessMeshData (Gi.cpp:628)
  m->faces.assign(d->faces, d->faces + d->faceCount);
  if (d->faceIds) m->faceIds.assign(d->faceIds, d->faceIds + d->faceCount);
  m->id = d->id; m->doubleSided = d->isDoubleSided != 0; m->flipFacing = d->isLeftHanded != 0; m->maxFaceId = d->maxFaceId;
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->meshes.push_back(m.get());
  scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
  return m.release();
}

void giCSetMeshTransform(GiCMesh* mesh, const float* mat4x4)
{
  if (!mesh || !mat4x4) return;
  std::lock_guard<std::mutex> g(mesh->scene->mutex); // buildScene reads the mesh under this lock (giCRender on another thread)
  memcpy(mesh->transform, mat4x4, sizeof(float) * 16);
  if (mesh->builtInstances != 0xffffffffu) { mesh->xformDirty = true; mesh->instDirty.clear(); mesh->scene->dirty |= DIRTY_XFORM | DIRTY_FRAMEBUFFER; } // same triangles elsewhere: incremental update (every instance of the mesh moves)
  else mesh->scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
}

void giCSetMeshInstanceTransforms(GiCMesh* mesh, uint32_t count, const float* transforms)
{
  if (!mesh || (count && !transforms)) return;
  std::vector<float> copy(transforms, transforms + (size_t)count * 16); // copy outside the lock, swap inside
  std::lock_guard<std::mutex> g(mesh->scene->mutex);
  mesh->instanceTransforms.swap(copy); // (`copy` now holds the previous transforms)
  if (mesh->builtInstances == count && copy.size() == (size_t)count * 16) { // same instance count: instances moved -- note which
    const bool all = mesh->xformDirty && mesh->instDirty.empty();
    if (!all) {
      if (mesh->instDirty.size() != count) mesh->instDirty.assign(count, 0);
      for (uint32_t i = 0; i < count; i++) if (memcmp(&copy[16 * (size_t)i], &mesh->instanceTransforms[16 * (size_t)i], 64) != 0) mesh->instDirty[i] = 1;
    }
    mesh->xformDirty = true; mesh->scene->dirty |= DIRTY_XFORM | DIRTY_FRAMEBUFFER;
  }
  else mesh->scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
}

void giCSetMeshInstanceIds(GiCMesh* mesh, uint32_t count, const int32_t* ids)
{
  if (!mesh || (count && !ids)) return;
  std::vector<int32_t> copy(ids, ids + count);
  std::lock_guard<std::mutex> g(mesh->scene->mutex);
  mesh->instanceIds.swap(copy);
  mesh->scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
}

void giCSetMeshMaterial(GiCMesh* mesh, GiCMaterial* mat)
{
  if (!mesh) return;
  std::lock_guard<std::mutex> g(mesh->scene->mutex);
  mesh->material = mat;
  mesh->scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
}

void giCSetMeshVisibility(GiCMesh* mesh, int32_t visible)
{
  if (!mesh) return;
  std::lock_guard<std::mutex> g(mesh->scene->mutex);
  mesh->visible = visible != 0;
  mesh->scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
}

void giCDestroyMesh(GiCMesh* mesh)
{
  if (!mesh) return;
  GiCScene* s = mesh->scene;
  {
    std::lock_guard<std::mutex> g(s->mutex);
    s->meshes.erase(std::remove(s->meshes.begin(), s->meshes.end(), mesh), s->meshes.end());
    s->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
  }
  delete mesh;
}

// ---------------------------------------------------------------------------------------------------------------
// lights (defaults and derived fields: Gi.cpp:2573-2976)
// ---------------------------------------------------------------------------------------------------------------
#define LIGHT_DIRTY(l) (l)->scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER

GiCSphereLight* giCCreateSphereLight(GiCScene* scene)
{
  if (!scene) return nullptr;
  std::lock_guard<std::mutex> g(scene->mutex);
  auto* l = new GiCSphereLight{scene, 0};
  SphereLightRec r{}; r.ds = packHalf2x16(1.0f, 1.0f); r.area = 1.0f; r.radius[0] = r.radius[1] = r.radius[2] = 0.5f;
  l->index = scene->sphereLights.add(l, r);
  scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  return l;
}
void giCDestroySphereLight(GiCScene* scene, GiCSphereLight* l)
{
  if (!scene || !l) return;
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->sphereLights.remove(l->index); scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  delete l;
}
void giCSetSphereLightPosition(GiCSphereLight* l, const float* p) { std::lock_guard<std::mutex> lk(l->scene->mutex); memcpy(l->scene->sphereLights.recs[l->index].pos, p, 12); LIGHT_DIRTY(l); }
void giCSetSphereLightBaseEmission(GiCSphereLight* l, const float* c) { std::lock_guard<std::mutex> lk(l->scene->mutex); memcpy(l->scene->sphereLights.recs[l->index].em, c, 12); LIGHT_DIRTY(l); }
void giCSetSphereLightRadius(GiCSphereLight* l, float rx, float ry, float rz)
{ std::lock_guard<std::mutex> lk(l->scene->mutex);
  // Knud Thomsen ellipsoid surface approximation (Gi.cpp:2635-2651)
  float ab = powf(rx * ry, 1.6f), ac = powf(rx * rz, 1.6f), bc = powf(ry * rz, 1.6f);
  float area = float(powf((ab + ac + bc) / 3.0f, 1.0f / 1.6f) * 4.0f * M_PI);
  SphereLightRec& r = l->scene->sphereLights.recs[l->index];
  r.radius[0] = rx; r.radius[1] = ry; r.radius[2] = rz; r.area = area;
  LIGHT_DIRTY(l);
}
void giCSetSphereLightDiffuseSpecular(GiCSphereLight* l, float d, float s) { std::lock_guard<std::mutex> lk(l->scene->mutex); l->scene->sphereLights.recs[l->index].ds = packHalf2x16(d, s); LIGHT_DIRTY(l); }

GiCDistantLight* giCCreateDistantLight(GiCScene* scene)
{
  if (!scene) return nullptr;
  std::lock_guard<std::mutex> g(scene->mutex);
  auto* l = new GiCDistantLight{scene, 0};
  DistantLightRec r{}; r.ds = packHalf2x16(1.0f, 1.0f); r.invPdf = 1.0f;
  l->index = scene->distantLights.add(l, r);
  scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  return l;
}
void giCDestroyDistantLight(GiCScene* scene, GiCDistantLight* l)
{
  if (!scene || !l) return;
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->distantLights.remove(l->index); scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  delete l;
}
void giCSetDistantLightDirection(GiCDistantLight* l, const float* d) { std::lock_guard<std::mutex> lk(l->scene->mutex); memcpy(l->scene->distantLights.recs[l->index].dir, d, 12); LIGHT_DIRTY(l); }
void giCSetDistantLightBaseEmission(GiCDistantLight* l, const float* c) { std::lock_guard<std::mutex> lk(l->scene->mutex); memcpy(l->scene->distantLights.recs[l->index].em, c, 12); LIGHT_DIRTY(l); }
void giCSetDistantLightAngle(GiCDistantLight* l, float angle)
{ std::lock_guard<std::mutex> lk(l->scene->mutex);
  float half = 0.5f * angle; // Gi.cpp:2723-2735
  DistantLightRec& r = l->scene->distantLights.recs[l->index];
  r.angle = angle; r.invPdf = (half > 0.0f) ? float(2.0f * M_PI * (1.0f - cosf(half))) : 1.0f;
  LIGHT_DIRTY(l);
}
void giCSetDistantLightDiffuseSpecular(GiCDistantLight* l, float d, float s) { std::lock_guard<std::mutex> lk(l->scene->mutex); l->scene->distantLights.recs[l->index].ds = packHalf2x16(d, s); LIGHT_DIRTY(l); }

GiCRectLight* giCCreateRectLight(GiCScene* scene)
{
  if (!scene) return nullptr;
  std::lock_guard<std::mutex> g(scene->mutex);
  auto* l = new GiCRectLight{scene, 0};
  const float t0[3] = {1, 0, 0}, t1[3] = {0, 1, 0};
  RectLightRec r{}; r.width = 1.0f; r.height = 1.0f; r.t0 = encodeDirection(t0); r.t1 = encodeDirection(t1); r.ds = packHalf2x16(1.0f, 1.0f);
  l->index = scene->rectLights.add(l, r);
  scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  return l;
}
void giCDestroyRectLight(GiCScene* scene, GiCRectLight* l)
{
  if (!scene || !l) return;
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->rectLights.remove(l->index); scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  delete l;
}
void giCSetRectLightOrigin(GiCRectLight* l, const float* o) { std::lock_guard<std::mutex> lk(l->scene->mutex); memcpy(l->scene->rectLights.recs[l->index].origin, o, 12); LIGHT_DIRTY(l); }
void giCSetRectLightTangents(GiCRectLight* l, const float* t0, const float* t1)
{ std::lock_guard<std::mutex> lk(l->scene->mutex);
  RectLightRec& r = l->scene->rectLights.recs[l->index]; r.t0 = encodeDirection(t0); r.t1 = encodeDirection(t1); LIGHT_DIRTY(l);
}
void giCSetRectLightBaseEmission(GiCRectLight* l, const float* c) { std::lock_guard<std::mutex> lk(l->scene->mutex); memcpy(l->scene->rectLights.recs[l->index].em, c, 12); LIGHT_DIRTY(l); }
void giCSetRectLightDimensions(GiCRectLight* l, float w, float h) { std::lock_guard<std::mutex> lk(l->scene->mutex); RectLightRec& r = l->scene->rectLights.recs[l->index]; r.width = w; r.height = h; LIGHT_DIRTY(l); }
void giCSetRectLightDiffuseSpecular(GiCRectLight* l, float d, float s) { std::lock_guard<std::mutex> lk(l->scene->mutex); l->scene->rectLights.recs[l->index].ds = packHalf2x16(d, s); LIGHT_DIRTY(l); }

GiCDiskLight* giCCreateDiskLight(GiCScene* scene)
{
  if (!scene) return nullptr;
  std::lock_guard<std::mutex> g(scene->mutex);
  auto* l = new GiCDiskLight{scene, 0};
  const float t0[3] = {1, 0, 0}, t1[3] = {0, 1, 0};
  DiskLightRec r{}; r.rx = 0.5f; r.ry = 0.5f; r.t0 = encodeDirection(t0); r.t1 = encodeDirection(t1); r.ds = packHalf2x16(1.0f, 1.0f);
  l->index = scene->diskLights.add(l, r);
  scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  return l;
}
void giCDestroyDiskLight(GiCScene* scene, GiCDiskLight* l)
{
  if (!scene || !l) return;
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->diskLights.remove(l->index); scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  delete l;
}
void giCSetDiskLightOrigin(GiCDiskLight* l, const float* o) { std::lock_guard<std::mutex> lk(l->scene->mutex); memcpy(l->scene->diskLights.recs[l->index].origin, o, 12); LIGHT_DIRTY(l); }
void giCSetDiskLightTangents(GiCDiskLight* l, const float* t0, const float* t1)
{ std::lock_guard<std::mutex> lk(l->scene->mutex);
  DiskLightRec& r = l->scene->diskLights.recs[l->index]; r.t0 = encodeDirection(t0); r.t1 = encodeDirection(t1); LIGHT_DIRTY(l);
}
void giCSetDiskLightBaseEmission(GiCDiskLight* l, const float* c) { std::lock_guard<std::mutex> lk(l->scene->mutex); memcpy(l->scene->diskLights.recs[l->index].em, c, 12); LIGHT_DIRTY(l); }
void giCSetDiskLightRadius(GiCDiskLight* l, float rx, float ry) { std::lock_guard<std::mutex> lk(l->scene->mutex); DiskLightRec& r = l->scene->diskLights.recs[l->index]; r.rx = rx; r.ry = ry; LIGHT_DIRTY(l); }
void giCSetDiskLightDiffuseSpecular(GiCDiskLight* l, float d, float s) { std::lock_guard<std::mutex> lk(l->scene->mutex); l->scene->diskLights.recs[l->index].ds = packHalf2x16(d, s); LIGHT_DIRTY(l); }


// ---------------------------------------------------------------------------------------------------------------
// scene data (primvars): Gi.h:76-92, 134, 213
// ---------------------------------------------------------------------------------------------------------------
static int setPrimvarsImpl(GiCMesh* mesh, std::vector<GiCPrimvar>& dst, uint32_t count, const GiCPrimvarData* pv);
static int setPrimvars(GiCMesh* mesh, std::vector<GiCPrimvar>& dst, uint32_t count, const GiCPrimvarData* pv)
{
  try { return setPrimvarsImpl(mesh, dst, count, pv); }
  catch (const std::exception& e) { setError(std::string("giCSetMesh*Primvars: ") + e.what()); return GI_C_ERROR; }
}
static int setPrimvarsImpl(GiCMesh* mesh, std::vector<GiCPrimvar>& dst, uint32_t count, const GiCPrimvarData* pv)
{
  if (!mesh || (count && !pv)) { setError("giCSetMesh*Primvars: bad arguments"); return GI_C_ERROR; }
  std::vector<GiCPrimvar> v;
  for (uint32_t i = 0; i < count; i++) {
    if (!pv[i].name || pv[i].type < 0 || pv[i].type > GI_C_PRIMVAR_INT4 || pv[i].interpolation < 0 || pv[i].interpolation > GI_C_INTERP_VERTEX) { setError("giCSetMesh*Primvars: bad primvar"); return GI_C_ERROR; }
    GiCPrimvar p{pv[i].name, pv[i].type, pv[i].interpolation, {}};
    // float and int32 elements are both 4 bytes: integer primvars keep their bit patterns in the float array (scene_data_lookup_int reads them back)
    if (pv[i].data) p.data.assign((const float*)pv[i].data, (const float*)pv[i].data + pv[i].dataSize / 4);
    v.push_back(std::move(p));
  }
  std::lock_guard<std::mutex> g(mesh->scene->mutex);
  dst = std::move(v);
  mesh->scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER; // Gi.cpp:685-700
  return GI_C_OK;
}
int giCSetMeshPrimvars(GiCMesh* mesh, uint32_t count, const GiCPrimvarData* pv) { return mesh ? setPrimvars(mesh, mesh->primvars, count, pv) : (setError("giCSetMeshPrimvars: null mesh"), GI_C_ERROR); }
int giCSetMeshInstancerPrimvars(GiCMesh* mesh, uint32_t count, const GiCPrimvarData* pv) { return mesh ? setPrimvars(mesh, mesh->instancerPrimvars, count, pv) : (setError("giCSetMeshInstancerPrimvars: null mesh"), GI_C_ERROR); }
int giCSetMaterialPrimvarInput(GiCMaterial* mat, int32_t input, const char* name)
{
  if (!mat || input < 0 || input >= GI_C_TEX_SLOT_COUNT || input == GI_C_TEX_NORMAL || input == GI_C_TEX_OPACITY || input == GI_C_TEX_COAT_NORMAL) { setError("giCSetMaterialPrimvarInput: bad arguments"); return GI_C_ERROR; }
  std::lock_guard<std::mutex> g(mat->scene->mutex);
  mat->primvarInput[input] = name ? name : "";
  mat->scene->dirty |= DIRTY_MATERIALS | DIRTY_BVH | DIRTY_FRAMEBUFFER;
  return GI_C_OK;
}

GiCDomeLight* giCCreateDomeLight(GiCScene* scene, const char* filePath)
{
  if (!scene) return nullptr;
  auto* l = new GiCDomeLight(); l->scene = scene; l->filePath = filePath ? filePath : "";
  // the reference decodes the file through imgio (Gi.cpp:2215-2230); here: Radiance RGBE, PFM and PNG, anything else stays unloaded
  uint32_t w = 0, h = 0; std::vector<float> px;
  if (!l->filePath.empty() && loadImage(l->filePath.c_str(), /*srgbToLinear=*/false, /*keepHdr=*/true, w, h, px)) {
    GiCTextureDesc td{w, h, px.data()};
    l->texture = giCCreateTexture(scene, &td);
    l->ownsTexture = l->texture != nullptr;
  } else if (!l->filePath.empty()) {
    fprintf(stderr, "[gatling_gi] unable to load dome light texture at '%s' (.hdr, .pfm, .png and baseline .jpg are decoded in-library, other formats through giCSetImageLoader)\n", l->filePath.c_str());
  }
  return l;
}
void giCDestroyDomeLight(GiCDomeLight* l)
{
  if (!l) return;
  if (l->ownsTexture) giCDestroyTexture(l->texture);
  std::lock_guard<std::mutex> g(l->scene->mutex);
  l->scene->dirty |= DIRTY_FRAMEBUFFER;
  delete l;
}
void giCSetDomeLightTexture(GiCDomeLight* l, GiCTexture* t)
{
  if (!l) return;
  GiCTexture* old = nullptr;
  {
    std::lock_guard<std::mutex> lk(l->scene->mutex);
    if (l->ownsTexture) { old = l->texture; l->ownsTexture = false; }
    l->texture = t; l->scene->dirty |= DIRTY_FRAMEBUFFER;
  }
  if (old) giCDestroyTexture(old); // takes the scene mutex itself
}
void giCSetDomeLightRotation(GiCDomeLight* l, const float* q) { std::lock_guard<std::mutex> lk(l->scene->mutex); memcpy(l->rotation, q, 16); l->scene->dirty |= DIRTY_FRAMEBUFFER; }
void giCSetDomeLightBaseEmission(GiCDomeLight* l, const float* c) { std::lock_guard<std::mutex> lk(l->scene->mutex); memcpy(l->baseEmission, c, 12); l->scene->dirty |= DIRTY_FRAMEBUFFER; }
void giCSetDomeLightDiffuseSpecular(GiCDomeLight* l, float d, float s) { std::lock_guard<std::mutex> lk(l->scene->mutex); l->diffuse = d; l->specular = s; l->scene->dirty |= DIRTY_FRAMEBUFFER; }

// ---------------------------------------------------------------------------------------------------------------
// render buffers (Gi.cpp:2978-3006; memory is created lazily by giRender, :1997-2034 -- here at creation so that
// giCGetRenderBufferMem is valid immediately)
// ---------------------------------------------------------------------------------------------------------------
GiCRenderBuffer* giCCreateRenderBuffer(uint32_t width, uint32_t height, int32_t format)
{
  if (!g_ctx.initialized) { setError("giCCreateRenderBuffer before giCInitialize"); return nullptr; }
  uint32_t stride = (format == GI_C_FORMAT_FLOAT32_VEC4) ? 16u : 4u;
  auto* rb = new GiCRenderBuffer{width, height, stride, (size_t)width * height * stride};
  size_t bytes = rb->size ? rb->size : 16;
  if (hipMalloc(&rb->deviceMem, bytes) != hipSuccess || hipHostMalloc(&rb->hostMem, bytes, hipHostMallocDefault) != hipSuccess) {
    setError("failed to allocate render buffer");
    if (rb->deviceMem) (void)hipFree(rb->deviceMem);
    delete rb; return nullptr;
  }
  (void)hipMemset(rb->deviceMem, 0, bytes);
  memset(rb->hostMem, 0, bytes);
  return rb;
}
void giCDestroyRenderBuffer(GiCRenderBuffer* rb)
{
  if (!rb) return;
  std::lock_guard<std::mutex> g(g_ctx.resourceMutex);
  (void)hipStreamSynchronize(g_ctx.stream);
  if (rb->deviceMem) (void)hipFree(rb->deviceMem);
  if (rb->hostMem) (void)hipHostFree(rb->hostMem);
  if (rb->stageMem) (void)hipHostFree(rb->stageMem);
  for (size_t i = 0; i < rb->replicaMem.size(); i++)
    if (rb->replicaMem[i] && i + 1 < g_ctx.devs.size()) { (void)hipSetDevice(g_ctx.devs[i + 1].device); (void)hipFree(rb->replicaMem[i]); }
  (void)hipSetDevice(g_ctx.device);
  delete rb;
}
void* giCGetRenderBufferMem(GiCRenderBuffer* rb) { return rb ? rb->hostMem : nullptr; }
void* giCGetRenderBufferDeviceMem(GiCRenderBuffer* rb) { return rb ? rb->deviceMem : nullptr; }
void giCSetRenderBufferDeviceOnly(GiCRenderBuffer* rb, int32_t deviceOnly) { if (rb) rb->deviceOnly = deviceOnly != 0; }

int giCSetSceneOption(GiCScene* scene, int32_t option, int32_t value)
{
  if (!scene) return GI_C_ERROR;
  std::lock_guard<std::mutex> g(scene->mutex);
  if (option == GI_C_SCENE_OPTION_COUNT_TRAVERSAL) { scene->countTraversal = value != 0; return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_KERNEL_TIMERS) { scene->kernelTimers = value != 0; scene->kernelTimerStride = value > 0 ? (uint32_t)value : 1u; return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_POOL_SLOTS) { scene->optPoolSlots = value > 0 ? (uint64_t)value : 0; return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_TWO_LEVEL) { scene->optTwoLevel = value < 0 ? -1 : (value ? 1 : 0); scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER; return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_TRACE_DYNAMIC) { scene->optTraceDyn = value < 0 ? -1 : (value > 64 ? 64 : value); return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_FUSED_PATH) { scene->optFusedPath = value < 0 ? -1 : (value > 2 ? -1 : value); return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_SAMPLE_BUFFER_MB) { scene->optSampleBufferMb = value > 0 ? (uint64_t)value : 0; return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_DEVICES) { scene->optDevices = value > 0 ? value : 0; scene->dirty |= DIRTY_BVH | DIRTY_LIGHTS | DIRTY_FRAMEBUFFER; /* replicas are made with the build; a NEW replica also needs the lights, which travel under DIRTY_LIGHTS only */ return GI_C_OK; }
  setError("unknown scene option"); return GI_C_ERROR;
}

int giCGetRenderStats(const GiCScene* scene, GiCRenderStats* out)
{
  if (!scene || !out) return GI_C_ERROR;
  *out = scene->stats;
  return GI_C_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// scene build: flatten instances into world space, pack vertex data, build + upload the BVH8
// ---------------------------------------------------------------------------------------------------------------
namespace {

// world = local * M_prim * M_instance with USD row vectors (Gi.cpp:641-658, 1191); returns rows of the 3x4
// column-vector affine.  Same operation order as glm's mat4 * mat4.
void composeTransform(const float* prim, const float* inst, float out[12])
{
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

inline void xformPoint(const float a[12], const float p[3], float out[3])
{
  out[0] = ((a[0] * p[0] + a[1] * p[1]) + a[2] * p[2]) + a[3];
  out[1] = ((a[4] * p[0] + a[5] * p[1]) + a[6] * p[2]) + a[7];
  out[2] = ((a[8] * p[0] + a[9] * p[1]) + a[10] * p[2]) + a[11];
}

// Shade class of a material (gi_types.h MAT_CLASS_COUNT): its BSDF class, or -- the reference's per-material feature #defines done the wavefront way,
// GlslShaderGen.cpp:204-274, Gi.cpp:1545-1562 -- the specialised variant its hits are binned and shaded by.  OpenPBR BASE: every optional lobe absent (no coat, fuzz,
// thin film, anisotropy, transmission, subsurface, not thin-walled), no bound texture / primvar input, every parameter finite (the variant drops products with exact
// zeros, which a NaN or an infinity would not honour).  GATLING_OPTIONS=shade_variants=0 keeps every material in its full kernel (tests: same bits).
uint32_t shadeClassOf(const MaterialRec& m)
{
  if (m.klass != GI_C_MAT_OPEN_PBR || optionValue("shade_variants", 1) == 0) return m.klass & 0xfu;
  if (m.flags & MAT_FLAG_TEXTURED) return m.klass;
  for (uint32_t i = 0; i < MAT_PARAM_COUNT; i++) if (!std::isfinite(m.p[i])) return m.klass;
  if ((uint32_t)m.p[MP_FEATURES] != 0u) return m.klass;
  if (m.p[MP_COAT] != 0.0f || m.p[GI_C_P_CLEARCOAT] != 0.0f || m.p[GI_C_P_TRANSMISSION_WEIGHT] != 0.0f) return m.klass;
  return SHADE_CLASS_OPBR_BASE;
}

// Hostile geometry (bvh8.h "Inactive items").  A coordinate the build works with: finite, at most 1e18 in magnitude.
inline bool usableCoordinate(float x) { return std::fabs(x) <= 1.0e18f; } // (false for NaN)
// An instance the flattening can use: every entry of its affine finite and its 3x3 invertible with an inverse that is finite in fp32 (w2o transforms normals and,
// in the two-level layout, rays).  Every triangle of an instance that is not -- a NaN or singular giCSetMeshTransform / instance transform -- is inactive.
inline bool usableInstance(const InstanceRec& ir)
{
  for (int i = 0; i < 12; i++) if (!std::isfinite(ir.o2w[i])) return false;
  for (int i = 0; i < 9; i++) if (!std::isfinite(ir.w2o[i])) return false;
  return true;
}
// Shading attributes of a vertex as the scene build takes them: a normal or tangent with a non-finite component becomes +Z, a non-finite texture coordinate 0, a
// non-finite bitangent sign +1 (the position is left alone: it decides whether the triangle is active).  The reference uploads what it is given (Gi.cpp:848-861) and
// a NaN attribute is a NaN pixel there; here hostile attributes cost the shading of the faces that use them, nothing else.
inline GiCVertex usableShadingAttributes(const GiCVertex& in)
{
  GiCVertex v = in;
  auto direction = [](float* d) { if (!std::isfinite(d[0]) || !std::isfinite(d[1]) || !std::isfinite(d[2])) { d[0] = 0.0f; d[1] = 0.0f; d[2] = 1.0f; } };
  direction(v.norm); direction(v.tangent);
  if (!std::isfinite(v.u)) v.u = 0.0f;
  if (!std::isfinite(v.v)) v.v = 0.0f;
  if (!std::isfinite(v.bitangentSign)) v.bitangentSign = 1.0f;
  return v;
}
// one flattened triangle (Gi.cpp:1188-1202 hands the instance transform to the TLAS; here it is applied); `usable` false: marked inactive for the builder
inline void flattenTriangle(const InstanceRec& ir, bool usable, const GiCMesh* m, uint32_t f, TriRec& t)
{
  float p0[3], p1[3], p2[3];
  xformPoint(ir.o2w, m->vertices[m->faces[f].v_i[0]].pos, p0);
  xformPoint(ir.o2w, m->vertices[m->faces[f].v_i[1]].pos, p1);
  xformPoint(ir.o2w, m->vertices[m->faces[f].v_i[2]].pos, p2);
  for (int a = 0; a < 3; a++) { t.v0[a] = p0[a]; t.e1[a] = p1[a] - p0[a]; t.e2[a] = p2[a] - p0[a]; }
  if (!usable) t.v0[0] = std::numeric_limits<float>::quiet_NaN();
}

// Two-level layout (SceneView::tlasNodes ...): built next to the flat BVH for instanced scenes that do not fit LDS.  The flat
// arrays stay (k_shade reads the hit's TriRec, k_aov / giCTraceRays traverse them); the two-level ones are what k_trace_dyn2 walks,
// and they are small: one BLAS per MESH instead of one subtree per instance, so traversal stays in the caches.
template <class MB>
int buildTwoLevel(GiCScene* s, const std::vector<MB>& meshBuilds, const std::vector<InstanceRec>& instances, size_t flatTris, size_t flatNodes, TwoLevelHost& out)
{
  s->twoLevel = false;
  int want = s->optTwoLevel;
  want = (int)optionValue("two_level", want);
  size_t uniqueTris = 0;
  for (const MB& mb : meshBuilds) uniqueTris += mb.instCount ? mb.m->faces.size() : 0;
  const bool beyondLds = flatNodes > 384u || flatTris > 128u;
  (void)uniqueTris;
  // Opt-in only.  Measured (r01k): although its working set is tiny (C4: 1.5 MB of BLAS nodes + 2.6 MB of mesh triangles instead of
  // 41 + 335 MB) the first version is SLOWER than the flat layout -- C4 trace 185 -> 207 ms, C5 834 -> 1945 ms (29 instead of 20 nodes
  // per ray: overlapping instance boxes, each visit pays a ray transform, a BLAS root and a restore; candidates cost a rebuild).
  // ... except where the flat traversal cannot address the scene: its wave-cooperative triangle ring packs (lane, flat triangle) into 32 bits, 2^26 triangles; the
  // two-level walk queues MESH triangles there (one BLAS per mesh), so heavily instanced scenes beyond that bound take it automatically (r04; the hit record's
  // triangle word, flat index | class << 28, then bounds the scene at 2^28 flattened triangles)
  if (flatTris >= ((size_t)1 << 26) && want < 0) want = 1;
  if (flatNodes * sizeof(Node8) >= ((size_t)1 << 32) && want < 0) want = 1; // (the flat walk addresses nodes by 32-bit byte offset, gi_traversal.h node_load: 53 M nodes -- beyond any 2^26-triangle tree)
  if (want <= 0 || instances.empty() || !beyondLds) return GI_C_OK;
  std::vector<Node8> blasNodes; std::vector<BlasTri> blasTris; std::vector<InstTrav> instTrav(instances.size());
  uint32_t blasDepth = 0;
  std::vector<float> instBoxes(instances.size() * 6);
  auto padBox = [](float* lo, float* hi) { // as bvh8.cpp pads triangle boxes: 2^-20 relative, covers the rounding of the exact test's inputs
    for (int a = 0; a < 3; a++) { const float mag = std::max(std::fabs(lo[a]), std::fabs(hi[a])) + (hi[a] - lo[a]); const float pad = mag * 9.5367431640625e-7f + 1.0e-30f; lo[a] -= pad; hi[a] += pad; }
  };
  for (const MB& mb : meshBuilds) {
    if (mb.instCount == 0) continue;
    const GiCMesh* m = mb.m;
    const size_t nf = m->faces.size();
    std::vector<float> boxes(nf * 6);
    float mlo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (size_t f = 0; f < nf; f++) {
      float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
      bool faceOk = true; // (an unusable face keeps an inverted box: the builder leaves it out, and it must not widen the mesh magnitude below)
      for (int k = 0; k < 3; k++) { const float* p = m->vertices[m->faces[f].v_i[k]].pos; for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); faceOk = faceOk && usableCoordinate(p[a]); } }
      if (faceOk) padBox(lo, hi); else for (int a = 0; a < 3; a++) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; }
      for (int a = 0; a < 3; a++) { boxes[6 * f + a] = lo[a]; boxes[6 * f + 3 + a] = hi[a]; }
      if (faceOk) for (int a = 0; a < 3; a++) { mlo[a] = std::min(mlo[a], lo[a]); mhi[a] = std::max(mhi[a], hi[a]); }
    }
    Bvh8 b; std::vector<uint32_t> order;
    buildBvh8Boxes(boxes.data(), nf, b, order);
    const uint32_t nodeBase = (uint32_t)blasNodes.size(), triBase = (uint32_t)blasTris.size();
    for (Node8 n : b.nodes) { n.childBase += nodeBase; n.triBase += triBase; blasNodes.push_back(n); }
    for (uint32_t f : order) {
      BlasTri bt{};
      memcpy(bt.p0, m->vertices[m->faces[f].v_i[0]].pos, 12); memcpy(bt.p1, m->vertices[m->faces[f].v_i[1]].pos, 12); memcpy(bt.p2, m->vertices[m->faces[f].v_i[2]].pos, 12);
      bt.prim = f;
      blasTris.push_back(bt);
    }
    blasDepth = std::max(blasDepth, b.maxDepth);
    // object-space magnitude the transformed ray's rounding error scales with inside this mesh (see wave_step2)
    const float extent = (std::fabs(mlo[0]) + std::fabs(mlo[1]) + std::fabs(mlo[2])) + (std::fabs(mhi[0]) + std::fabs(mhi[1]) + std::fabs(mhi[2]));
    for (uint32_t ii = 0; ii < mb.instCount; ii++) {
      const uint32_t inst = mb.instFirst + ii;
      InstTrav& tv = instTrav[inst];
      tv = InstTrav{};
      memcpy(tv.o2w, instances[inst].o2w, sizeof(tv.o2w)); memcpy(tv.w2o, instances[inst].w2o, sizeof(tv.w2o));
      tv.blasRoot = nodeBase; tv.triBase = mb.triFirst + ii * (uint32_t)nf; tv.matFlags = mb.matFlags; tv.slack = extent;
      float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
      // inactive triangles (bvh8.h): a face with an unusable OBJECT-space vertex is left out of the BLAS by the builder and out of this box; an unusable instance
      // keeps the inverted box (the builder leaves it out of the TLAS).  A usable face whose WORLD-space vertex is unusable is inactive in the flat tree but would
      // be walked here: such scenes keep the flat layout
      if (usableInstance(instances[inst]))
        for (size_t f = 0; f < nf; f++) {
          bool objectOk = true, worldOk = true; float q[3][3];
          for (int k = 0; k < 3; k++) {
            const float* o = m->vertices[m->faces[f].v_i[k]].pos;
            xformPoint(instances[inst].o2w, o, q[k]);
            for (int a = 0; a < 3; a++) { objectOk = objectOk && usableCoordinate(o[a]); worldOk = worldOk && usableCoordinate(q[k][a]); }
          }
          if (!objectOk) continue;
          if (!worldOk) { if (want > 0) fprintf(stderr, "[gatling_gi] two-level layout not used: an instance carries triangles that leave the usable coordinate range in world space\n"); return GI_C_OK; }
          for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], q[k][a]); hi[a] = std::max(hi[a], q[k][a]); }
        }
      padBox(lo, hi);
      for (int a = 0; a < 3; a++) { instBoxes[6 * inst + a] = lo[a]; instBoxes[6 * inst + 3 + a] = hi[a]; }
    }
  }
  Bvh8 tlas; std::vector<uint32_t> tlasItems;
  buildBvh8Boxes(instBoxes.data(), instances.size(), tlas, tlasItems);
  // per-lane stack: a TLAS level can leave a node group and an instance group behind, a BLAS level a node group
  if (blasTris.size() >= ((size_t)1 << 26)) { if (want > 0) fprintf(stderr, "[gatling_gi] two-level layout not used: 2^26 or more unique mesh triangles\n"); return GI_C_OK; }
  if (2u * tlas.maxDepth + blasDepth + 1u > 16u) { if (want > 0) fprintf(stderr, "[gatling_gi] two-level layout not used: trees too deep for the 16-entry stack\n"); return GI_C_OK; }
  s->twoLevel = true;
  if (getenv("GATLING_BUILD_TIMING")) fprintf(stderr, "[gatling_gi] two-level: TLAS %zu nodes over %zu instances, %zu BLAS nodes, %zu mesh triangles (flat: %zu nodes, %zu triangles)\n",
                                              tlas.nodes.size(), instances.size(), blasNodes.size(), blasTris.size(), flatNodes, flatTris);
  out.tlasNodes.swap(tlas.nodes); out.tlasItems.swap(tlasItems); out.blasNodes.swap(blasNodes); out.blasTris.swap(blasTris); out.instTrav.swap(instTrav);
  return GI_C_OK;
}

// ... and its upload into one device's memory (the primary's and every replica's: multi-device renders replicate the scene)
int uploadSceneTo(GiCScene* s, SceneDevice& D, const SceneHost& H)
{
  const DevCtx& ctx = g_ctx.devs[D.slot];
  HIP_TRY(hipSetDevice(ctx.device));
  hipStream_t st = ctx.stream;
  if (s->twoLevel) {
    if (D.dTlasNodes.upload(H.two.tlasNodes, st) || D.dTlasItems.upload(H.two.tlasItems, st) || D.dBlasNodes.upload(H.two.blasNodes, st) || D.dBlasTris.upload(H.two.blasTris, st) ||
        D.dInstTrav.upload(H.two.instTrav, st) || D.dFlatOfOrig.upload(H.flatOfOrig, st))
      return GI_C_ERROR;
  }
  if (D.dTriFaceId.upload(H.triFaceId, st) || D.dTriShade.upload(H.triShade, st)) return GI_C_ERROR;
  if (D.dMeshes.upload(H.meshRecs, st) || D.dSceneData.upload(H.sceneData, st)) return GI_C_ERROR;
  { // textures: one device array per image + the TextureRec table
    for (auto* b : D.dTexels) { b->release(); delete b; }
    D.dTexels.clear();
    std::vector<TextureRec> recs(s->textures.size());
    for (size_t i = 0; i < s->textures.size(); i++) {
      auto* b = new DeviceBuffer<float>();
      D.dTexels.push_back(b);
      if (b->upload(s->textures[i]->rgba, st)) return GI_C_ERROR;
      recs[i] = TextureRec{b->ptr, s->textures[i]->width, s->textures[i]->height};
    }
    if (D.dTextures.upload(recs, st)) return GI_C_ERROR;
    HIP_TRY(hipStreamSynchronize(st)); // `recs` goes out of scope
  }
  if (D.dNodes.upload(H.bvh.nodes, st) || D.dTris.upload(H.bvh.tris, st) || D.dInstances.upload(H.instances, st) ||
      D.dVerts.upload(H.verts, st) || D.dMaterials.upload(H.mats, st))
    return GI_C_ERROR;
  HIP_TRY(hipStreamSynchronize(st)); // host vectors may go out of scope
  return GI_C_OK;
}

// devices a render of this scene may use (replicas exist for slots 1 .. n-1 after buildScene)
uint32_t sceneDeviceCount(const GiCScene* s)
{
  uint32_t n = (uint32_t)g_ctx.devs.size();
  if (s->optDevices > 0) n = std::min<uint32_t>(n, (uint32_t)s->optDevices);
  return std::max(n, 1u);
}
SceneDevice& sceneDevice(GiCScene* s, uint32_t slot) { return slot == 0u ? static_cast<SceneDevice&>(*s) : *s->replicas[slot - 1u]; }

void nodeBounds(const Node8& n, float box[6]);
// The flat tree's root bounds for FLAG_BOUNDS_RETIRE: the dequantised child boxes of node 0 (which contain every triangle's padded box), padded once more by 1e-5 of
// their magnitude and extent -- k_raygen's slab test adds its own per-ray rounding allowance on top.
static void setSceneBounds(GiCScene* s, const std::vector<Node8>& nodes)
{
  s->boundsValid = false;
  if (nodes.empty()) return;
  float b[6]; nodeBounds(nodes[0], b);
  for (int a = 0; a < 3; a++) {
    if (!(b[a] <= b[3 + a]) || !std::isfinite(b[a]) || !std::isfinite(b[3 + a])) return; // empty root (no triangles) or overflowing planes: no early retire
    const float pad = (std::fabs(b[a]) + std::fabs(b[3 + a]) + (b[3 + a] - b[a])) * 1.0e-5f + 1.0e-30f;
    s->bounds[a] = b[a] - pad; s->bounds[3 + a] = b[3 + a] + pad;
  }
  s->boundsValid = true;
}

int buildScene(GiCScene* s)
{
  double t0 = nowMs();
  std::unique_ptr<SceneHost> hostPtr(new SceneHost());
  SceneHost& H = *hostPtr;
  s->host.reset(); // (a failed build leaves no stale host copy behind)
  for (GiCMesh* m : s->meshes) { m->builtInstances = 0xffffffffu; m->xformDirty = false; m->instDirty.clear(); }
  std::vector<FVertex>& verts = H.verts; std::vector<InstanceRec>& instances = H.instances; std::vector<TriRec> tris; std::vector<int32_t> faceIdOf;
  std::vector<MaterialRec>& mats = H.mats; mats.resize(s->materials.size());
  for (size_t i = 0; i < s->materials.size(); i++) {
    mats[i].klass = s->materials[i]->desc.klass; mats[i].flags = s->materials[i]->desc.flags & ~(MAT_FLAG_TEXTURED | MAT_FLAG_OPACITY_TEX);
    for (uint32_t slot = 0; slot < TEX_SLOT_COUNT; slot++) {
      const GiCTextureBinding& b = s->materials[i]->tex[slot];
      TexBindingRec& r = mats[i].tex[slot];
      r = TexBindingRec{};
      auto tit = b.texture ? std::find(s->textures.begin(), s->textures.end(), b.texture) : s->textures.end();
      if (tit == s->textures.end()) {
        if (!s->materials[i]->primvarInput[slot].empty()) {
          r.mode = TEX_MODE_PRIMVAR; mats[i].flags |= MAT_FLAG_TEXTURED;
          if (s->materials[i]->primvarInput[slot] == "CAMERA_POSITION") r.mode |= TEX_MODE_CAMERA_POSITION; // Frontend.cpp:251-252: named scene data answered from the UBO
          if (s->materials[i]->primvarInput[slot] == "FRAME") r.mode |= TEX_MODE_FRAME;
        }
        continue;
      }
      r.tex = (uint32_t)(tit - s->textures.begin()) + 1u;
      r.mode = (uint32_t)b.wrapS | ((uint32_t)b.wrapT << 8) | (((uint32_t)b.channel & 3u) << 16);
      memcpy(r.scale, b.scale, 16); memcpy(r.bias, b.bias, 16);
      if (s->materials[i]->hasTexXf[slot]) { r.mode |= TEX_MODE_XFORM; memcpy(r.xf, s->materials[i]->texXf[slot], sizeof(r.xf)); }
      mats[i].flags |= slot == TEX_OPACITY ? MAT_FLAG_OPACITY_TEX : MAT_FLAG_TEXTURED; // opacity is looked up by the any-hit test, not by k_shade
    }
    memcpy(mats[i].p, s->materials[i]->desc.p, sizeof(float) * MAT_PARAM_COUNT);
    deriveMaterialConstants(mats[i]);
  }
  uint32_t meshIdx = 0;
  std::vector<MeshBuild>& meshBuilds = H.meshBuilds; // visible meshes in scene order (two-level layout, incremental updates)
  std::vector<MeshRec>& meshRecs = H.meshRecs; std::vector<float>& sceneData = H.sceneData;
  s->classMask = 0; s->hasCutouts = false; s->classTextured = 0; s->shadeClassMask = 0; s->shadeClassTextured = 0;
  for (GiCMesh* m : s->meshes) {
    if (!m->visible) continue; // Gi.cpp:801-804
    if (m->faces.empty()) continue;
    auto mit = std::find(s->materials.begin(), s->materials.end(), m->material);
    if (mit == s->materials.end()) { fprintf(stderr, "[gatling_gi] invalid BLAS material for mesh %s\n", m->name.c_str()); continue; } // Gi.cpp:818-822
    const uint32_t material = (uint32_t)(mit - s->materials.begin());
    if (material > 0x00ffffffu) { setError("too many materials"); return GI_C_ERROR; }
    const bool cutoutMat = mats[material].p[MP_CUTOUT] < 1.0f || (mats[material].flags & MAT_FLAG_OPACITY_TEX) != 0u;
    if (cutoutMat) s->hasCutouts = true;
    const uint32_t shadeClass = shadeClassOf(mats[material]);
    const uint32_t matFlags = material | (shadeClass << 24) | (cutoutMat ? (1u << 28) : 0u) | (((m->flipFacing ? 1u : 0u) | (m->doubleSided ? 2u : 0u)) << 30);
    s->classMask |= 1u << (mats[material].klass & 0xfu); s->shadeClassMask |= 1u << shadeClass;
    if (mats[material].flags & MAT_FLAG_TEXTURED) { s->classTextured |= 1u << (mats[material].klass & 0xfu); s->shadeClassTextured |= 1u << shadeClass; }
    const uint32_t vertexOffset = (uint32_t)verts.size();
    { // scene data the mesh's material reads (Gi.cpp:905-1019): instancer primvars first, mesh primvars override, by name
      MeshRec mr{}; mr.vertexOffset = vertexOffset;
      for (uint32_t slot = 0; slot < TEX_SLOT_COUNT; slot++) {
        const std::string& want = (*mit)->primvarInput[slot];
        if (want.empty()) continue;
        const GiCPrimvar* pv = nullptr;
        for (const GiCPrimvar& p : m->instancerPrimvars) if (p.name == want && !p.data.empty()) { pv = &p; break; }
        for (const GiCPrimvar& p : m->primvars) if (p.name == want && !p.data.empty()) { pv = &p; break; }
        if (!pv) continue; // SCENE_DATA_INVALID
        const bool isInt = pv->type > GI_C_PRIMVAR_VEC4; // Int .. Int4 (Gi.h:76-79)
        const uint32_t stride = (uint32_t)(isInt ? pv->type - GI_C_PRIMVAR_INT : pv->type) + 1u;
        size_t entries = 1; // what a lookup can index: zero-padded so that short arrays read 0 like the oracle
        if (pv->interpolation == GI_C_INTERP_VERTEX) entries = m->vertices.size();
        else if (pv->interpolation == GI_C_INTERP_UNIFORM) entries = m->faces.size();
        else if (pv->interpolation == GI_C_INTERP_INSTANCE) { int32_t mx = (int32_t)(m->instanceTransforms.size() / 16) - 1; for (int32_t id : m->instanceIds) mx = std::max(mx, id); entries = (size_t)std::max(mx, 0) + 1; }
        mr.sdOffset[slot] = (uint32_t)sceneData.size();
        mr.sdInfo[slot] = 1u | ((stride - 1u) << 1) | ((uint32_t)pv->interpolation << 3) | (isInt ? SD_INFO_INT : 0u);
        const size_t need = std::max(entries * stride, pv->data.size());
        sceneData.insert(sceneData.end(), pv->data.begin(), pv->data.end());
        sceneData.resize(mr.sdOffset[slot] + need, 0.0f);
      }
      meshRecs.push_back(mr);
    }
    for (const GiCVertex& vIn : m->vertices) { // Gi.cpp:848-861: quantise normal/tangent to octahedral unorm2x16, then decode once
      const GiCVertex v = usableShadingAttributes(vIn);
      FVertex fv; memcpy(fv.pos, v.pos, 12); fv.bsign = v.bitangentSign;
      decodeDirection(encodeDirection(v.norm), fv.normal); decodeDirection(encodeDirection(v.tangent), fv.tangent);
      fv.u = v.u; fv.v = v.v;
      verts.push_back(fv);
    }
    // FaceId AOV values, bug-compatible: face ids are stored with a 1/2/4-byte stride chosen from maxFaceId (Gi.cpp:878-885);
    // the shader fetches the 32-bit word prim / (4/stride), shifts it by (prim % (4/stride)) * 8 bits (sic) and masks it
    // with (stride*8 - 1) (rp_main.chit:231-240).  Evaluated once per primitive here.
    std::vector<int32_t> meshFaceIdAov(m->faces.size());
    {
      const int stride = m->maxFaceId <= 255u ? 1 : (m->maxFaceId <= 65535u ? 2 : 4), invStride = 4 / stride;
      std::vector<uint8_t> packed(((size_t)m->faces.size() * stride + 3) / 4 * 4, 0);
      for (size_t i = 0; i < m->faces.size(); i++) { int32_t fid = i < m->faceIds.size() ? m->faceIds[i] : 0; memcpy(&packed[i * stride], &fid, stride); }
      for (size_t i = 0; i < m->faces.size(); i++) {
        int32_t word; memcpy(&word, &packed[(i / (size_t)invStride) * 4], 4);
        word >>= (int)((i % (size_t)invStride) * 8);
        meshFaceIdAov[i] = word & (stride * 8 - 1);
      }
    }
    size_t instCount = m->instanceTransforms.size() / 16;
    m->builtInstances = (uint32_t)instCount;
    meshBuilds.push_back(MeshBuild{m, vertexOffset, matFlags, (uint32_t)instances.size(), (uint32_t)instCount, (uint32_t)tris.size(), meshIdx, meshFaceIdAov});
    for (size_t ii = 0; ii < instCount; ii++) { // Gi.cpp:1188-1202
      InstanceRec ir{};
      composeTransform(m->transform, &m->instanceTransforms[16 * ii], ir.o2w);
      invert3x3(ir.o2w, ir.w2o);
      ir.mesh = meshIdx; ir.instanceId = ii < m->instanceIds.size() ? m->instanceIds[ii] : (int32_t)ii;
      ir.pad = (uint32_t)m->id; // object id
      uint32_t instIdx = (uint32_t)instances.size();
      instances.push_back(ir);
      const bool usable = usableInstance(ir);
      for (uint32_t f = 0; f < (uint32_t)m->faces.size(); f++) {
        TriRec t;
        flattenTriangle(ir, usable, m, f, t);
        for (int a = 0; a < 3; a++) t.vi[a] = vertexOffset + m->faces[f].v_i[a];
        t.instance = instIdx; t.prim = f; t.origId = (uint32_t)tris.size(); t.matFlags = matFlags;
        tris.push_back(t);
        faceIdOf.push_back(meshFaceIdAov[f]);
      }
    }
    meshIdx++;
  }
  Bvh8& bvh = H.bvh;
  buildBvh8(tris, bvh);
  { std::vector<TriRec>().swap(tris); } // the BVH holds its own (leaf-ordered) copy
  s->stats.inactiveTriangleCount = (uint32_t)bvh.tris.size() - bvh.activeTris;
  if (bvh.activeTris < bvh.tris.size()) { // one line per mesh (bvh8.h "Inactive items")
    std::vector<uint32_t> perMesh(meshBuilds.size(), 0u);
    for (size_t i = bvh.activeTris; i < bvh.tris.size(); i++) perMesh[instances[bvh.tris[i].instance].mesh]++;
    for (const MeshBuild& mb : meshBuilds)
      if (perMesh[mb.meshIdx]) fprintf(stderr, "[gatling_gi] warning: mesh %s: %u of %zu instanced triangle(s) have a non-finite or out-of-range (> 1e18) vertex or a non-invertible transform and are inactive\n",
                                       mb.m->name.c_str(), perMesh[mb.meshIdx], mb.m->faces.size() * (size_t)mb.instCount);
  }
  if (buildTwoLevel(s, meshBuilds, instances, bvh.tris.size(), bvh.nodes.size(), H.two) != GI_C_OK) return GI_C_ERROR;
  if (s->twoLevel) {
    H.flatOfOrig.resize(bvh.tris.size());
    for (size_t i = 0; i < bvh.tris.size(); i++) H.flatOfOrig[bvh.tris[i].origId] = (uint32_t)i;
  }
  double t1 = nowMs();
  // the deepest traversal variant keeps 8 (SPILL8) or 16 stack entries in LDS and OVF_STACK = 40 in scratch; trav_node_pick does not bound-check the spill
  if (bvh.maxDepth > 1u + 8u + 40u) { setError("scene BVH is deeper than the traversal stack (49 levels): degenerate geometry (long chains of nested splits)"); return GI_C_ERROR; }
  if (bvh.tris.size() >= (1u << 26) && !s->twoLevel) { setError("scene has 2^26 or more triangles after instancing and no two-level layout (it is switched off, or its unique mesh triangles exceed 2^26 too): the traversal queues pack (lane, triangle) into 32 bits"); return GI_C_ERROR; }
  if (bvh.tris.size() >= (1u << 28)) { setError("scene has 2^28 or more triangles after instancing: the hit record packs (triangle, material class) into 32 bits"); return GI_C_ERROR; }
  H.triFaceId.resize(bvh.tris.size());
  for (size_t i = 0; i < bvh.tris.size(); i++) H.triFaceId[i] = faceIdOf[bvh.tris[i].origId];
  // Scenes beyond LDS: one 128-byte shading record per mesh triangle (gi_types.h TriShade); the flattened triangles name theirs in vi[0].  LDS-resident
  // scenes keep vertex indices there: the fused kernels are VALU-bound and read the host-decoded FVertex records.
  H.shadePacked = bvh.nodes.size() > 384u || bvh.tris.size() > 128u;
  H.triShade.clear();
  if (H.shadePacked) {
    std::vector<uint32_t> shadeBaseOfMesh(meshBuilds.size(), 0u);
    for (MeshBuild& mb : meshBuilds) {
      mb.shadeBase = (uint32_t)H.triShade.size(); shadeBaseOfMesh[mb.meshIdx] = mb.shadeBase;
      const GiCMesh* m = mb.m;
      for (const GiCFace& f : m->faces) {
        TriShade q{};
        for (int k = 0; k < 3; k++) {
          const GiCVertex v = usableShadingAttributes(m->vertices[f.v_i[k]]);
          memcpy(q.p[k], v.pos, 12); q.n[k] = encodeDirection(v.norm); q.t[k] = encodeDirection(v.tangent);
          q.uv[k][0] = v.u; q.uv[k][1] = v.v; q.bsign[k] = v.bitangentSign; q.vi[k] = mb.vertexOffset + f.v_i[k];
        }
        H.triShade.push_back(q);
      }
    }
    for (TriRec& t : bvh.tris) t.vi[0] = shadeBaseOfMesh[instances[t.instance].mesh] + t.prim;
  }
  s->shadePacked = H.shadePacked;
  s->shadowOrder = -1; s->shadowOrderRays[0] = s->shadowOrderRays[1] = s->shadowOrderSteps[0] = s->shadowOrderSteps[1] = 0; // a new tree: the shadow walks' order is chosen anew
  // one copy of the scene per device this scene renders on
  const uint32_t nDev = sceneDeviceCount(s);
  while (s->replicas.size() + 1u < nDev) { s->replicas.emplace_back(new SceneDevice()); s->replicas.back()->slot = (uint32_t)s->replicas.size(); s->dirty |= DIRTY_LIGHTS; } // a new replica has no lights yet
  for (uint32_t d = 0; d < nDev; d++)
    if (uploadSceneTo(s, sceneDevice(s, d), H) != GI_C_OK) { (void)hipSetDevice(g_ctx.device); return GI_C_ERROR; }
  HIP_TRY(hipSetDevice(g_ctx.device));
  s->nodeCount = (uint32_t)bvh.nodes.size(); s->triCount = (uint32_t)bvh.tris.size(); s->bvhDepth = bvh.maxDepth > 1u ? bvh.maxDepth - 1u : 1u; // stack entries a walk can need: a pick at level L pushes the rest of level L-1's group (gi_traversal.h trav_node_pick), the root level pushes nothing
  setSceneBounds(s, bvh.nodes);
  s->stats.bvhBuildMs = t1 - t0; s->stats.uploadMs = nowMs() - t1;
  s->stats.nodeCount = s->nodeCount; s->stats.triangleCount = s->triCount;
  if (getenv("GATLING_BUILD_TIMING")) fprintf(stderr, "[gatling_gi] scene: %u nodes, %u triangles, %u levels (traversal stack need %u)\n", s->nodeCount, s->triCount, bvh.maxDepth, s->bvhDepth);
  s->host = std::move(hostPtr);
  return GI_C_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Incremental transform updates (the reference keeps every mesh's BLAS and rebuilds only the TLAS, Gi.cpp:1180-1202).
//
// A scene is first built as ONE tree over all instanced triangles (buildScene: the best tree).  The first time only transforms change, it is re-laid out
// PARTITIONED: every flattened mesh instance gets its own subtree in its own node range and keeps its triangles in its own (scene-order) range; a top tree
// over the subtree roots (buildTopBvh8: the roots are copied in as ordinary internal children) makes it one ordinary BVH8 again -- the traversal kernels, the
// shading code and the triangle ids do not change, so images stay bit-identical to a full rebuild (traversal contract: results do not depend on the tree).
// From then on moving an instance costs: its triangles re-transformed, its subtree rebuilt (a few thousand triangles), the top tree rebuilt (one item per
// instance), and those ranges uploaded -- not a 10 M-triangle SAH build and a 0.7 GB upload.  Any other edit (geometry, materials, visibility, instance
// counts) raises DIRTY_BVH and the next render rebuilds everything as one tree again.
// ---------------------------------------------------------------------------------------------------------------
void nodeBounds(const Node8& n, float box[6])
{
  for (int a = 0; a < 3; a++) { box[a] = 3.0e38f; box[3 + a] = -3.0e38f; }
  for (int sl = 0; sl < 8; sl++) {
    if (n.meta[sl] == 0) continue;
    for (int a = 0; a < 3; a++) {
      uint32_t eb = (uint32_t)n.e[a] << 23; float scale; memcpy(&scale, &eb, 4);
      box[a] = std::min(box[a], n.p[a] + (float)n.qlo[a][sl] * scale); box[3 + a] = std::max(box[3 + a], n.p[a] + (float)n.qhi[a][sl] * scale);
    }
  }
  // the dequantised planes are evaluated in fp32 here and with an fma on the device: one more ulp-scale pad keeps the item box outside both
  for (int a = 0; a < 3; a++) { const float pad = (std::fabs(box[a]) + std::fabs(box[3 + a])) * 2.4e-7f + 1.0e-30f; box[a] -= pad; box[3 + a] += pad; }
}

// One instance's InstanceRec, world-space triangles (scene order) and subtree
struct PartBuild { InstanceRec inst; Bvh8 bvh; };
void buildPart(const MeshBuild& mb, uint32_t instInMesh, bool packed, PartBuild& out)
{
  const GiCMesh* m = mb.m;
  InstanceRec ir{};
  composeTransform(m->transform, &m->instanceTransforms[16 * (size_t)instInMesh], ir.o2w);
  invert3x3(ir.o2w, ir.w2o);
  ir.mesh = mb.meshIdx; ir.instanceId = instInMesh < m->instanceIds.size() ? m->instanceIds[instInMesh] : (int32_t)instInMesh;
  ir.pad = (uint32_t)m->id;
  out.inst = ir;
  const uint32_t nf = (uint32_t)m->faces.size(), instIdx = mb.instFirst + instInMesh;
  std::vector<TriRec> tris(nf);
  const bool usable = usableInstance(ir);
  for (uint32_t f = 0; f < nf; f++) { // as buildScene
    TriRec& t = tris[f];
    flattenTriangle(ir, usable, m, f, t);
    for (int a = 0; a < 3; a++) t.vi[a] = mb.vertexOffset + m->faces[f].v_i[a];
    t.instance = instIdx; t.prim = f; t.origId = f; t.matFlags = mb.matFlags;
    if (packed) t.vi[0] = mb.shadeBase + f;
  }
  buildBvh8(tris, out.bvh);
}

// writes a built part into the scene arrays at the part's ranges (node / triangle indices rebased to absolute)
void placePart(SceneHost& H, InstPart& P, const PartBuild& B)
{
  const MeshBuild& mb = H.meshBuilds[P.meshBuild];
  P.nodeCount = (uint32_t)B.bvh.nodes.size(); P.depth = B.bvh.maxDepth;
  for (uint32_t i = 0; i < P.nodeCount; i++) { Node8 n = B.bvh.nodes[i]; n.childBase += P.nodeOff; n.triBase += P.triFirst; H.bvh.nodes[P.nodeOff + i] = n; }
  for (uint32_t k = 0; k < P.nf; k++) {
    TriRec t = B.bvh.tris[k];
    H.triFaceId[P.triFirst + k] = mb.faceIdAov[t.prim];
    t.origId += P.triFirst; // scene-order id: the instance's triangles are numbered in face order from triFirst, as in buildScene
    H.bvh.tris[P.triFirst + k] = t;
  }
  H.instances[mb.instFirst + P.instInMesh] = B.inst;
  nodeBounds(H.bvh.nodes[P.nodeOff], P.box);
}

template <class Fn> void parallelOver(size_t n, Fn&& fn)
{
  int workers = (int)std::thread::hardware_concurrency();
  if (const char* e = getenv("GATLING_BUILD_THREADS")) workers = atoi(e);
  workers = (int)std::min<size_t>((size_t)std::min(std::max(workers, 1), 32), std::max<size_t>(n, 1));
  if (workers <= 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (int w = 0; w < workers; w++) th.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) fn(i); });
  for (auto& t : th) t.join();
}

int rebuildTop(GiCScene* s, SceneHost& H)
{
  std::vector<float> boxes(H.parts.size() * 6); std::vector<Node8> roots(H.parts.size());
  uint32_t subDepth = 0;
  for (size_t i = 0; i < H.parts.size(); i++) { memcpy(&boxes[6 * i], H.parts[i].box, 24); roots[i] = H.bvh.nodes[H.parts[i].nodeOff]; subDepth = std::max(subDepth, H.parts[i].depth); }
  Bvh8 top;
  buildTopBvh8(boxes.data(), H.parts.size(), roots.data(), top);
  if (top.nodes.size() > H.topCap) { setError("internal: top tree larger than its reserved range"); return GI_C_ERROR; }
  std::copy(top.nodes.begin(), top.nodes.end(), H.bvh.nodes.begin());
  for (size_t i = top.nodes.size(); i < H.topCap; i++) memset(&H.bvh.nodes[i], 0, sizeof(Node8));
  H.bvh.maxDepth = top.maxDepth + (subDepth > 0u ? subDepth - 1u : 0u); // the copied roots are the subtrees' first level
  if (H.bvh.maxDepth > 1u + 8u + 40u) { setError("scene BVH is deeper than the traversal stack (49 levels)"); return GI_C_ERROR; }
  s->bvhDepth = H.bvh.maxDepth > 1u ? H.bvh.maxDepth - 1u : 1u;
  return GI_C_OK;
}

// true: handled incrementally; false: the caller must run a full buildScene (not an error)
int updateTransforms(GiCScene* s, bool& handled)
{
  handled = false;
  if (!s->host || s->twoLevel || s->triCount < 4096u) return GI_C_OK; // small scenes rebuild in no time (and must stay LDS-resident)
  if (!optionValue("incremental", 1)) return GI_C_OK;
  SceneHost& H = *s->host;
  for (const MeshBuild& mb : H.meshBuilds) if (mb.m->builtInstances != mb.instCount) return GI_C_OK; // (cannot happen: count changes raise DIRTY_BVH)
  const double t0 = nowMs();
  std::vector<uint32_t> dirtyParts;
  bool converted = false;
  if (!H.partitioned) {
    // --- one-time re-layout: every instance its own subtree + ranges (costs about one full build, in parallel over the instances)
    std::vector<InstPart> parts;
    for (uint32_t b = 0; b < (uint32_t)H.meshBuilds.size(); b++) {
      const MeshBuild& mb = H.meshBuilds[b];
      const uint32_t nf = (uint32_t)mb.m->faces.size();
      for (uint32_t ii = 0; ii < mb.instCount; ii++) { InstPart P{}; P.meshBuild = b; P.instInMesh = ii; P.triFirst = mb.triFirst + ii * nf; P.nf = nf; parts.push_back(P); }
    }
    if (parts.empty()) return GI_C_OK;
    std::vector<PartBuild> built(parts.size());
    parallelOver(parts.size(), [&](size_t i) { buildPart(H.meshBuilds[parts[i].meshBuild], parts[i].instInMesh, H.shadePacked, built[i]); });
    H.topCap = (uint32_t)parts.size() * 2u + 16u; // top nodes <= internal top nodes + one copied root per part
    uint32_t off = H.topCap;
    for (size_t i = 0; i < parts.size(); i++) { const uint32_t n = (uint32_t)built[i].bvh.nodes.size(); parts[i].nodeOff = off; parts[i].nodeCap = n + n / 4u + 8u; off += parts[i].nodeCap; }
    H.bvh.nodes.assign(off, Node8{});
    H.parts.swap(parts);
    parallelOver(H.parts.size(), [&](size_t i) { placePart(H, H.parts[i], built[i]); });
    H.partitioned = true; converted = true;
  } else {
    for (uint32_t i = 0; i < (uint32_t)H.parts.size(); i++) {
      const GiCMesh* m = H.meshBuilds[H.parts[i].meshBuild].m;
      if (m->xformDirty && (m->instDirty.empty() || m->instDirty[H.parts[i].instInMesh])) dirtyParts.push_back(i);
    }
    std::vector<PartBuild> built(dirtyParts.size());
    parallelOver(dirtyParts.size(), [&](size_t k) { const InstPart& P = H.parts[dirtyParts[k]]; buildPart(H.meshBuilds[P.meshBuild], P.instInMesh, H.shadePacked, built[k]); });
    for (size_t k = 0; k < dirtyParts.size(); k++)
      if (built[k].bvh.nodes.size() > H.parts[dirtyParts[k]].nodeCap) { H.partitioned = false; H.parts.clear(); return GI_C_OK; } // a subtree outgrew its range (rare): full rebuild
    parallelOver(dirtyParts.size(), [&](size_t k) { placePart(H, H.parts[dirtyParts[k]], built[k]); });
  }
  if (rebuildTop(s, H) != GI_C_OK) return GI_C_ERROR;
  for (GiCMesh* m : s->meshes) { m->xformDirty = false; m->instDirty.clear(); }
  const double t1 = nowMs();
  // --- upload: everything after the re-layout, else the moved parts' ranges, their InstanceRecs and the top region
  const uint32_t nDev = std::min<uint32_t>(sceneDeviceCount(s), (uint32_t)s->replicas.size() + 1u);
  s->nodeCount = (uint32_t)H.bvh.nodes.size();
  setSceneBounds(s, H.bvh.nodes);
  for (uint32_t d = 0; d < nDev; d++) {
    SceneDevice& D = sceneDevice(s, d);
    if (converted) { if (uploadSceneTo(s, D, H) != GI_C_OK) { (void)hipSetDevice(g_ctx.device); return GI_C_ERROR; } continue; }
    const DevCtx& ctx = g_ctx.devs[d];
    HIP_TRY(hipSetDevice(ctx.device));
    hipStream_t st = ctx.stream;
    HIP_TRY(hipMemcpyAsync(D.dNodes.ptr, H.bvh.nodes.data(), (size_t)H.topCap * sizeof(Node8), hipMemcpyHostToDevice, st));
    for (uint32_t i : dirtyParts) {
      const InstPart& P = H.parts[i];
      const uint32_t instIdx = H.meshBuilds[P.meshBuild].instFirst + P.instInMesh;
      HIP_TRY(hipMemcpyAsync(D.dNodes.ptr + P.nodeOff, &H.bvh.nodes[P.nodeOff], (size_t)P.nodeCount * sizeof(Node8), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(D.dTris.ptr + P.triFirst, &H.bvh.tris[P.triFirst], (size_t)P.nf * sizeof(TriRec), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(D.dTriFaceId.ptr + P.triFirst, &H.triFaceId[P.triFirst], (size_t)P.nf * sizeof(int32_t), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(D.dInstances.ptr + instIdx, &H.instances[instIdx], sizeof(InstanceRec), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
  }
  HIP_TRY(hipSetDevice(g_ctx.device));
  s->stats.bvhBuildMs = t1 - t0; s->stats.uploadMs = nowMs() - t1;
  s->stats.nodeCount = s->nodeCount; s->stats.triangleCount = s->triCount;
  if (getenv("GATLING_BUILD_TIMING")) fprintf(stderr, "[gatling_gi] transform update: %s, %zu part(s) rebuilt of %zu, host %.1f ms, upload %.1f ms\n",
                                              converted ? "scene re-laid out as per-instance subtrees" : "incremental", converted ? H.parts.size() : dirtyParts.size(), H.parts.size(), t1 - t0, nowMs() - t1);
  handled = true;
  return GI_C_OK;
}

// brings the device scene up to date with the host-side edits: incremental for transform-only edits, else a full build
int syncSceneGeometry(GiCScene* s)
{
  if ((s->dirty & DIRTY_XFORM) && !(s->dirty & (DIRTY_BVH | DIRTY_MATERIALS))) { // only transforms changed: re-transform / re-braid those instances
    bool handled = false;
    if (updateTransforms(s, handled) != GI_C_OK) return GI_C_ERROR;
    if (!handled) s->dirty |= DIRTY_BVH;
    s->dirty |= DIRTY_FRAMEBUFFER;
  }
  if (s->dirty & (DIRTY_BVH | DIRTY_MATERIALS)) {
    if (buildScene(s) != GI_C_OK) return GI_C_ERROR;
    s->dirty &= ~(DIRTY_BVH | DIRTY_MATERIALS); s->dirty |= DIRTY_FRAMEBUFFER;
  }
  s->dirty &= ~DIRTY_XFORM;
  return GI_C_OK;
}

int uploadLights(GiCScene* s)
{
  const uint32_t nDev = std::min<uint32_t>(sceneDeviceCount(s), (uint32_t)s->replicas.size() + 1u);
  for (uint32_t d = 0; d < nDev; d++) {
    SceneDevice& D = sceneDevice(s, d);
    HIP_TRY(hipSetDevice(g_ctx.devs[d].device));
    hipStream_t st = g_ctx.devs[d].stream;
    if (D.dSphere.upload(s->sphereLights.recs, st) || D.dDistant.upload(s->distantLights.recs, st) || D.dRect.upload(s->rectLights.recs, st) ||
        D.dDisk.upload(s->diskLights.recs, st))
      return GI_C_ERROR;
    HIP_TRY(hipStreamSynchronize(st));
  }
  HIP_TRY(hipSetDevice(g_ctx.device));
  return GI_C_OK;
}

bool settingsEqual(const GiCRenderSettings& a, const GiCRenderSettings& b) { return memcmp(&a, &b, sizeof(a)) == 0; }

SceneView makeView(GiCScene* s, SceneDevice& D)
{
  SceneView v{};
  v.textures = D.dTextures.ptr; v.meshes = D.dMeshes.ptr; v.sceneData = D.dSceneData.ptr;
  v.nodes = D.dNodes.ptr; v.tris = D.dTris.ptr; v.instances = D.dInstances.ptr;
  v.verts = D.dVerts.ptr; v.triShade = D.dTriShade.ptr; v.shadePacked = s->shadePacked ? 1u : 0u; v.materials = D.dMaterials.ptr; v.sphereLights = D.dSphere.ptr; v.distantLights = D.dDistant.ptr;
  v.tlasNodes = D.dTlasNodes.ptr; v.tlasItems = D.dTlasItems.ptr; v.blasNodes = D.dBlasNodes.ptr; v.blasTris = D.dBlasTris.ptr; v.instTrav = D.dInstTrav.ptr; v.flatOfOrig = D.dFlatOfOrig.ptr; v.twoLevel = s->twoLevel ? 1u : 0u;
  v.rectLights = D.dRect.ptr; v.diskLights = D.dDisk.ptr; v.triFaceId = D.dTriFaceId.ptr; v.nodeCount = s->nodeCount; v.triCount = s->triCount; v.bvhDepth = s->bvhDepth; v.hasCutouts = s->hasCutouts ? 1u : 0u;
  return v;
}
SceneView makeView(GiCScene* s) { return makeView(s, *s); }

// k_trace_dyn refill threshold for scenes that do not fit LDS (0 = use the block-synchronous k_trace)
static uint32_t traceDynRefill(const GiCScene* s)
{
  uint32_t r = s->optTraceDyn >= 0 ? (uint32_t)s->optTraceDyn : 8u;
  if (optionSet("trace_dyn")) r = (uint32_t)std::max(0L, std::min(64L, optionValue("trace_dyn", 8)));
  if (r && optionValue("trace_dyn_spill8", 0)) r |= TRACE_DYN_SPILL8;
  return r;
}

uint32_t shardCapacity(size_t slots, uint32_t gridA, uint32_t gridB)
{
  // A queue holds at most `slots` records in total (a path sits in one queue at a time), but it is fed by SEVERAL launches before it is
  // consumed -- TRACE[par] by k_raygen and one k_shade per material class, REGEN by k_trace / k_route, every k_shade and k_raygen
  // (maxBounces == 0) -- and every launch starts dealing its blocks at shard 0.  A launch of G blocks that appends n records gives one
  // shard at most ceil(G/NSHARD) * ceil(n/(256 G)) * 256 <= n/NSHARD + n/G + 32 G + 256 of them; summed over P producers with
  // sum(n) <= slots this is slots/NSHARD + P * (slots/Gmin + 32 Gmax + 256).  block_append also raises Counters::overflow if a shard
  // ever runs past its capacity (giCRender then fails instead of returning a corrupt image).
  // (a producer that appends I records per thread and trip -- k_route: ROUTE_ITEMS, k_raygen: RAYGEN_ITEMS, gi_kernels.h APPEND_ITEMS_MAX -- deals 256 * I records per
  // block and trip: the slack term is 32 * I * G + 256 * I)
  const size_t P = 2 + MAT_CLASS_COUNT, I = APPEND_ITEMS_MAX;
  const size_t gMin = std::max<size_t>(1, std::min(gridA, gridB)), gMax = std::max<size_t>(1, std::max(gridA, gridB));
  const size_t cap = (slots + NSHARD - 1) / NSHARD + P * ((slots + gMin - 1) / gMin + 32 * I * gMax + 256 * I);
  return (uint32_t)std::min<size_t>(cap, slots + 256); // a shard can never hold more than the pool
}

int ensurePathState(SceneDevice* s, size_t slots, uint32_t gridA, uint32_t gridB)
{
  const uint32_t cap = shardCapacity(slots, gridA, gridB);
  int rc;
#define GI_ALLOC(x) do { rc = (x); if (rc != GI_C_OK) return rc; } while (0) /* GI_C_ERROR, or GI_C_OUT_OF_MEMORY_INTERNAL for the caller's fallback */
  GI_ALLOC(s->slots.alloc(slots));
  if (!s->dCounters.ptr) { GI_ALLOC(s->dCounters.alloc(1)); HIP_TRY(hipMemset(s->dCounters.ptr, 0, sizeof(Counters))); } // AOV-only renders never run k_init
  if (cap > s->queueCap) {
    const size_t n = (size_t)cap * NSHARD;
    for (uint32_t q = 0; q < Q_COUNT; q++) {
      const bool hasRecord = (q == Q_TRACE_A || q == Q_TRACE_B || q == Q_SHADOW); // (the HIT queues hold indices into the TRACE queue: gi_queues.h)
      GI_ALLOC(s->qSlot[q].alloc(n));
      if (hasRecord) { GI_ALLOC(s->qA[q].alloc(n)); GI_ALLOC(s->qB[q].alloc(n)); }
      if (q == Q_SHADOW) GI_ALLOC(s->qC[q].alloc(n));
      if (q == Q_TRACE_A || q == Q_TRACE_B) GI_ALLOC(s->qFresh[q - Q_TRACE_A].alloc(n));
    }
    s->queueCap = cap;
  }
#undef GI_ALLOC
  if (!s->hCounters) HIP_TRY(hipHostMalloc((void**)&s->hCounters, sizeof(Counters), hipHostMallocDefault));
  if (!s->hPoll) {
    HIP_TRY(hipHostMalloc((void**)&s->hPoll, sizeof(PaddedCounter) * Q_COUNT * NSHARD * SceneDevice::POLL_RING, hipHostMallocDefault));
    for (hipEvent_t& e : s->pollEvent) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  return GI_C_OK;
}

QueueSet makeQueueSet(SceneDevice* s)
{
  QueueSet qs{};
  for (uint32_t q = 0; q < Q_COUNT; q++) { qs.slot[q] = s->qSlot[q].ptr; qs.a[q] = s->qA[q].ptr; qs.b[q] = s->qB[q].ptr; qs.c[q] = s->qC[q].ptr; }
  qs.fresh[0] = s->qFresh[0].ptr; qs.fresh[1] = s->qFresh[1].ptr;
  qs.cap = s->queueCap;
  return qs;
}

hipEvent_t poolEvent(SceneDevice* s, size_t idx)
{
  while (s->eventPool.size() <= idx) { hipEvent_t e; (void)hipEventCreate(&e); s->eventPool.push_back(e); }
  return s->eventPool[idx];
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------
// giCRender
// ---------------------------------------------------------------------------------------------------------------
// C++ exceptions (allocation failure on a huge scene) must not cross the C ABI: the heavy entry points run through a guarded wrapper
static int giCRenderImpl(const GiCRenderParams* params);
extern "C" int giCRender(const GiCRenderParams* params)
{
  try { return giCRenderImpl(params); }
  catch (const std::exception& e) { setError(std::string("giCRender: ") + e.what()); return GI_C_ERROR; }
}
// One device's part of a render: the rows rowBegin, rowBegin + rowStride, ... < rowEnd of the frame, on device D.slot with D's copy of the scene.  Called
// under the scene mutex, after the dirty handling; with several devices, once per device from its own host thread (the bounce loop polls the queue sizes).
struct RenderJob { const GiCRenderParams* params; const GiCAovBinding* colorBinding; uint32_t width, height, rowBegin, rowEnd, rowStride, tileRows; uint8_t clear[GI_C_MAX_AOV_COMP_SIZE]; bool readback; };

static void* rbMem(GiCRenderBuffer* rb, uint32_t slot) { return (slot == 0u || rb->scratch) ? rb->deviceMem : rb->replicaMem[slot - 1u]; }

static int renderOnDevice(GiCScene* s, SceneDevice& D, const RenderJob& job)
{
  const DevCtx& ctx = g_ctx.devs[D.slot];
  HIP_TRY(hipSetDevice(ctx.device));
  hipStream_t st = ctx.stream;
  const GiCRenderParams* params = job.params;
  const GiCRenderSettings& rs = params->renderSettings;
  const GiCAovBinding* colorBinding = job.colorBinding;
  const uint32_t width = job.width, height = job.height, rowBegin = job.rowBegin, rowEnd = job.rowEnd, rowStride = job.rowStride, tileRows = job.tileRows;
  const uint8_t* clear = job.clear;
  (void)height; (void)rowEnd;
  D.stats.bvhBuildMs = s->stats.bvhBuildMs; D.stats.uploadMs = s->stats.uploadMs;
  // --- non-colour AOV bindings (Gi.h:36-56).  NEE, Bounces and ClockCycles follow whole paths: they are filled by the colour pass
  // (clear value first), see PathState.  ClockCycles is a deterministic cost proxy (ray segments per pixel), heat-mapped like the reference.
  AovTargets aovT{}; bool anyAov = false;
  GiCRenderBuffer* neeRb = nullptr; GiCRenderBuffer* bouncesRb = nullptr; GiCRenderBuffer* clockRb = nullptr;
  std::vector<GiCRenderBuffer*> aovBuffers;
  for (uint32_t i = 0; i < params->aovBindingCount; i++) {
    const GiCAovBinding& b = params->aovBindings[i];
    if (b.aovId == GI_C_AOV_COLOR) continue;
    GiCRenderBuffer* rb = b.renderBuffer;
    if (rb->width != width || rb->height != height) { setError("giCRender: AOV buffers must share one size"); return GI_C_ERROR; }
    if (b.aovId < 0 || b.aovId >= GI_C_AOV_COUNT) { setError("giCRender: bad AOV id"); return GI_C_ERROR; }
    memcpy(aovT.clear[b.aovId], b.clearValue, 16);
    const bool vec = rb->stride == 16;
    F4* v4 = vec ? reinterpret_cast<F4*>(rbMem(rb, D.slot)) : nullptr;
    bool produced = true;
    switch (b.aovId) {
      case GI_C_AOV_NORMAL: aovT.normal = v4; break; case GI_C_AOV_BARYCENTRICS: aovT.barycentrics = v4; break;
      case GI_C_AOV_TEXCOORDS: aovT.texcoords = v4; break; case GI_C_AOV_OPACITY: aovT.opacity = v4; break;
      case GI_C_AOV_TANGENTS: aovT.tangents = v4; break; case GI_C_AOV_BITANGENTS: aovT.bitangents = v4; break;
      case GI_C_AOV_THIN_WALLED: aovT.thinWalled = v4; break; case GI_C_AOV_DOUBLE_SIDED: aovT.doubleSided = v4; break;
      case GI_C_AOV_ALBEDO: aovT.albedo = v4; break;
      case GI_C_AOV_DEPTH: aovT.depth = vec ? nullptr : reinterpret_cast<float*>(rbMem(rb, D.slot)); break;
      case GI_C_AOV_OBJECT_ID: aovT.objectId = vec ? nullptr : reinterpret_cast<int32_t*>(rbMem(rb, D.slot)); break;
      case GI_C_AOV_FACE_ID: aovT.faceId = vec ? nullptr : reinterpret_cast<int32_t*>(rbMem(rb, D.slot)); break;
      case GI_C_AOV_INSTANCE_ID: aovT.instanceId = vec ? nullptr : reinterpret_cast<int32_t*>(rbMem(rb, D.slot)); break;
      case GI_C_AOV_NEE: if (vec) neeRb = rb; produced = false; break;
      case GI_C_AOV_BOUNCES: if (vec) bouncesRb = rb; produced = false; break;
      case GI_C_AOV_CLOCK_CYCLES: if (vec) clockRb = rb; produced = false; break;
      default: produced = false; break;
    }
    if (produced) {
      const bool wantsVec = !(b.aovId == GI_C_AOV_DEPTH || b.aovId == GI_C_AOV_OBJECT_ID || b.aovId == GI_C_AOV_FACE_ID || b.aovId == GI_C_AOV_INSTANCE_ID);
      if (wantsVec != vec) { setError("giCRender: AOV render buffer format does not match the AOV (Gi.cpp:302-316)"); return GI_C_ERROR; }
      anyAov = true; aovBuffers.push_back(rb);
    } else { // clear value everywhere (the host copy was filled by fillClearValues before the device threads started)
      HIP_TRY(hipMemcpyAsync(rbMem(rb, D.slot), rb->hostMem, rb->size, hipMemcpyHostToDevice, st));
    }
  }
  if (!colorBinding && !anyAov && !neeRb && !bouncesRb && !clockRb) { HIP_TRY(hipStreamSynchronize(st)); return GI_C_OK; }
  GiCRenderBuffer dummyColor{};
  GiCRenderBuffer* colorRb = colorBinding ? colorBinding->renderBuffer : nullptr;
  if (!colorRb && (neeRb || bouncesRb || clockRb)) { // the path-following debug AOVs need the colour pass: render it into a scratch buffer
    if (D.scratchColor.alloc((size_t)width * height)) return GI_C_ERROR;
    dummyColor.width = width; dummyColor.height = height; dummyColor.stride = 16; dummyColor.size = (size_t)width * height * 16;
    dummyColor.deviceMem = D.scratchColor.ptr; dummyColor.deviceOnly = true; dummyColor.scratch = true;
    colorRb = &dummyColor;
  }
  if (colorRb && colorRb->stride != 16) { setError("giCRender: colour AOV needs a Float32Vec4 buffer"); return GI_C_ERROR; }
  (void)dummyColor;

  // --- uniforms (Gi.cpp:2373-2426; camera terms rp_main.rgen:199-212 evaluated once on the host)
  const size_t pixels = (size_t)tileRows * width;
  // device -> host copy of the tile's rows (one 2D copy: the rows are rowStride image rows apart)
  auto copyTileRows = [&](GiCRenderBuffer* rb, size_t texel) -> hipError_t {
    const size_t off = (size_t)rowBegin * width * texel, rowBytes = (size_t)width * texel, pitch = rowBytes * rowStride;
    if (rowStride == 1u) return hipMemcpyAsync((uint8_t*)rb->hostMem + off, (uint8_t*)rbMem(rb, D.slot) + off, rowBytes * tileRows, hipMemcpyDeviceToHost, st);
    return hipMemcpy2DAsync((uint8_t*)rb->hostMem + off, pitch, (uint8_t*)rbMem(rb, D.slot) + off, pitch, rowBytes, tileRows, hipMemcpyDeviceToHost, st);
  };
  if (pixels == 0) return GI_C_OK;
  FrameUniforms U{};
  {
    const GiCCameraDesc& c = params->camera;
    auto norm3 = [](const float* v, float* o) { float inv = 1.0f / sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); o[0] = v[0] * inv; o[1] = v[1] * inv; o[2] = v[2] * inv; };
    norm3(c.forward, U.camFwd); norm3(c.up, U.camUp);
    memcpy(U.camPos, c.position, 12);
    U.camRight[0] = U.camFwd[1] * U.camUp[2] - U.camFwd[2] * U.camUp[1];
    U.camRight[1] = U.camFwd[2] * U.camUp[0] - U.camFwd[0] * U.camUp[2];
    U.camRight[2] = U.camFwd[0] * U.camUp[1] - U.camFwd[1] * U.camUp[0];
    float aspect = (float)width / (float)height;
    float H = 1.0f, W = H * aspect;
    float d = H / (2.0f * tanf(c.vfov * 0.5f));
    U.WX = W / (float)width; U.HY = H / (float)height;
    for (int a = 0; a < 3; a++) {
      float C = U.camPos[a] + U.camFwd[a] * d;
      U.L[a] = (C - U.camRight[a] * W * 0.5f) - U.camUp[a] * H * 0.5f;
    }
    U.lensRadius = (c.fStop > 0.0f) ? c.focalLength / (2.0f * c.fStop) : 0.0f;
    U.focusDistance = c.focusDistance;
    uint32_t cr = packHalf2x16(c.clipStart, c.clipEnd);
    U.clipNear = f16ToF32((uint16_t)(cr & 0xffffu)); U.clipFar = f16ToF32((uint16_t)(cr >> 16));
    float cv[4]; memcpy(cv, clear, 16);
    for (int a = 0; a < 3; a++) { // fallback dome texel: glm::u8vec4(bg * 255) as RGBA8 unorm (Gi.cpp:2194-2199)
      int q = (int)(cv[a] * 255.0f); if (q < 0) q = 0; if (q > 255) q &= 255;
      U.background[a] = (float)q / 255.0f;
    }
    U.exposureScale = exp2f(c.exposure);
    U.spp = rs.spp; U.sampleOffset = s->sampleOffset; U.invSpp = 1.0f / (float)rs.spp; U.sppF = (float)rs.spp; U.sampleOffsetF = (float)s->sampleOffset;
    U.invTotalSampleCount = 1.0f / float(s->sampleOffset + rs.spp);
    U.maxSampleValue = rs.maxSampleValue; U.rrInvMinTermProb = rs.rrInvMinTermProb; U.lightIntensityMultiplier = rs.lightIntensityMultiplier;
    U.metersPerSceneUnit = rs.metersPerSceneUnit;
    U.mediumStackSize = rs.mediumStackSize; U.maxVolumeWalkLength = rs.maxVolumeWalkLength;
    U.mediumStackSize = rs.mediumStackSize; U.maxVolumeWalkLength = rs.maxVolumeWalkLength;
    U.maxBounces = std::min(rs.maxBounces, 0xfffu); U.rrBounceOffset = rs.rrBounceOffset & 0xffffu;
    U.imageWidth = width; U.imageHeight = height; U.rowBegin = rowBegin; U.rowStride = rowStride; U.pixelCount = (uint32_t)pixels;
    U.flags = (rs.jitteredSampling ? FLAG_JITTER : 0u) | (rs.filterImportanceSampling ? FLAG_FIS : 0u) | (rs.depthOfField ? FLAG_DOF : 0u) |
              (rs.clippingPlanes ? FLAG_CLIP : 0u) | (rs.nextEventEstimation ? FLAG_NEE : 0u) | (rs.progressiveAccumulation ? FLAG_PROGRESSIVE : 0u);
    U.sphereCount = (uint32_t)s->sphereLights.recs.size(); U.distantCount = (uint32_t)s->distantLights.recs.size();
    U.rectCount = (uint32_t)s->rectLights.recs.size(); U.diskCount = (uint32_t)s->diskLights.recs.size();
    U.totalLightCount = U.sphereCount + U.distantCount + U.rectCount + U.diskCount;
  }

  double tStart = nowMs();
  uint32_t iters = 0, traceLaunches = 0;
  bool usedFused = false;
  size_t ev = 0;
  std::vector<int> evKind; // 0 raygen, 1 trace, 2 shade, 3 shadow
  struct IterRow { size_t evEnd; uint64_t traced, hits, shadow, ended, cont; };
  std::vector<IterRow> iterRows; // (GATLING_ITER_LOG)
  const bool timers = s->kernelTimers;
  uint64_t sampledIters = 0, totalIters = 0;
  SceneView view = makeView(s, D);
  { // dome light (Gi.cpp:2201-2238, 2384-2396): an image-less dome light is ignored, like one whose file failed to load
    const GiCDomeLight* dl = params->domeLight;
    auto tit = (dl && dl->texture) ? std::find(s->textures.begin(), s->textures.end(), dl->texture) : s->textures.end();
    view.domeTexture = tit != s->textures.end() ? (uint32_t)(tit - s->textures.begin()) + 1u : 0u;
    view.domeCameraVisible = rs.domeLightCameraVisible ? 1u : 0u;
    for (int a = 0; a < 4; a++) view.domeRotation[a] = dl ? dl->rotation[a] : (a == 3 ? 1.0f : 0.0f);
    for (int a = 0; a < 3; a++) { view.domeEmission[a] = dl ? dl->baseEmission[a] : 1.0f; view.background[a] = U.background[a]; view.cameraPosition[a] = params->camera.position[a]; }
    view.frame = rs.frame;
  }
  if (ensurePathState(&D, 1, 1, 1) != GI_C_OK) return GI_C_ERROR; // counters / pinned mirror exist even for AOV-only renders
  if (colorRb) {
    // --- work decomposition (DESIGN.md "Persistent path pool"): work item = (pixel, sample); the frame is cut into batches of
    // consecutive samples whose per-sample colour buffer fits the budget; a pool of `slots` paths is kept full from a running
    // work counter until the batch's items run out.
    auto envU64 = [](const char* key, uint64_t def) { return optionSet(key) ? (uint64_t)optionValue(key, 0) : def; };
    // Memory plan (r04).  The per-sample colour buffer wants to hold the whole frame's samples (every batch ends in a drain / a kernel tail: C2's 34 GB for 1024 spp at
    // 1080p in one batch 213.4 ms per step, in four 215.5) and scenes beyond LDS want a 64 Mi-slot pool (17 GB with its queues) -- on an empty 288 GB device.  A Hydra
    // plugin shares the device with other scenes, other processes and the host application, so the plan starts from what is FREE now (plus what this scene already
    // holds in these buffers, which is reused), and an allocation that still fails (someone else was faster) is answered with a smaller plan -- more batches first,
    // then a smaller pool -- never with a failed render while a workable plan exists.  Results do not depend on the plan (test_pool_and_batch_invariance).
    size_t memFree = 0, memTotal = 0; (void)hipMemGetInfo(&memFree, &memTotal);
    if (optionSet("assume_free_mb")) memFree = (size_t)optionValue("assume_free_mb", 0) << 20; // tests: plan as if this much were free (a planner overtaken by another allocation: the fallback below must recover)
    if (!D.memTotalMb) D.memTotalMb = std::max<uint64_t>(1, (uint64_t)(memTotal >> 20));
    uint64_t held = D.sampleBuf.bytes() + D.slots.bytes() + D.media.bytes();
    for (uint32_t q = 0; q < Q_COUNT; q++) held += D.qSlot[q].bytes() + D.qA[q].bytes() + D.qB[q].bytes() + D.qC[q].bytes();
    held += D.qFresh[0].bytes() + D.qFresh[1].bytes();
    const uint64_t availMb = ((uint64_t)memFree + held) >> 20;
    const uint64_t capMb = std::max<uint64_t>(1024, std::min<uint64_t>(49152, D.memTotalMb / 6)); // the budget of an empty device: 48 GiB of 288 GB
    const uint64_t defaultMb = std::max<uint64_t>(256, std::min<uint64_t>(capMb, availMb / 3));    // ... and a third of what is available now, 256 MiB at least
    const uint64_t budgetBytes = envU64("sample_buffer_mb", s->optSampleBufferMb ? s->optSampleBufferMb : defaultMb) << 20;
    // Pool size: a launch of k_trace_dyn ends when its longest ray ends, and ray cost is heavy-tailed in scenes beyond LDS
    // (a 100-step ray outlives the average one six times over), so those scenes get a pool large enough to amortise that
    // tail (measured on C3 at spp 64 / 256: 4 Mi slots 630, 16 Mi 790, 32 Mi 831 / 810, 64 Mi - / 868 Msamples/s); LDS-resident scenes have uniform, short rays.
    const bool sceneInLds = s->nodeCount <= 384u && s->triCount <= 128u;
    const uint64_t poolDefault = sceneInLds ? (4u << 20) : (64u << 20);
    const uint64_t poolMax = std::min<uint64_t>((1ull << 30) - 1ull, // regen-queue entries keep two flag bits above the slot index (REGEN_MISSED, REGEN_FRESH)
                                                std::max<uint64_t>(64, envU64("pool_slots", s->optPoolSlots ? s->optPoolSlots : poolDefault)));
    uint64_t batchSamples = std::min<uint64_t>(rs.spp, std::max<uint64_t>(1, budgetBytes / (pixels * 16)));
    batchSamples = std::min<uint64_t>(batchSamples, std::max<uint64_t>(1, 0xffffffffull / pixels)); // work ids stay 32-bit
    // LDS-resident scenes without medium stacks / dome images: the fused persistent kernel k_path (gi_path.hip) keeps the paths in
    // registers -- no pool, no queues; the stage kernels below remain the path for everything else (and on request: option / env)
    view.mediumStackSize = rs.mediumStackSize;
    bool fused = pathKernelSupports(view) && s->optFusedPath != 0;
    fused = fused && optionValue("fused", 1) != 0;
    usedFused = fused;
    // work order of the wavefront pipeline and layout of its per-sample buffer (gi_queues.h work_item); the fused kernels hand work out sample-major
    if (!fused && optionValue("work_order", WORK_ORDER_PIXEL_MAJOR_DEFAULT ? 1 : 0) != 0) U.flags |= FLAG_PIXEL_MAJOR;
    size_t slots = fused ? 1 : (size_t)std::min<uint64_t>(poolMax, (uint64_t)pixels * batchSamples);

    // persistent grids: blocks per CU limited by registers (<= 6 waves/SIMD for k_trace) and, for k_trace, by the LDS it stages
    uint32_t wideBlocks = 1u, traceBlocks = 1u;
    auto sizeGrids = [&]() {
      SceneView v0 = makeView(s, D);
      uint32_t ln, lt, ldsBytes; traceLdsLayout(v0, ln, lt, ldsBytes);
      uint32_t perCu = std::min<uint32_t>(6u, (160u * 1024u) / (ldsBytes + traceStaticLdsBytes() + 256u));
      const bool allLds = ln == v0.nodeCount && lt == v0.triCount && v0.triCount > 0u;
      if (!allLds && traceDynRefill(s)) perCu = 8u; // k_trace_dyn is persistent per wave: blocks beyond what is resident find the cursor exhausted
      uint32_t widePerCu = 8u;
      perCu = std::max(perCu, 1u); widePerCu = std::max(widePerCu, 1u);
      wideBlocks = (uint32_t)std::min<size_t>((slots + 255) / 256, (size_t)ctx.cuCount * widePerCu);
      traceBlocks = (uint32_t)std::min<size_t>((slots + 255) / 256, (size_t)ctx.cuCount * perCu);
    };
    sizeGrids();
    // HIT-queue entries (and giCTraceRays) hold a TRACE-queue RECORD index in 30 bits (HIT_INDEX_MASK); records run up to shardCapacity * NSHARD, which exceeds the
    // slot count by the shards' slack -- a pinned pool near 2^30 would push indices past the mask and k_shade would gather the wrong record
    while (!fused && (uint64_t)shardCapacity(slots, wideBlocks, traceBlocks) * NSHARD > 0x3fffffffull /* HIT_INDEX_MASK, gi_queues.h */) { slots -= slots / 8; sizeGrids(); }
    const uint32_t mediaStride = rs.mediumStackSize ? rs.mediumStackSize * MEDIUM_FLOATS + 4u : 0u;
    // what a plan costs: the slot pool with its queues (per slot: the Slot, the medium stack, and a share of every queue's records) and the sample buffer
    auto planBytes = [&](size_t nSlots, uint64_t nBatch) -> uint64_t {
      const uint64_t cap = shardCapacity(nSlots, wideBlocks, traceBlocks);
      const uint64_t perQueueEntry = 4ull * Q_COUNT + 32ull * (2 + 1) + 16ull + 8ull * 2;
      return (fused ? 0ull : (uint64_t)nSlots * (sizeof(Slot) + 4ull * mediaStride) + cap * NSHARD * perQueueEntry) + (uint64_t)pixels * nBatch * 16ull + (uint64_t)pixels * 16ull;
    };
    const bool pinnedPlan = optionSet("pool_slots") || s->optPoolSlots || optionSet("sample_buffer_mb") || s->optSampleBufferMb; // the caller's sizes are taken as given
    auto shrink = [&]() -> bool { // the next smaller plan: halve the sample buffer down to 64 MiB (more batches), then the pool down to 64 Ki slots
      if (batchSamples > 1 && (uint64_t)pixels * batchSamples * 16ull > (64ull << 20)) { batchSamples = std::max<uint64_t>(1, batchSamples / 2); if (!fused) slots = (size_t)std::min<uint64_t>(slots, (uint64_t)pixels * batchSamples); return true; }
      if (!fused && slots > (64u << 10)) { slots /= 2; return true; }
      return false;
    };
    if (!pinnedPlan) while (planBytes(slots, batchSamples) > (availMb << 20) - std::min<uint64_t>(availMb << 19, 512ull << 20) && shrink()) sizeGrids(); // (leave 512 MiB, or half of a tiny remainder)
    for (int attempt = 0;; attempt++) {
      int rc = fused ? GI_C_OK : ensurePathState(&D, slots, wideBlocks, traceBlocks);
      if (rc == GI_C_OK) rc = D.sampleBuf.alloc(pixels * batchSamples);
      if (rc == GI_C_OK) rc = D.accum.alloc(pixels);
      if (rc == GI_C_OK && mediaStride) rc = D.media.alloc(slots * mediaStride);
      if (rc == GI_C_OK) { if (attempt > 0) t_lastError.clear(); break; } // (a smaller plan fitted: the "out of memory" of the larger ones is not this render's error)
      if (rc != GI_C_OUT_OF_MEMORY_INTERNAL) return GI_C_ERROR;
      // out of memory: drop what this scene holds in the resizable buffers (a half-grown plan must not stand in the way of the smaller one) and try the next plan
      D.sampleBuf.release(); D.slots.release(); D.media.release();
      for (uint32_t q = 0; q < Q_COUNT; q++) { D.qSlot[q].release(); D.qA[q].release(); D.qB[q].release(); D.qC[q].release(); }
      D.qFresh[0].release(); D.qFresh[1].release(); D.queueCap = 0;
      if (attempt >= 40 || !shrink()) { setError("giCRender: out of device memory even with the smallest sample buffer and path pool"); return GI_C_ERROR; }
      sizeGrids();
    }
    const uint32_t numBatches = (uint32_t)((rs.spp + batchSamples - 1) / batchSamples);
    D.stats.poolSlots = fused ? 0u : (uint32_t)slots; D.stats.batches = numBatches;
    PathState ps{D.slots.ptr, D.media.ptr, mediaStride, nullptr, 0u, nullptr};
    if (neeRb && rs.nextEventEstimation) { // the reference compiles the NEE AOV write out with NEXT_EVENT_ESTIMATION (rp_main.rgen:397, 431)
      if (D.neeKey.alloc(pixels)) return GI_C_ERROR;
      HIP_TRY(hipMemsetAsync(D.neeKey.ptr, 0, pixels * sizeof(unsigned long long), st));
      ps.neeKey = D.neeKey.ptr;
    }
    if (bouncesRb) ps.bouncesAov = reinterpret_cast<F4*>(rbMem(bouncesRb, D.slot));
    if (clockRb) {
      if (D.pathSegments.alloc(pixels)) return GI_C_ERROR;
      HIP_TRY(hipMemsetAsync(D.pathSegments.ptr, 0, pixels * sizeof(uint32_t), st));
      ps.pathSegments = D.pathSegments.ptr;
    }
    view.mediumStackSize = rs.mediumStackSize;
    // Deferred Slot initialisation (r04): k_raygen hands a camera ray its (rng, work item) beside the ray record instead of writing the path's 64-byte Slot; the
    // slot is written where the first segment hits (k_route / k_trace) and a camera ray that leaves the scene retires there without ever touching one.  The
    // debug AOVs that follow whole paths read the slot when a sample retires (NEE / Bounces / ClockCycles): renders that bind them keep the eager form.
    if (!fused && optionValue("defer_slot", 1) != 0 && !ps.neeKey && !ps.bouncesAov && !ps.pathSegments) U.flags |= FLAG_DEFER_SLOT;
    QueueSet qs = makeQueueSet(&D);
    F4* colorOut = reinterpret_cast<F4*>(rbMem(colorRb, D.slot));
    const bool nee = rs.nextEventEstimation != 0;
    const uint32_t dynRefill = traceDynRefill(s);
    const int32_t shadowOrderNow = optionSet("shadow_order") ? (int32_t)optionValue("shadow_order", -1) : s->shadowOrder.load(); // (GATLING_OPTIONS=shadow_order=0|1 pins it)
    // Bounds retire (r04n): on the k_trace_dyn path a deferred-slot camera ray that cannot reach the scene's bounds is retired by k_raygen itself (C4: 58 % of the
    // camera rays, C3: ~45 %) -- same sample, same segment count, no ray record, no traversal step, no routing.  Not with a dome image / medium stack (a miss needs
    // the slot), not in counting builds (the root visit of such a ray is part of nodes-per-ray), not on the two-level layout (bounds of the TLAS root: not kept).
    {
      SceneView v0 = view; uint32_t ln, lt, ldsBytes; traceLdsLayout(v0, ln, lt, ldsBytes);
      const bool allLds = ln == v0.nodeCount && lt == v0.triCount && v0.triCount > 0u;
      if ((U.flags & FLAG_DEFER_SLOT) && !allLds && dynRefill && !view.twoLevel && view.domeTexture == 0u && rs.mediumStackSize == 0u && !s->countTraversal && s->boundsValid &&
          optionValue("bounds_retire", 1) != 0) {
        U.flags |= FLAG_BOUNDS_RETIRE;
        for (int a = 0; a < 3; a++) { U.sceneLo[a] = s->bounds[a]; U.sceneHi[a] = s->bounds[3 + a]; }
      }
    }

    // --- the bounce loop (rp_main.rgen:215, 295): every pool slot advances one stage per iteration
    HIP_TRY(hipStreamSynchronize(st));
    tStart = nowMs();
    // HIP events around the stage launches of every `timerStride`-th iteration (events on every launch cost ~16 % of the
    // frame); per-stage totals are scaled back up by the sampling factor.
    const uint32_t timerStride = std::max(1u, s->kernelTimerStride);
    uint64_t curIter = 0;
    auto timedOn = [&](hipStream_t on, int kind, auto&& fn) {
      if (timers && (curIter % timerStride) == 0u) { (void)hipEventRecord(poolEvent(&D, ev), on); fn(); (void)hipEventRecord(poolEvent(&D, ev + 1), on); ev += 2; evKind.push_back(kind); }
      else fn();
    };
    auto timed = [&](int kind, auto&& fn) { timedOn(st, kind, fn); };
    // Two streams (the reference's default frame is ONE sample per pixel, renderDelegate.cpp:93-110).  In a batch whose work fits the
    // pool every path starts in iteration 0, so from iteration 1 on k_raygen only FINISHES samples and the closest-hit launch of iteration i + 1 needs nothing from the
    // shadow launch of iteration i -- which k_raygen(i + 1) (it reads the radiance of paths that ended) and k_shade(i + 1) (it goes on adding to it: the float order of
    // rp_main.rgen:397-480) do need.  Such batches run
    //     main stream:    Z(i)  [R(0)]  T(i) + route(i)   <wait for Sh(i-1)>   [R(i), i > 0]   S(i)
    //     second stream:                                  <wait for S(i)>  Sh(i)
    // so that Sh(i) runs beside T(i + 1): an iteration lasts max(trace, shadow) + raygen + shade instead of their sum.  Per-path arithmetic and per-pixel sample order are
    // untouched (same kernels, same records); what changes is who zeroes which queue counter (gi_queues.h zero_next_counters / zero_closest_counters: Z = k_zero_closest).
    // Not with a dome image (a miss adds the dome's radiance to the Slot in k_route while the previous bounce's shadow launch may still be adding to it: two float
    // additions in an order that would depend on timing) or a medium stack.  GATLING_OPTIONS=two_stream=0 switches it off; two_stream_delay=1|2 (tests) holds the
    // main | the second stream back for 0.3 ms per iteration so that the other one runs ahead.
    const bool twoStreamOk = nee && rs.mediumStackSize == 0u && view.domeTexture == 0u && optionValue("two_stream", 1) != 0 && ctx.stream2 != nullptr;
    const long twoStreamDelay = optionValue("two_stream_delay", 0);
    hipStream_t st2 = ctx.stream2;
    if (twoStreamOk && !D.evShade) { HIP_TRY(hipEventCreateWithFlags(&D.evShade, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&D.evShadow, hipEventDisableTiming)); }
    const bool iterLog = timers && timerStride == 1u && getenv("GATLING_ITER_LOG") && atoi(getenv("GATLING_ITER_LOG")) != 0;
    for (uint32_t batch = 0; batch < numBatches; batch++) {
      U.batchFirstSample = (uint32_t)(batch * batchSamples);
      U.batchSamples = (uint32_t)std::min<uint64_t>(batchSamples, rs.spp - (uint64_t)batch * batchSamples);
      U.workTotal = (uint32_t)(pixels * U.batchSamples);
      ps.neeSampleBase = U.batchFirstSample;
      const uint32_t poolNow = (uint32_t)std::min<uint64_t>(slots, U.workTotal);
      U.poolSlots = poolNow;
      launchInit(st, ps, qs, D.dCounters.ptr, fused ? 0u : poolNow, batch == 0);
      if (fused && U.maxBounces != 0u) {
        // work items are claimed in chunks of consecutive ids; small frames get small chunks so that every resident wave finds work
        uint32_t chunk = 2048u;
        const uint64_t waves = (uint64_t)ctx.cuCount * 16u;
        // (a wave's last chunk is the launch's tail: 16 claims per wave keep it at ~6 % of a small frame -- C1 5 895 -> 6 360 Msamples/s; C2 does not care, 256 ... 2048 measure the same)
        chunk = (uint32_t)std::min<uint64_t>(chunk, std::max<uint64_t>(64u, ((uint64_t)U.workTotal / (waves * 16u)) & ~63ull));
        curIter = totalIters; if (timers) sampledIters++;
        if (timers) { (void)hipEventRecord(poolEvent(&D, ev), st); }
        // which fused kernel: k_path (one path per lane, in registers) unless the wave-local wavefront k_path_bw is asked for (GI_C_SCENE_OPTION_FUSED_PATH = 1 / GATLING_OPTIONS=path_bw=1).  Measured
        // r03 on C2 (1080p, spp 256, SLP vectorisation off): k_path 55.4 ms per batch, k_path_bw 57.3 -- k_path's 114 VGPRs give 4 resident waves per SIMD (3 blocks
        // per CU cost 11 %), k_path_bw's 168 VGPRs and 50 KB of LDS per block give 3; at 128 VGPRs k_path_bw spills 43 registers and falls to 84 ms.
        const int envBw = (int)optionValue("path_bw", -1);
        const bool useBw = !nee && (envBw >= 0 ? envBw != 0 : s->optFusedPath == 1);
        if (useBw) launchPathBw(st, (uint32_t)ctx.cuCount, s->classMask, s->classTextured != 0u, s->countTraversal, chunk, U, view, ps, D.dCounters.ptr, D.sampleBuf.ptr);
        else launchPath(st, (uint32_t)ctx.cuCount, s->classMask, s->classTextured != 0u, s->countTraversal, chunk, U, view, ps, D.dCounters.ptr, D.sampleBuf.ptr);
        if (timers) { (void)hipEventRecord(poolEvent(&D, ev + 1), st); ev += 2; evKind.push_back(1); }
        iters++; totalIters++; traceLaunches++;
        launchAccumulate(st, U, D.sampleBuf.ptr, D.accum.ptr, colorOut, batch == 0, batch + 1 == numBatches);
        continue;
      }
      if (U.maxBounces == 0u) {
        // rp_main.rgen:298-304: the bounce loop's exit test comes first, so with max-bounces 0 no ray is traced at all and every sample is
        // black (no emission at the primary hit, no dome / background term); the accumulation still runs (progressive blend, alpha 1)
        HIP_TRY(hipMemsetAsync(D.sampleBuf.ptr, 0, pixels * U.batchSamples * sizeof(F4), st));
        launchAccumulate(st, U, D.sampleBuf.ptr, D.accum.ptr, colorOut, batch == 0, batch + 1 == numBatches);
        continue;
      }
      const uint64_t rounds = ((uint64_t)U.workTotal + poolNow - 1) / poolNow; // raygen rounds needed to hand out all work
      const uint64_t maxIters = (rounds + 2) * (std::max(1u, U.maxBounces) + 1) + 8;
      const bool two = twoStreamOk && rounds == 1 && !iterLog;
      if (two) U.flags |= FLAG_TWO_STREAM; else U.flags &= ~FLAG_TWO_STREAM;
      bool shadowInFlight = false;
      for (uint64_t it = 0; it < maxIters; it++) {
        const uint32_t par = (uint32_t)(it & 1u);
        curIter = totalIters; if (timers && (totalIters % timerStride) == 0u) sampledIters++;
        auto raygen = [&] { timed(0, [&] { launchRaygen(st, wideBlocks, U, ps, qs, D.dCounters.ptr, par, D.sampleBuf.ptr); }); };
        // k_raygen(it) after the shadow launch of it - 1 (two streams, it > 0: it runs behind this iteration's closest-hit launch)
        auto raygenBehindShadow = [&] { if (shadowInFlight) { (void)hipStreamWaitEvent(st, D.evShadow, 0); shadowInFlight = false; } raygen(); };
        if (two) launchZeroClosest(st, D.dCounters.ptr, par);
        if (!two || it == 0) raygen();
        else if (rounds == 1 && it == (uint64_t)std::max(1u, U.maxBounces)) raygenBehindShadow(); // (the last k_raygen of the batch: the test below ends the loop)
        // A batch whose work fits the pool (a low-spp frame: hdGatling renders ONE sample per pixel and call) starts every path in iteration 0, a path traces at most
        // maxBounces segments, one per iteration (the bounce counter, rp_main.rgen:298-304) -- so k_raygen(maxBounces) has just retired the last samples and nothing is in
        // flight: no need to find that out two empty iterations later through the poll below (10 launches of ~90 in a spp-1 call).
        if (rounds == 1 && it == (uint64_t)std::max(1u, U.maxBounces) && rs.mediumStackSize == 0u) { totalIters++; break; }
        if (it >= rounds) {
          // All work cannot be handed out earlier.  From here on every iteration snapshots the queue sizes behind its k_raygen (asynchronous copy into a pinned ring)
          // and tests the snapshot of POLL_LAG iterations ago: the wait is for work the GPU finished long ago -- it still holds the iterations in between, so the
          // stream never runs dry -- and the loop stops at most POLL_LAG empty iterations after the pool drained.  (Until r03 the loop synchronised every 16th
          // iteration: C4 ran 15 empty iterations of 0.2 ms each, `tools/exp_iter_log.py`.)
          constexpr uint32_t R = SceneDevice::POLL_RING, LAG = SceneDevice::POLL_LAG;
          constexpr size_t snapshot = (size_t)Q_COUNT * NSHARD;
          HIP_TRY(hipMemcpyAsync(D.hPoll + (it % R) * snapshot, D.dCounters.ptr, sizeof(PaddedCounter) * snapshot, hipMemcpyDeviceToHost, st));
          HIP_TRY(hipEventRecord(D.pollEvent[it % R], st));
          if (it >= rounds + LAG) {
            const uint64_t j = it - LAG;
            HIP_TRY(hipEventSynchronize(D.pollEvent[j % R]));
            const PaddedCounter* snap = D.hPoll + (j % R) * snapshot + (size_t)(Q_TRACE_A + (uint32_t)(j & 1u)) * NSHARD;
            // (FLAG_BOUNDS_RETIRE: a k_raygen whose camera rays all miss the scene's bounds queues no ray either, but hands its slots on -- REGEN[(j&1)^1], zero at
            // this point otherwise -- and work is left)
            const PaddedCounter* again = D.hPoll + (j % R) * snapshot + (size_t)(Q_REGEN_A + (uint32_t)((j & 1u) ^ 1u)) * NSHARD;
            uint32_t pending = 0; for (uint32_t k = 0; k < NSHARD; k++) pending += snap[k].v + again[k].v;
            if (pending == 0) { totalIters++; break; } // k_raygen(j) consumed the regen queue and produced no rays: the pool had drained
          }
        }
        if (two && twoStreamDelay == 1) launchSpin(st, 300000ull);
        timed(1, [&] { launchTrace(st, traceBlocks, false, s->countTraversal, view, ps, qs, D.dCounters.ptr, Q_TRACE_A + par, Q_REGEN_A + (par ^ 1u), dynRefill, wideBlocks, U, D.sampleBuf.ptr); });
        traceLaunches++;
        if (two && it > 0) raygenBehindShadow();
        // one launch per shade class in use (scattering events inside a medium are routed to class 2, k_route: it is launched whenever a medium stack exists and OpenPBR does)
        const uint32_t shadeMask = s->shadeClassMask | ((rs.mediumStackSize != 0u && (s->shadeClassMask & (1u << SHADE_CLASS_OPBR_BASE))) ? 4u : 0u);
        for (uint32_t klass = 0; klass < MAT_CLASS_COUNT; klass++)
          if (shadeMask & (1u << klass)) timed(2, [&] { launchShade(st, wideBlocks, klass, (s->shadeClassTextured & (1u << klass)) != 0u, rs.mediumStackSize != 0u, U, view, ps, qs, D.dCounters.ptr, par); });
        if (nee) {
          // (the slot-order flag belongs to k_trace_dyn: with dynamic refill off -- TRACE_DYNAMIC 0 -- dynRefill stays 0 so that launchTrace picks the block-synchronous
          // k_trace the grid was sized for, and there is no order to measure)
          const int32_t order = (dynRefill & 0xffu) == 0u ? 0 : (shadowOrderNow >= 0 ? shadowOrderNow : (int32_t)(totalIters & 1u)); // not chosen yet: alternate, and count (below)
          hipStream_t on = st;
          if (two) { // the shadow launch moves to the second stream, behind this iteration's k_shade
            HIP_TRY(hipEventRecord(D.evShade, st)); HIP_TRY(hipStreamWaitEvent(st2, D.evShade, 0));
            if (twoStreamDelay == 2) launchSpin(st2, 300000ull);
            on = st2;
          }
          timedOn(on, 3, [&] { launchTrace(on, traceBlocks, true, s->countTraversal, view, ps, qs, D.dCounters.ptr, Q_SHADOW, Q_SHADOW, dynRefill | (order ? TRACE_DYN_SLOT_ORDER : 0u), wideBlocks, U, D.sampleBuf.ptr); });
          if (two) { HIP_TRY(hipEventRecord(D.evShadow, st2)); shadowInFlight = true; }
        }
        if (iterLog) { // (GATLING_ITER_LOG, with kernel timers on every iteration: what each iteration's queues held -- one sync per iteration, for measurements only)
          HIP_TRY(hipMemcpyAsync(D.hCounters, D.dCounters.ptr, sizeof(PaddedCounter) * Q_COUNT * NSHARD, hipMemcpyDeviceToHost, st));
          HIP_TRY(hipStreamSynchronize(st));
          auto total = [&](uint32_t q) { uint64_t n = 0; for (uint32_t k = 0; k < NSHARD; k++) n += D.hCounters->count[q][k].v; return n; };
          uint64_t hits = 0; for (uint32_t c = 0; c < MAT_CLASS_COUNT; c++) hits += total(Q_HIT + c);
          iterRows.push_back({ev, total(Q_TRACE_A + par), hits, total(Q_SHADOW), total(Q_REGEN_A + (par ^ 1u)), total(Q_TRACE_A + (par ^ 1u))});
        }
        iters++; totalIters++;
      }
      if (shadowInFlight) { (void)hipStreamWaitEvent(st, D.evShadow, 0); shadowInFlight = false; } // (a batch that ended through the poll: its last shadow launches were empty)
      launchAccumulate(st, U, D.sampleBuf.ptr, D.accum.ptr, colorOut, batch == 0, batch + 1 == numBatches);
    }
    U.flags &= ~FLAG_TWO_STREAM;
    if (neeRb && ps.neeKey) launchResolveNee(st, U, D.neeKey.ptr, reinterpret_cast<F4*>(rbMem(neeRb, D.slot)), (uint32_t)pixels);
    if (clockRb) { // ClockCycles: per-pixel cost -> heat map normalised to the frame maximum, on the host like _EncodeRenderBufferAsHeatmap (Gi.cpp:327-343)
      std::vector<uint32_t> counts(pixels);
      HIP_TRY(hipMemcpyAsync(counts.data(), D.pathSegments.ptr, pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      float maxValue = 0.0f;
      for (uint32_t c : counts) maxValue = std::max(maxValue, (float)c);
      float* img = reinterpret_cast<float*>(clockRb->hostMem);
      for (size_t p = 0; p < pixels; p++) {
        const size_t y = rowBegin + (p / width) * rowStride, x = p % width;
        float* o = img + (y * width + x) * 4;
        if (maxValue > 0.0f) {
          const int idx = std::min((int)(((float)counts[p] / maxValue) * 255.0), 255);
          turboColormap((float)idx / 255.0f, o);
          o[3] = 255.0f;
        } else { o[0] = (float)counts[p]; o[1] = 0.0f; o[2] = 0.0f; }
      }
      HIP_TRY(hipMemcpyAsync(rbMem(clockRb, D.slot), clockRb->hostMem, clockRb->size, hipMemcpyHostToDevice, st));
    }
    if (bouncesRb && U.maxBounces == 0u) { // rp_main.rgen:483-486 evaluates inferno(0 / 0) = NaN for every pixel of the tile
      float* img = reinterpret_cast<float*>(bouncesRb->hostMem);
      for (size_t p = 0; p < pixels; p++) { float* o = img + ((rowBegin + (p / width) * rowStride) * width + p % width) * 4; o[0] = o[1] = o[2] = NAN; }
      HIP_TRY(hipMemcpyAsync(rbMem(bouncesRb, D.slot), bouncesRb->hostMem, bouncesRb->size, hipMemcpyHostToDevice, st));
    }
    for (GiCRenderBuffer* rb : {neeRb, bouncesRb}) {
      if (!rb || rb->deviceOnly || !job.readback) continue;
      HIP_TRY(copyTileRows(rb, rb->stride));
    }
  }
  if (anyAov) { // the non-colour AOV pass (k_aov) + read-back of the rows of this tile
    launchAov(st, U, view, aovT);
    if (hipGetLastError() != hipSuccess) { setError("k_aov launch failed"); return GI_C_ERROR; }
    for (GiCRenderBuffer* rb : aovBuffers) {
      if (rb->deviceOnly || !job.readback) continue;
      HIP_TRY(copyTileRows(rb, rb->stride));
    }
  }
  HIP_TRY(hipMemcpyAsync(D.hCounters, D.dCounters.ptr, sizeof(Counters), hipMemcpyDeviceToHost, st));
  if (colorRb && !colorRb->deviceOnly && job.readback) {
    HIP_TRY(copyTileRows(colorRb, 16));
  }
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  double tEnd = nowMs();

  GiCRenderStats& S = D.stats;
  S.renderMs = tEnd - tStart; S.samples = (uint64_t)pixels * rs.spp; S.iterations = iters; S.traceLaunches = traceLaunches; S.fusedPath = usedFused ? 1u : 0u;
  S.segments = D.hCounters->segments; S.shadowRays = D.hCounters->shadowRays; S.nodesVisited = D.hCounters->nodesVisited; S.trisTested = D.hCounters->trisTested;
  S.shadowNodesVisited = D.hCounters->shadowNodesVisited; S.shadowTrisTested = D.hCounters->shadowTrisTested;
  if (D.slot == 0u && colorRb && s->shadowOrder.load() < 0) { // (colorRb: an AOV-only render never ran k_init -- the counters would be the previous render's) choose the shadow walks' order once both have been measured on enough rays of this scene: fewer node visits per ray wins
    for (int m = 0; m < 2; m++) {
      s->shadowOrderRays[m] += D.hCounters->shadowOrderRays[m];
      for (int k = 0; k < 16; k++) s->shadowOrderSteps[m] += D.hCounters->shadowOrderSteps[m][k].v;
    }
    constexpr uint64_t ENOUGH = 1u << 16;
    if (s->shadowOrderRays[0] >= ENOUGH && s->shadowOrderRays[1] >= ENOUGH)
    {
      s->shadowOrder = (double)s->shadowOrderSteps[1] * (double)s->shadowOrderRays[0] < (double)s->shadowOrderSteps[0] * (double)s->shadowOrderRays[1] ? 1 : 0;
      if (getenv("GATLING_BUILD_TIMING")) fprintf(stderr, "[gatling_gi] shadow walks: near-to-far %.3f node visits per ray (%llu rays), slot order %.3f (%llu rays) -> %s\n",
                                                  (double)s->shadowOrderSteps[0] / (double)s->shadowOrderRays[0], (unsigned long long)s->shadowOrderRays[0],
                                                  (double)s->shadowOrderSteps[1] / (double)s->shadowOrderRays[1], (unsigned long long)s->shadowOrderRays[1], s->shadowOrder.load() ? "slot order" : "near-to-far");
    }
  }
  if (s->countTraversal && D.hCounters->phaseTrips && optionValue("phase_stats", 0)) { // k_path's phase split (counting build)
    const Counters& c = *D.hCounters; const double tot = (double)(c.phaseCycles[0] + c.phaseCycles[1] + c.phaseCycles[2] + c.phaseCycles[3]);
    static const char* names[4] = {"regen", "trace", "shade", "shadow+finish"};
    for (int k = 0; k < 4; k++) fprintf(stderr, "[gatling_gi] k_path phase %-14s %5.1f %% of wave cycles, %5.1f of 64 lanes busy per trip\n", names[k], 100.0 * (double)c.phaseCycles[k] / tot, (double)c.phaseLanes[k] / (double)c.phaseTrips);
    fprintf(stderr, "[gatling_gi] k_path trips %llu, %.0f cycles per trip and wave\n", (unsigned long long)c.phaseTrips, tot / (double)c.phaseTrips);
  }
  if (s->countTraversal && D.hCounters->dynStats[0] && optionValue("phase_stats", 0)) { // k_trace_dyn's lane accounting (counting build, closest-hit launches)
    const unsigned long long* d = D.hCounters->dynStats; const double st = (double)d[0];
    fprintf(stderr, "[gatling_gi] k_trace_dyn<closest> %llu wave steps: per step %.1f lanes hold a ray, %.1f run the node test, %.1f wait for the triangle ring; %.3f batches per step of %.1f pairs; "
                    "a refill every %.2f steps, %.1f lanes each\n", d[0], (double)d[1] / st, (double)d[2] / st, (double)d[3] / st, (double)d[4] / st, d[4] ? (double)d[5] / (double)d[4] : 0.0,
            d[6] ? st / (double)d[6] : 0.0, d[6] ? (double)d[7] / (double)d[6] : 0.0);
  }
  if (D.hCounters->overflow) { setError("giCRender: a work-queue shard overflowed its capacity (internal sizing error); the image is invalid"); return GI_C_ERROR; }
  S.traceMs = S.shadeMs = S.raygenMs = S.shadowMs = 0.0;
  if (timers) {
    for (size_t k = 0; k < evKind.size(); k++) {
      float ms = 0.0f; (void)hipEventElapsedTime(&ms, D.eventPool[2 * k], D.eventPool[2 * k + 1]);
      if (evKind[k] == 0) S.raygenMs += ms; else if (evKind[k] == 1) S.traceMs += ms; else if (evKind[k] == 2) S.shadeMs += ms; else S.shadowMs += ms;
    }
    if (!iterRows.empty()) { // per iteration: the stage times of its launches and what its queues held
      size_t k = 0;
      for (size_t r = 0; r < iterRows.size(); r++) {
        double ms4[4] = {0.0, 0.0, 0.0, 0.0};
        for (; 2 * k < iterRows[r].evEnd && k < evKind.size(); k++) { float ms = 0.0f; (void)hipEventElapsedTime(&ms, D.eventPool[2 * k], D.eventPool[2 * k + 1]); ms4[evKind[k]] += ms; }
        fprintf(stderr, "[gatling_gi] iter %3zu: rays %9llu hits %9llu shadow %9llu ended %9llu continuing %9llu | raygen %7.3f trace+route %7.3f shade %7.3f shadow %7.3f ms\n", r,
                (unsigned long long)iterRows[r].traced, (unsigned long long)iterRows[r].hits, (unsigned long long)iterRows[r].shadow, (unsigned long long)iterRows[r].ended,
                (unsigned long long)iterRows[r].cont, ms4[0], ms4[1], ms4[2], ms4[3]);
      }
    }
    // scale the sampled totals to the whole frame (the early-exit poll can leave one raygen-only iteration unsampled)
    const double scale = sampledIters ? (double)totalIters / (double)sampledIters : 1.0;
    S.raygenMs *= scale; S.traceMs *= scale; S.shadeMs *= scale; S.shadowMs *= scale;
  }
  return GI_C_OK;
}

// The frame on nDev devices: rows d, d + nDev, ... on device d (interleaved shares cost the same), every device from its own host thread;
// then the shares of devices 1 .. n-1 are copied INTO PLACE in the primary device's render buffers (strided 2-D peer copies over xGMI: no staging buffer, no
// re-interleave pass) and the primary does the one D2H.  Per-pixel arithmetic does not depend on which device renders a row (RNG streams use the global
// pixel index), so the image is bit-identical to a one-device render.
static int renderOnDevices(GiCScene* s, uint32_t nDev, const RenderJob& frame)
{
  const GiCRenderParams* params = frame.params;
  // per-device copies of every bound render buffer
  for (uint32_t i = 0; i < params->aovBindingCount; i++) {
    GiCRenderBuffer* rb = params->aovBindings[i].renderBuffer;
    if (rb->replicaMem.size() + 1u < g_ctx.devs.size()) rb->replicaMem.resize(g_ctx.devs.size() - 1u, nullptr);
    for (uint32_t d = 1; d < nDev; d++) {
      if (rb->replicaMem[d - 1u]) continue;
      HIP_TRY(hipSetDevice(g_ctx.devs[d].device));
      HIP_TRY(hipMalloc(&rb->replicaMem[d - 1u], rb->size ? rb->size : 16));
      HIP_TRY(hipMemset(rb->replicaMem[d - 1u], 0, rb->size ? rb->size : 16));
    }
  }
  HIP_TRY(hipSetDevice(g_ctx.device));
  std::vector<int> rcs(nDev, GI_C_OK); std::vector<std::string> errs(nDev);
  auto work = [&](uint32_t d) {
    RenderJob job = frame;
    job.rowBegin = d; job.rowEnd = frame.height; job.rowStride = nDev; job.tileRows = (frame.height - d + nDev - 1u) / nDev; job.readback = false;
    try { rcs[d] = renderOnDevice(s, sceneDevice(s, d), job); }
    catch (const std::exception& e) { rcs[d] = GI_C_ERROR; t_lastError = e.what(); }
    if (rcs[d] != GI_C_OK) errs[d] = t_lastError; // (thread-local)
  };
  {
    std::lock_guard<std::mutex> own(g_ctx.workerMutex);
    while (g_ctx.workers.size() + 1u < nDev) { g_ctx.workers.emplace_back(new DeviceWorker()); g_ctx.workers.back()->start(); }
    for (uint32_t d = 1; d < nDev; d++) g_ctx.workers[d - 1u]->post([&work, d] { work(d); });
    work(0u);
    for (uint32_t d = 1; d < nDev; d++) g_ctx.workers[d - 1u]->wait();
  }
  HIP_TRY(hipSetDevice(g_ctx.device));
  for (uint32_t d = 0; d < nDev; d++) if (rcs[d] != GI_C_OK) { setError("device " + std::to_string(g_ctx.devs[d].device) + ": " + errs[d]); return GI_C_ERROR; }
  // gather: rows d::nDev of device d -> the same rows of the primary's buffer, then the D2H of the whole frame
  hipStream_t st = g_ctx.stream;
  for (uint32_t i = 0; i < params->aovBindingCount; i++) {
    GiCRenderBuffer* rb = params->aovBindings[i].renderBuffer;
    const size_t rowBytes = (size_t)rb->width * rb->stride, pitch = rowBytes * nDev;
    for (uint32_t d = 1; d < nDev; d++) {
      const uint32_t rows = (rb->height - d + nDev - 1u) / nDev;
      const size_t off = (size_t)d * rowBytes;
      if (g_ctx.devs[d].peer == 1 && optionValue("peer_copies", 1) != 0) { // strided 2-D peer copy over xGMI, straight into place
        HIP_TRY(hipMemcpy2DAsync((uint8_t*)rb->deviceMem + off, pitch, (uint8_t*)rb->replicaMem[d - 1u] + off, pitch, rowBytes, rows, hipMemcpyDefault, st));
      } else { // no peer access (or GATLING_OPTIONS=peer_copies=0): the share's rows -> pinned staging frame -> the primary's buffer, both strided, in place
        if (!rb->stageMem) HIP_TRY(hipHostMalloc(&rb->stageMem, rb->size ? rb->size : 16, hipHostMallocPortable));
        HIP_TRY(hipSetDevice(g_ctx.devs[d].device));
        HIP_TRY(hipMemcpy2DAsync((uint8_t*)rb->stageMem + off, pitch, (uint8_t*)rb->replicaMem[d - 1u] + off, pitch, rowBytes, rows, hipMemcpyDeviceToHost, g_ctx.devs[d].stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.devs[d].stream));
        HIP_TRY(hipSetDevice(g_ctx.device));
        HIP_TRY(hipMemcpy2DAsync((uint8_t*)rb->deviceMem + off, pitch, (uint8_t*)rb->stageMem + off, pitch, rowBytes, rows, hipMemcpyHostToDevice, st));
      }
    }
    if (!rb->deviceOnly) HIP_TRY(hipMemcpyAsync(rb->hostMem, rb->deviceMem, rb->size, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  // statistics: counts add up, times are the slowest device's
  GiCRenderStats S = s->stats; // (device 0's, written by its renderOnDevice)
  for (uint32_t d = 1; d < nDev; d++) {
    const GiCRenderStats& R = sceneDevice(s, d).stats;
    S.samples += R.samples; S.segments += R.segments; S.shadowRays += R.shadowRays; S.nodesVisited += R.nodesVisited; S.trisTested += R.trisTested;
    S.shadowNodesVisited += R.shadowNodesVisited; S.shadowTrisTested += R.shadowTrisTested;
    S.renderMs = std::max(S.renderMs, R.renderMs); S.traceMs = std::max(S.traceMs, R.traceMs); S.shadeMs = std::max(S.shadeMs, R.shadeMs);
    S.raygenMs = std::max(S.raygenMs, R.raygenMs); S.shadowMs = std::max(S.shadowMs, R.shadowMs); S.iterations = std::max(S.iterations, R.iterations);
    S.traceLaunches = std::max(S.traceLaunches, R.traceLaunches);
  }
  s->stats = S;
  return GI_C_OK;
}

static int giCRenderImpl(const GiCRenderParams* params)
{
  if (!g_ctx.initialized) { setError("giCRender before giCInitialize"); return GI_C_ERROR; }
  if (!params || !params->scene) { setError("giCRender: null params/scene"); return GI_C_ERROR; }
  GiCScene* s = params->scene;
  const GiCRenderSettings& rs = params->renderSettings;
  const GiCAovBinding* colorBinding = nullptr;
  for (uint32_t i = 0; i < params->aovBindingCount; i++) {
    if (!params->aovBindings[i].renderBuffer) { setError("giCRender: AOV binding without render buffer"); return GI_C_ERROR; }
    if (params->aovBindings[i].aovId == GI_C_AOV_COLOR) colorBinding = &params->aovBindings[i];
  }
  if (params->aovBindingCount == 0) { setError("giCRender: no AOV bindings"); return GI_C_ERROR; }
  if (rs.spp == 0) { setError("giCRender: spp must be > 0"); return GI_C_ERROR; }
  if (rs.mediumStackSize > MAX_MEDIUM_STACK) { setError("giCRender: mediumStackSize > 15 cannot be addressed (the payload's medium index has four bits, rp_main_payload.glsl:4-5)"); return GI_C_ERROR; }
  const GiCRenderBuffer* sizeRb = (colorBinding ? colorBinding : &params->aovBindings[0])->renderBuffer;
  const uint32_t width = sizeRb->width, height = sizeRb->height;
  if (width == 0 || height == 0) return GI_C_OK; // Render.Empty-style degenerate target: nothing to do
  if (width > 65535u || height > 65535u) { setError("giCRender: image dimensions exceed 65535 (imageDims packing, rp_main.h:38)"); return GI_C_ERROR; }
  { // a camera the ray generation can use (the reference passes whatever Hydra hands it, Gi.cpp:2373-2426; a NaN there is a NaN image): refused, with the field named
    const GiCCameraDesc& c = params->camera;
    const float fields[] = {c.position[0], c.position[1], c.position[2], c.forward[0], c.forward[1], c.forward[2], c.up[0], c.up[1], c.up[2],
                            c.vfov, c.fStop, c.focusDistance, c.focalLength, c.clipStart, c.clipEnd, c.exposure};
    for (float f : fields) if (!std::isfinite(f)) { setError("giCRender: the camera has a non-finite field"); return GI_C_ERROR; }
    const float f2 = (c.forward[0] * c.forward[0] + c.forward[1] * c.forward[1]) + c.forward[2] * c.forward[2], u2 = (c.up[0] * c.up[0] + c.up[1] * c.up[1]) + c.up[2] * c.up[2];
    if (!(f2 > 0.0f) || !(u2 > 0.0f) || !std::isfinite(f2) || !std::isfinite(u2) || !std::isfinite(1.0f / sqrtf(f2)) || !std::isfinite(1.0f / sqrtf(u2))) { setError("giCRender: the camera's forward or up vector cannot be normalised (zero, denormal or overflowing length)"); return GI_C_ERROR; }
    if (!(c.vfov > 0.0f && c.vfov < 3.14159265f)) { setError("giCRender: the camera's vertical field of view must lie inside (0, pi) radians"); return GI_C_ERROR; }
    if (!std::isfinite(1.0f / (2.0f * tanf(c.vfov * 0.5f)))) { setError("giCRender: the camera's vertical field of view is too small for the image plane distance to be finite"); return GI_C_ERROR; }
  }
  { // render settings that enter the arithmetic as floats
    const float fields[] = {rs.rrInvMinTermProb, rs.lightIntensityMultiplier, rs.metersPerSceneUnit, rs.frame};
    for (float f : fields) if (!std::isfinite(f)) { setError("giCRender: the render settings have a non-finite field"); return GI_C_ERROR; }
    if (std::isnan(rs.maxSampleValue)) { setError("giCRender: maxSampleValue is NaN"); return GI_C_ERROR; } // (+inf: no clamp)
  }
  uint32_t rowBegin = params->rowBegin, rowEnd = params->rowEnd ? params->rowEnd : height;
  const uint32_t rowStride = params->rowStride ? params->rowStride : 1u;
  if (rowBegin > rowEnd || rowEnd > height) { setError("giCRender: bad row range"); return GI_C_ERROR; }
  const uint32_t tileRows = rowEnd > rowBegin ? (rowEnd - rowBegin + rowStride - 1u) / rowStride : 0u; // rows rowBegin + k * rowStride < rowEnd

  std::lock_guard<std::mutex> guard(s->mutex);
  HIP_TRY(hipSetDevice(g_ctx.device));

  // --- dirty handling (_CalcDirtyFlagsForRenderParams, Gi.cpp:1859-1987; sample offset reset :2125-2129)
  uint8_t clear[GI_C_MAX_AOV_COMP_SIZE] = {0};
  if (colorBinding) memcpy(clear, colorBinding->clearValue, GI_C_MAX_AOV_COMP_SIZE);
  const float* domeEm = params->domeLight ? params->domeLight->baseEmission : nullptr;
  if (!s->haveOldParams || memcmp(&s->oldCamera, &params->camera, sizeof(GiCCameraDesc)) != 0 || !settingsEqual(s->oldSettings, rs) ||
      memcmp(s->oldClear, clear, sizeof(clear)) != 0 || s->oldRowBegin != rowBegin || s->oldRowEnd != rowEnd || s->oldRowStride != rowStride || s->oldDome != params->domeLight ||
      (domeEm && memcmp(domeEm, s->oldDomeEmission, 12) != 0))
    s->dirty |= DIRTY_FRAMEBUFFER;
  s->haveOldParams = true; s->oldCamera = params->camera; s->oldSettings = rs; memcpy(s->oldClear, clear, sizeof(clear));
  s->oldRowBegin = rowBegin; s->oldRowEnd = rowEnd; s->oldRowStride = rowStride; s->oldDome = params->domeLight;
  if (domeEm) memcpy(s->oldDomeEmission, domeEm, 12);

  s->stats.bvhBuildMs = 0.0; s->stats.uploadMs = 0.0; s->stats.nodeCount = s->nodeCount; s->stats.triangleCount = s->triCount;
  if (syncSceneGeometry(s) != GI_C_OK) return GI_C_ERROR;
  if (s->dirty & DIRTY_LIGHTS) { if (uploadLights(s) != GI_C_OK) return GI_C_ERROR; s->dirty &= ~DIRTY_LIGHTS; s->dirty |= DIRTY_FRAMEBUFFER; }
  if (!rs.progressiveAccumulation) s->dirty |= DIRTY_FRAMEBUFFER;
  // --- one device, or the rows dealt to all of them
  // Multi-device: a whole-frame render (the caller does not shard rows itself) with at least as many rows as devices.  ClockCycles is normalised to the
  // frame maximum on the host (Gi.cpp:327-343), a cross-device reduction nobody needs fast: such renders stay on the primary device.
  bool wantsClock = false;
  for (uint32_t i = 0; i < params->aovBindingCount; i++) if (params->aovBindings[i].aovId == GI_C_AOV_CLOCK_CYCLES) wantsClock = true;
  uint32_t nDev = std::min<uint32_t>(sceneDeviceCount(s), (uint32_t)s->replicas.size() + 1u);
  if (rowStride != 1u || rowBegin != 0u || rowEnd != height || wantsClock || height < nDev) nDev = 1u;
  // every device blends progressive frames against ITS OWN copy of the render buffers: when the device count of this call differs from the previous call's
  // (a ClockCycles binding came or went, the DEVICES option changed) the copies disagree, so the accumulation restarts
  if (s->lastRenderDevices != nDev) s->dirty |= DIRTY_FRAMEBUFFER;
  s->lastRenderDevices = nDev;
  if (s->dirty & DIRTY_FRAMEBUFFER) { s->sampleOffset = 0; s->dirty &= ~DIRTY_FRAMEBUFFER; }


  RenderJob job{params, colorBinding, width, height, rowBegin, rowEnd, rowStride, tileRows, {0}, true};
  memcpy(job.clear, clear, sizeof(job.clear));
  // AOVs the colour pass fills along whole paths (NEE, Bounces, ClockCycles) and unknown ids start from their clear value: host copy filled once, here
  for (uint32_t i = 0; i < params->aovBindingCount; i++) {
    const GiCAovBinding& b = params->aovBindings[i];
    GiCRenderBuffer* rb = b.renderBuffer;
    const bool pathAov = b.aovId == GI_C_AOV_NEE || b.aovId == GI_C_AOV_BOUNCES || b.aovId == GI_C_AOV_CLOCK_CYCLES || b.aovId < 0 || b.aovId >= GI_C_AOV_COUNT;
    if (b.aovId == GI_C_AOV_COLOR || !pathAov) continue;
    const size_t n = (size_t)rb->width * rb->height;
    for (size_t k = 0; k < n; k++) memcpy((uint8_t*)rb->hostMem + k * rb->stride, b.clearValue, rb->stride);
  }
  int rc = GI_C_OK;
  if (nDev == 1u) {
    rc = renderOnDevice(s, *s, job);
  } else {
    rc = renderOnDevices(s, nDev, job);
  }
  if (rc != GI_C_OK) return rc;
  s->sampleOffset += rs.spp; // Gi.cpp:2515
  return GI_C_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// giCTraceRays: closest hits through the device traversal kernel (parity tests of the BVH8 path)
// ---------------------------------------------------------------------------------------------------------------
static int giCTraceRaysImpl(GiCScene* s, uint32_t count, const float* origins, const float* dirs, float tMin, float tMax, float* outTUV, int32_t* outInstPrim);
extern "C" int giCTraceRays(GiCScene* s, uint32_t count, const float* origins, const float* dirs, float tMin, float tMax, float* outTUV, int32_t* outInstPrim)
{
  try { return giCTraceRaysImpl(s, count, origins, dirs, tMin, tMax, outTUV, outInstPrim); }
  catch (const std::exception& e) { setError(std::string("giCTraceRays: ") + e.what()); return -1; }
}
static int giCTraceRaysImpl(GiCScene* s, uint32_t count, const float* origins, const float* dirs, float tMin, float tMax, float* outTUV, int32_t* outInstPrim)
{
  if (!g_ctx.initialized || !s || (count && (!origins || !dirs || !outTUV || !outInstPrim))) { setError("giCTraceRays: bad arguments"); return -1; }
  if (count == 0) return 0;
  std::lock_guard<std::mutex> guard(s->mutex);
  hipStream_t st = g_ctx.stream;
  if (syncSceneGeometry(s) != GI_C_OK) return -1;
  // the render loop's grids: k_trace_dyn (scenes beyond LDS) is persistent per wave and wants every resident wave slot filled (8 blocks per CU offered)
  const bool inLds = s->nodeCount <= 384u && s->triCount <= 128u;
  const uint32_t blocks = std::min<uint32_t>((count + 255u) / 256u, (uint32_t)g_ctx.cuCount * (inLds ? 3u : 8u));
  if (ensurePathState(s, count, blocks, blocks) != GI_C_OK) return -1;
  // ray records go straight into the TRACE_A queue (segment k holds rays [k*per, (k+1)*per))
  const size_t qn = (size_t)s->queueCap * NSHARD;
  std::vector<uint32_t> qslot(qn, 0u); std::vector<F4> qa(qn), qb(qn);
  Counters c{};
  const uint32_t per = (count + NSHARD - 1u) / NSHARD;
  for (uint32_t k = 0; k < NSHARD; k++) { uint32_t lo = k * per; c.count[Q_TRACE_A][k].v = lo < count ? std::min(per, count - lo) : 0u; }
  for (uint32_t i = 0; i < count; i++) {
    size_t r = (size_t)(i / per) * s->queueCap + (i % per);
    qslot[r] = i;
    qa[r] = F4{origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], tMin};
    qb[r] = F4{dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], tMax};
  }
  if (hipMemcpyAsync(s->qSlot[Q_TRACE_A].ptr, qslot.data(), qn * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(s->qA[Q_TRACE_A].ptr, qa.data(), qn * sizeof(F4), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(s->qB[Q_TRACE_A].ptr, qb.data(), qn * sizeof(F4), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(s->dCounters.ptr, &c, sizeof(c), hipMemcpyHostToDevice, st) != hipSuccess) { setError("giCTraceRays: upload failed"); return -1; }
  PathState ps{s->slots.ptr, nullptr, 0u, nullptr, 0u, nullptr};
  launchTrace(st, blocks, false, false, makeView(s), ps, makeQueueSet(s), s->dCounters.ptr, Q_TRACE_A, Q_REGEN_B, traceDynRefill(s), blocks, FrameUniforms{}, nullptr); // (no TRACE_FRESH entries: the uniforms are not read)
  std::vector<TriRec> tris(s->triCount);
  if (hipMemcpyAsync(&c, s->dCounters.ptr, sizeof(c), hipMemcpyDeviceToHost, st) != hipSuccess ||
      (s->triCount && hipMemcpyAsync(tris.data(), s->dTris.ptr, s->triCount * sizeof(TriRec), hipMemcpyDeviceToHost, st) != hipSuccess) ||
      hipStreamSynchronize(st) != hipSuccess) { setError("giCTraceRays: readback failed"); return -1; }
  std::vector<F4> hit(count, F4{tMax, 0.0f, 0.0f, 0.0f});
  for (uint32_t i = 0; i < count; i++) { uint32_t m = 0xffffffffu; memcpy(&hit[i].w, &m, 4); }
  // results stay in the ray records (a = t, u, v, triangle | class << 28); the class queues hold their indices
  std::vector<uint32_t> hitIdx(qn);
  if (hipMemcpy(qa.data(), s->qA[Q_TRACE_A].ptr, qn * sizeof(F4), hipMemcpyDeviceToHost) != hipSuccess) { setError("giCTraceRays: readback failed"); return -1; }
  for (uint32_t klass = 0; klass < MAT_CLASS_COUNT; klass++) {
    if (hipMemcpy(hitIdx.data(), s->qSlot[Q_HIT + klass].ptr, qn * 4, hipMemcpyDeviceToHost) != hipSuccess) { setError("giCTraceRays: readback failed"); return -1; }
    for (uint32_t k = 0; k < NSHARD; k++)
      for (uint32_t j = 0; j < c.count[Q_HIT + klass][k].v; j++) {
        const uint32_t ri = hitIdx[(size_t)k * s->queueCap + j] & 0x3fffffffu;
        if (ri < qn && qslot[ri] < count) { F4 h = qa[ri]; uint32_t w; memcpy(&w, &h.w, 4); w &= 0x0fffffffu; memcpy(&h.w, &w, 4); hit[qslot[ri]] = h; }
      }
  }
  int hits = 0;
  for (uint32_t i = 0; i < count; i++) {
    uint32_t tri; memcpy(&tri, &hit[i].w, 4);
    outTUV[3 * i] = hit[i].x; outTUV[3 * i + 1] = hit[i].y; outTUV[3 * i + 2] = hit[i].z;
    if (tri == 0xffffffffu || tri >= s->triCount) { outInstPrim[2 * i] = -1; outInstPrim[2 * i + 1] = -1; }
    else { outInstPrim[2 * i] = (int32_t)tris[tri].instance; outInstPrim[2 * i + 1] = (int32_t)tris[tri].prim; hits++; }
  }
  return hits;
}

// ---------------------------------------------------------------------------------------------------------------
// giCDebugValidateBvh: host-only check of the builder's conservativeness contract
// ---------------------------------------------------------------------------------------------------------------
static int validateTree(const std::vector<Node8>& nodes, const std::vector<TriRec>& trisArr, uint32_t triCount)
{
  int violations = 0;
  std::vector<uint8_t> seen(triCount, 0);
  struct Item { uint32_t node; float lo[3], hi[3]; };
  std::vector<Item> stack;
  Item root; root.node = 0; for (int a = 0; a < 3; a++) { root.lo[a] = -3.0e38f; root.hi[a] = 3.0e38f; }
  stack.push_back(root);
  while (!stack.empty()) {
    Item it = stack.back(); stack.pop_back();
    if (it.node >= nodes.size()) { violations++; continue; }
    const Node8& n = nodes[it.node];
    uint32_t rel = 0;
    for (int s = 0; s < 8; s++) {
      uint8_t meta = n.meta[s];
      if (meta == 0) { if (n.imask & (1u << s)) violations++; continue; }
      float lo[3], hi[3];
      for (int a = 0; a < 3; a++) {
        uint32_t eb = (uint32_t)n.e[a] << 23; float scale; memcpy(&scale, &eb, 4);
        lo[a] = n.p[a] + (float)n.qlo[a][s] * scale; hi[a] = n.p[a] + (float)n.qhi[a][s] * scale;
      }
      bool inner = (n.imask >> s) & 1u;
      if (inner) {
        if ((meta >> 5) != 1u || (meta & 31u) != 24u + (uint32_t)s) violations++;
        Item c; c.node = n.childBase + rel; rel++;
        for (int a = 0; a < 3; a++) { c.lo[a] = lo[a]; c.hi[a] = hi[a]; }
        // every triangle below must also be inside all ancestors: intersect the constraint boxes
        for (int a = 0; a < 3; a++) { c.lo[a] = std::max(c.lo[a], it.lo[a]); c.hi[a] = std::min(c.hi[a], it.hi[a]); }
        stack.push_back(c);
      } else {
        uint32_t unary = meta >> 5, off = meta & 31u, cnt = unary == 1u ? 1u : unary == 3u ? 2u : unary == 7u ? 3u : 0u;
        if (cnt == 0u || off + cnt > 24u) { violations++; continue; }
        for (uint32_t k = 0; k < cnt; k++) {
          uint32_t ti = n.triBase + off + k;
          if (ti >= trisArr.size()) { violations++; continue; }
          const TriRec& t = trisArr[ti];
          if (t.origId >= triCount || seen[t.origId]) { violations++; continue; }
          seen[t.origId] = 1;
          for (int v = 0; v < 3; v++)
            for (int a = 0; a < 3; a++) {
              float x = t.v0[a] + (v == 1 ? t.e1[a] : v == 2 ? t.e2[a] : 0.0f);
              float blo = std::max(lo[a], it.lo[a]), bhi = std::min(hi[a], it.hi[a]);
              if (x < blo || x > bhi) violations++;
            }
        }
      }
    }
  }
  // every active triangle sits in exactly one leaf; an inactive one (bvh8.h: a vertex that is not finite or beyond 1e18) in none
  std::vector<uint8_t> inactive(triCount, 0);
  for (const TriRec& t : trisArr) {
    if (t.origId >= triCount) { violations++; continue; }
    for (int a = 0; a < 3; a++) {
      const float x0 = t.v0[a], x1 = t.v0[a] + t.e1[a], x2 = t.v0[a] + t.e2[a];
      if (!(std::fabs(x0) <= 1.0e18f) || !(std::fabs(x1) <= 1.0e18f) || !(std::fabs(x2) <= 1.0e18f)) inactive[t.origId] = 1;
    }
  }
  for (uint32_t i = 0; i < triCount; i++) if ((seen[i] != 0) == (inactive[i] != 0)) violations++;
  return violations;
}

extern "C" int giCDebugValidateBvh(const float* triVerts, uint32_t triCount, uint32_t* outNodeCount, uint32_t* outMaxDepth)
{
  if (triCount && !triVerts) return -1;
  std::vector<TriRec> tris(triCount);
  for (uint32_t i = 0; i < triCount; i++) {
    const float* p = triVerts + 9 * (size_t)i;
    for (int a = 0; a < 3; a++) { tris[i].v0[a] = p[a]; tris[i].e1[a] = p[3 + a] - p[a]; tris[i].e2[a] = p[6 + a] - p[a]; }
    tris[i].instance = 0; tris[i].prim = i; tris[i].origId = i;
  }
  Bvh8 bvh; buildBvh8(tris, bvh);
  if (outNodeCount) *outNodeCount = (uint32_t)bvh.nodes.size();
  if (outMaxDepth) *outMaxDepth = bvh.maxDepth;
  return validateTree(bvh.nodes, bvh.tris, triCount);
}

// The same check for the PARTITIONED layout of incremental updates: the triangles are cut into `partCount` consecutive ranges, every range gets its own
// subtree in its own node range, and a top tree over the subtree roots (buildTopBvh8) joins them.  Returns the violations of the assembled tree.
extern "C" int giCDebugValidatePartitionedBvh(const float* triVerts, uint32_t triCount, uint32_t partCount, uint32_t* outNodeCount, uint32_t* outMaxDepth)
{
  if (!triVerts || triCount == 0 || partCount == 0 || partCount > triCount) return -1;
  const uint32_t topCap = partCount * 2u + 16u;
  std::vector<Node8> nodes(topCap, Node8{}); std::vector<TriRec> trisAll(triCount);
  std::vector<float> boxes(6 * (size_t)partCount); std::vector<Node8> roots(partCount);
  uint32_t subDepth = 0;
  for (uint32_t pi = 0; pi < partCount; pi++) {
    const uint32_t first = (uint32_t)((uint64_t)triCount * pi / partCount), end = (uint32_t)((uint64_t)triCount * (pi + 1) / partCount);
    std::vector<TriRec> tris(end - first);
    for (uint32_t i = first; i < end; i++) {
      const float* p = triVerts + 9 * (size_t)i; TriRec& t = tris[i - first];
      for (int a = 0; a < 3; a++) { t.v0[a] = p[a]; t.e1[a] = p[3 + a] - p[a]; t.e2[a] = p[6 + a] - p[a]; }
      t.instance = pi; t.prim = i - first; t.origId = i - first;
    }
    Bvh8 b; buildBvh8(tris, b);
    const uint32_t off = (uint32_t)nodes.size();
    for (Node8 n : b.nodes) { n.childBase += off; n.triBase += first; nodes.push_back(n); }
    for (uint32_t k = 0; k < end - first; k++) { TriRec t = b.tris[k]; t.origId += first; trisAll[first + k] = t; }
    roots[pi] = nodes[off]; nodeBounds(nodes[off], &boxes[6 * (size_t)pi]);
    subDepth = std::max(subDepth, b.maxDepth);
  }
  Bvh8 top; buildTopBvh8(boxes.data(), partCount, roots.data(), top);
  if (top.nodes.size() > topCap) return -2;
  std::copy(top.nodes.begin(), top.nodes.end(), nodes.begin());
  if (outNodeCount) *outNodeCount = (uint32_t)nodes.size();
  if (outMaxDepth) *outMaxDepth = top.maxDepth + subDepth - 1u;
  return validateTree(nodes, trisAll, triCount);
}

// giCDebugShadeClass: which k_shade variant an (untextured) material's hits are binned for -- host only
extern "C" int giCDebugShadeClass(const GiCMaterialDesc* desc)
{
  if (!desc) return -1;
  MaterialRec m{}; m.klass = desc->klass; m.flags = desc->flags & ~(MAT_FLAG_TEXTURED | MAT_FLAG_OPACITY_TEX); memcpy(m.p, desc->p, sizeof(m.p));
  deriveMaterialConstants(m);
  return (int)shadeClassOf(m);
}

// ---------------------------------------------------------------------------------------------------------------
// giCDebugEvalBsdf: closed-form BSDF sample/evaluate on the device for explicit shading frames
// ---------------------------------------------------------------------------------------------------------------
extern "C" int giCDebugEvalBsdf(const GiCMaterialDesc* desc, uint32_t count, const float* in, float* out)
{
  if (!g_ctx.initialized || !desc || (count && (!in || !out))) { setError("giCDebugEvalBsdf: bad arguments"); return GI_C_ERROR; }
  if (count == 0) return GI_C_OK;
  MaterialRec m{}; m.klass = desc->klass; m.flags = desc->flags & ~(MAT_FLAG_TEXTURED | MAT_FLAG_OPACITY_TEX); memcpy(m.p, desc->p, sizeof(m.p));
  deriveMaterialConstants(m);
  const uint32_t shadeClass = shadeClassOf(m); // the variant the render would shade this material's hits with (GATLING_OPTIONS=shade_variants=0: always the full closed form)
  MaterialRec* dm = nullptr; float* din = nullptr; float* dout = nullptr;
  hipStream_t st = g_ctx.stream;
  int rc = GI_C_ERROR;
  if (hipMalloc((void**)&dm, sizeof(m)) == hipSuccess && hipMalloc((void**)&din, (size_t)count * 22 * 4) == hipSuccess &&
      hipMalloc((void**)&dout, (size_t)count * 15 * 4) == hipSuccess &&
      hipMemcpyAsync(dm, &m, sizeof(m), hipMemcpyHostToDevice, st) == hipSuccess &&
      hipMemcpyAsync(din, in, (size_t)count * 22 * 4, hipMemcpyHostToDevice, st) == hipSuccess) {
    launchDebugBsdf(st, dm, shadeClass, count, din, dout);
    if (hipMemcpyAsync(out, dout, (size_t)count * 15 * 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) rc = GI_C_OK;
  }
  if (rc != GI_C_OK) setError("giCDebugEvalBsdf: HIP failure");
  if (dm) (void)hipFree(dm); if (din) (void)hipFree(din); if (dout) (void)hipFree(dout);
  return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// giCDebugTexRuntime: the MDL renderer runtime's remaining texture entry points (tex_texel_float4_2d, tex_resolution_2d, tex_lookup_float4_3d, tex_texel_float4_3d)
// on the device, for explicit queries
// ---------------------------------------------------------------------------------------------------------------
extern "C" int giCDebugTexRuntime(const float* rgba, uint32_t width, uint32_t height, uint32_t depth, uint32_t count, const float* queries, float* out)
{
  if (!g_ctx.initialized || !rgba || !width || !height || !depth || (count && (!queries || !out))) { setError("giCDebugTexRuntime: bad arguments"); return GI_C_ERROR; }
  if (count == 0) return GI_C_OK;
  const size_t texFloats = (size_t)width * height * depth * 4;
  float* dt = nullptr; float* dq = nullptr; float* dout = nullptr;
  hipStream_t st = g_ctx.stream;
  int rc = GI_C_ERROR;
  if (hipMalloc((void**)&dt, texFloats * 4) == hipSuccess && hipMalloc((void**)&dq, (size_t)count * 32) == hipSuccess && hipMalloc((void**)&dout, (size_t)count * 16) == hipSuccess &&
      hipMemcpyAsync(dt, rgba, texFloats * 4, hipMemcpyHostToDevice, st) == hipSuccess && hipMemcpyAsync(dq, queries, (size_t)count * 32, hipMemcpyHostToDevice, st) == hipSuccess) {
    launchDebugTex(st, dt, width, height, depth, count, dq, dout);
    if (hipMemcpyAsync(out, dout, (size_t)count * 16, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) rc = GI_C_OK;
  }
  if (rc != GI_C_OK) setError("giCDebugTexRuntime: HIP failure");
  if (dt) (void)hipFree(dt); if (dq) (void)hipFree(dq); if (dout) (void)hipFree(dout);
  return rc;
}
