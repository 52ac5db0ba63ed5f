// gi_host.h -- shared declarations of the host side (gi_c.cpp, gi_scene.cpp, gi_textures.cpp, gi_lights.cpp, gi_build.cpp, gi_render.cpp, gi_debug.cpp): host
// side of the MI355X-native gi core: scene containers, dirty flags, host packing, BVH build,
// uploads, the wavefront bounce loop, render buffers.  Implements include/gi_c.h.
//
// Restates the host logic of /root/reference/src/gi/impl/Gi.cpp behind the same API shape:
//   giCreateMesh/giSetMesh* (:620-782)          -> MeshData + dirty flags
//   _giBuildGeometryStructures/_giCreateBvh (:784-1315) -> flatten instances, pack FVertex, build BVH8, upload
//   giRender (:1989-2524)                        -> dirty handling, uniforms (:2373-2426), bounce loop, D2H
//   light setters (:2573-2976)                   -> CPU mirrors of the 48-byte device structs, dense stores
//   render buffers (:2978-3006)
// GPU plumbing (src/cgpu, src/ggpu in the reference) is the HIP runtime: hipMalloc / hipMemcpyAsync / streams.

#pragma once

// the C ABI keeps default visibility; everything else in the host translation units is built with -fvisibility=hidden (gatling_amd/build.py)
#pragma GCC visibility push(default)
#include "../../include/gi_c.h"
#pragma GCC visibility pop


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

constexpr bool WORK_ORDER_PIXEL_MAJOR_DEFAULT = true;  // (GATLING_OPTIONS work_order) FLAG_PIXEL_MAJOR, gi_queues.h work_item

extern thread_local std::string t_lastError;
void setError(const std::string& e);

#define HIP_TRY(expr)                                                                                      \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) { setError(std::string(#expr) + ": " + hipGetErrorString(_e)); return GI_C_ERROR; } \
  } while (0)

// One entry per HIP device the library renders on (giCInitializeDevices / $GATLING_DEVICES; giCInitialize: one).  devs[0] is the PRIMARY device:
// render buffers, textures and every single-device entry point live there; the others hold replicas of the scene and render row shares.
struct DevCtx { int device = 0; int cuCount = 256; hipStream_t stream = nullptr; hipStream_t stream2 = nullptr;
    /* the shadow launches of two-stream batches (renderOnDevice "two streams") */
                /* 1: the primary device and this one address each other's memory (peer access enabled both ways, or the same physical device); 0: no peer
                   access -- the device's row shares travel through pinned host memory; -1: hipDeviceCanAccessPeer / EnablePeerAccess failed with an error */
                int peer = 1; };
// One host thread per further device, created with the first multi-device render and kept until giCTerminate (a frame's share is handed to it as a job; until
// r04 every frame created and joined its own std::threads).  The thread binds its HIP device once.
struct DeviceWorker {
  std::thread th; std::mutex m; std::condition_variable cv;
  std::function<void()> job; bool busy = false, stop = false;
  void start() { th = std::thread([this] { std::unique_lock<std::mutex> lk(m);
      for (;;) { cv.wait(lk, [this] { return busy || stop; }); if (stop) return; lk.unlock(); job(); lk.lock(); busy = false; cv.notify_all(); } }); }
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
};
extern Context g_ctx;

double nowMs();

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
    if (const int rc = alloc(v.size())) {
      if (rc == GI_C_OUT_OF_MEMORY_INTERNAL) setError("hipMalloc: out of device memory (" + std::to_string(v.size() * sizeof(T)) + " bytes)");
      return GI_C_ERROR;
    }
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return GI_C_OK;
  }
  void release() { if (ptr) { (void)hipFree(ptr); ptr = nullptr; count = 0; } }
};


// host-side arithmetic shared by the scene build, the light setters and the debug hooks (gi_c.cpp)
uint16_t f32ToF16(float f);
float f16ToF32(uint16_t h);
uint32_t packHalf2x16(float a, float b);
uint32_t encodeDirection(const float* vin);
void decodeDirection(uint32_t e, float out[3]);
void turboColormap(float x, float* rgb);
float cutoutOpacity(const MaterialRec& m);
void deriveMaterialConstants(MaterialRec& m);

// ---------------------------------------------------------------------------------------------------------------
// handle types
// ---------------------------------------------------------------------------------------------------------------
// DIRTY_XFORM: only transforms of meshes that are part of the built scene changed -- the incremental path (updateTransforms) handles it unless a full
// rebuild is due anyway.  The reference keeps each mesh's BLAS and rebuilds the TLAS (Gi.cpp:1180-1202).
enum DirtyFlags : uint32_t { DIRTY_BVH = 1u, DIRTY_FRAMEBUFFER = 2u, DIRTY_LIGHTS = 4u, DIRTY_MATERIALS = 8u, DIRTY_ALL = 0xfu, DIRTY_XFORM = 16u };

struct GiCTexture { GiCScene* scene; uint32_t width, height; std::vector<float> rgba; std::string cacheKey; uint32_t refs = 1; };
struct GiCPrimvar { std::string name; int32_t type, interpolation; std::vector<float> data; };
struct GiCMaterial { GiCScene* scene; std::string name; GiCMaterialDesc desc; GiCTextureBinding tex[GI_C_TEX_SLOT_COUNT] = {};
    std::string primvarInput[GI_C_TEX_SLOT_COUNT];
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
struct GiCDomeLight { GiCScene* scene; std::string filePath; GiCTexture* texture = nullptr; bool ownsTexture = false; float rotation[4] = {0, 0, 0, 1};
    float baseEmission[3] = {1, 1, 1}; float diffuse = 1.0f, specular = 1.0f; };

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
      DeviceBuffer<F4> dTriGeomNormal;
  DeviceBuffer<SphereLightRec> dSphere; DeviceBuffer<DistantLightRec> dDistant; DeviceBuffer<RectLightRec> dRect; DeviceBuffer<DiskLightRec> dDisk;
  DeviceBuffer<LightFrame> dRectFrames, dDiskFrames; // decoded tangents + normal per rect / disk light (uploadLights)
  DeviceBuffer<Node8> dTlasNodes, dBlasNodes; DeviceBuffer<uint32_t> dTlasItems, dFlatOfOrig; DeviceBuffer<BlasTri> dBlasTris; DeviceBuffer<InstTrav> dInstTrav;
  // path state
  DeviceBuffer<Slot> slots;
  DeviceBuffer<float> media; // per-slot medium stack + walkSegmentPdf (mediumStackSize > 0)
  DeviceBuffer<F4> scratchColor; DeviceBuffer<unsigned long long> neeKey; // NEE / Bounces AOVs bound without / with the colour AOV
  DeviceBuffer<uint32_t> pathSegments;                                    // ClockCycles AOV (cost proxy)
  // per-sample colours of the current batch (rgb, -): [pixel][sample] under the pixel-major work order of the stage kernels, [sample][pixel] otherwise
  // (gi_queues.h sample_record)
  DeviceBuffer<F4> sampleBuf;
  DeviceBuffer<F4> accum;        // per-pixel running sum across batches
  DeviceBuffer<uint32_t> qSlot[Q_COUNT]; // NSHARD segments of queueCap records each
  DeviceBuffer<F4> qA[Q_COUNT], qB[Q_COUNT], qC[Q_COUNT];
  DeviceBuffer<FreshRec> qFresh[2]; // beside TRACE_A / TRACE_B: (rng, work item) of camera rays whose Slot is written only when they hit (FLAG_DEFER_SLOT)
  uint32_t queueCap = 0;
  DeviceBuffer<Counters> dCounters;
  Counters* hCounters = nullptr; // pinned
  uint64_t memTotalMb = 0;       // the device's memory (hipMemGetInfo, asked once): sizes the default sample-buffer budget
  // drain test of the bounce loop: iteration it reads the queue sizes of iteration it - POLL_LAG (giCRenderImpl)
  static constexpr uint32_t POLL_RING = 4, POLL_LAG = 2;
  PaddedCounter* hPoll = nullptr; hipEvent_t pollEvent[POLL_RING] = {}; // pinned ring of queue-size snapshots + their completion events
  // two-stream batches: k_shade(i) done (the second stream's shadow launch waits for it) / shadow launch (i) done (k_raygen(i + 1) waits for it)
  hipEvent_t evShade = nullptr, evShadow = nullptr;
  GiCRenderStats stats{};
  std::vector<hipEvent_t> eventPool;
  void releaseAll();
};

// The scene as host arrays (built once per scene change, kept for incremental transform updates) ...
struct TwoLevelHost { std::vector<Node8> tlasNodes, blasNodes; std::vector<uint32_t> tlasItems; std::vector<BlasTri> blasTris; std::vector<InstTrav> instTrav;
    };
struct MeshBuild { const GiCMesh* m; uint32_t vertexOffset, matFlags, instFirst, instCount, triFirst; uint32_t meshIdx; std::vector<int32_t> faceIdAov;
    uint32_t shadeBase = 0; };
// One flattened mesh instance of a PARTITIONED scene (after the first transform edit): its own subtree in its own node range, its triangles in its own
// (scene-order) range, joined by a top tree over the subtree roots (bvh8.h buildTopBvh8).  Moving it rebuilds these ranges and the top tree only.
struct InstPart { uint32_t meshBuild, instInMesh; uint32_t triFirst, nf; uint32_t nodeOff, nodeCount, nodeCap, depth; float box[6]; };
struct SceneHost {
  std::vector<FVertex> verts; std::vector<InstanceRec> instances; std::vector<MaterialRec> mats; std::vector<MeshRec> meshRecs; std::vector<float> sceneData;
  Bvh8 bvh; std::vector<int32_t> triFaceId; std::vector<uint32_t> flatOfOrig; TwoLevelHost two;
  std::vector<MeshBuild> meshBuilds;
  std::vector<F4> triGeomNormal; // LDS-resident scenes: per flattened triangle (gi_build.cpp)
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
  // sphere / distant / rect / disk lights the device arrays hold (uploadLights: the stores' records minus the unusable ones)
  uint32_t lightCounts[4] = {0, 0, 0, 0};
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
  // the same per SHADE class (gi_types.h: the HIT queues of the wavefront pipeline; classMask / classTextured pick the fused kernels)
  uint32_t shadeClassMask = 0, shadeClassTextured = 0;
  std::unique_ptr<SceneHost> host; // the scene as host arrays, kept for incremental transform updates
  std::vector<std::unique_ptr<SceneDevice>> replicas; // devices 1 .. N-1 (created with the first build when the library runs on several devices)
  // options + stats
  bool countTraversal = false, kernelTimers = false;
  uint32_t kernelTimerStride = 1;
  uint64_t optPoolSlots = 0, optSampleBufferMb = 0; // 0 = default
  // -1 = default: LDS-resident scenes run the fused persistent kernel k_path; 1 = k_path_bw (wave-local wavefront) when NEE is off; 2 = k_path; 0 = always the
  // wavefront stage kernels
  int32_t optFusedPath = -1;
  int32_t optTraceDyn = -1; // -1 = default; 0 = block-synchronous k_trace everywhere; N = k_trace_dyn refill threshold
  // Visiting order of shadow walks (k_trace_dyn<any>; any order gives the same image): -1 = not chosen yet -- launches alternate between near-to-far (0) and
  // slot order (1) and the frame's node-visit counts are added up below; once both orders have walked enough rays the cheaper one is kept until the tree is
  // rebuilt.
  std::atomic<int32_t> shadowOrder{-1};
      /* read by every device worker at the start of its render, written by the primary at the end of its own */ uint64_t shadowOrderRays[2] = {0, 0},
      shadowOrderSteps[2] = {0, 0};
  int32_t optDevices = 0;   // 0 = every device the library was initialised on; N = at most N of them
  // devices the previous giCRender used: progressive accumulation blends against each device's own buffer, so a change restarts it
  uint32_t lastRenderDevices = 0;
};


// ---------------------------------------------------------------------------------------------------------------
// functions the translation units share
// ---------------------------------------------------------------------------------------------------------------
// gi_textures.cpp asset reader -> loader hook -> in-library decoders
extern "C++" bool loadImage(const char* path, bool srgbToLinear, bool keepHdr, uint32_t& w, uint32_t& h, std::vector<float>& px);
// gi_lights.cpp
int uploadLights(GiCScene* s);
// gi_build.cpp
uint32_t shadeClassOf(const MaterialRec& m);                 // shade class (k_shade variant) of a derived material record
uint32_t sceneDeviceCount(const GiCScene* s);                // devices a render of this scene may use
SceneDevice& sceneDevice(GiCScene* s, uint32_t slot);
void nodeBounds(const Node8& n, float box[6]);               // dequantised bounds of a node's children (+ an ulp-scale pad)
int syncSceneGeometry(GiCScene* s);                          // brings the device scene up to date with the host-side edits (incremental or full build)
// gi_render.cpp
SceneView makeView(GiCScene* s, SceneDevice& D);
SceneView makeView(GiCScene* s);
uint32_t traceDynRefill(const GiCScene* s);
int ensurePathState(SceneDevice* s, size_t slots, uint32_t gridA, uint32_t gridB);
QueueSet makeQueueSet(SceneDevice* s);

