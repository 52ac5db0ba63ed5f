// gi_render.cpp -- giCRender: dirty handling, uniforms, memory plan, the bounce loop on one / several devices (Gi.cpp:1989-2524)
// (one of the translation units gi_c.cpp was split into in round 6; shared declarations: gi_host.h)
#include "gi_host.h"

bool settingsEqual(const GiCRenderSettings& a, const GiCRenderSettings& b) { return memcmp(&a, &b, sizeof(a)) == 0; }

SceneView makeView(GiCScene* s, SceneDevice& D)
{
  SceneView v{};
  v.textures = D.dTextures.ptr; v.meshes = D.dMeshes.ptr; v.sceneData = D.dSceneData.ptr;
  v.nodes = D.dNodes.ptr; v.tris = D.dTris.ptr; v.instances = D.dInstances.ptr;
  v.verts = D.dVerts.ptr; v.triShade = D.dTriShade.ptr; v.triGeomNormal = D.dTriGeomNormal.ptr; v.shadePacked = s->shadePacked ? 1u : 0u;
      v.materials = D.dMaterials.ptr;
      v.sphereLights = D.dSphere.ptr; v.distantLights = D.dDistant.ptr;
  v.tlasNodes = D.dTlasNodes.ptr; v.tlasItems = D.dTlasItems.ptr; v.blasNodes = D.dBlasNodes.ptr; v.blasTris = D.dBlasTris.ptr; v.instTrav = D.dInstTrav.ptr;
      v.flatOfOrig = D.dFlatOfOrig.ptr; v.twoLevel = s->twoLevel ? 1u : 0u;
  v.rectLights = D.dRect.ptr; v.diskLights = D.dDisk.ptr; v.rectFrames = D.dRectFrames.ptr; v.diskFrames = D.dDiskFrames.ptr; v.triFaceId = D.dTriFaceId.ptr;
      v.nodeCount = s->nodeCount; v.triCount = s->triCount;
      v.bvhDepth = s->bvhDepth; v.hasCutouts = s->hasCutouts ? 1u : 0u;
  return v;
}
SceneView makeView(GiCScene* s) { return makeView(s, *s); }

// k_trace_dyn refill threshold for scenes that do not fit LDS (0 = use the block-synchronous k_trace)
uint32_t traceDynRefill(const GiCScene* s)
{
  uint32_t r = s->optTraceDyn >= 0 ? (uint32_t)s->optTraceDyn : 8u;
  if (optionSet("trace_dyn")) r = (uint32_t)std::max(0L, std::min(64L, optionValue("trace_dyn", 8)));
  if (r && optionValue("trace_dyn_spill8", 0)) r |= TRACE_DYN_SPILL8;
  return r;
}

uint32_t shardCapacity(size_t slots, uint32_t gridA, uint32_t gridB)
{
  // A queue holds at most `slots` records in total (a path sits in one queue at a time), but it is fed by SEVERAL launches before it is consumed -- TRACE[par]
  // by k_raygen and one k_shade per material class, REGEN by k_trace / k_route, every k_shade and k_raygen (maxBounces == 0) -- and every launch starts dealing
  // its blocks at shard 0. A launch of G blocks that appends n records gives one shard at most ceil(G/NSHARD) * ceil(n/(256 G)) * 256 <= n/NSHARD + n/G + 32 G
  // + 256 of them; summed over P producers with sum(n) <= slots this is slots/NSHARD + P * (slots/Gmin + 32 Gmax + 256). block_append also raises
  // Counters::overflow if a shard ever runs past its capacity (giCRender then fails instead of returning a corrupt image). (a producer that appends I records
  // per thread and trip -- k_route: ROUTE_ITEMS, k_raygen: RAYGEN_ITEMS, gi_kernels.h APPEND_ITEMS_MAX -- deals 256 * I records per block and trip: the slack
  // term is 32 * I * G + 256 * I)
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
struct RenderJob { const GiCRenderParams* params; const GiCAovBinding* colorBinding; uint32_t width, height, rowBegin, rowEnd, rowStride, tileRows;
    uint8_t clear[GI_C_MAX_AOV_COMP_SIZE]; bool readback; };

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
    auto norm3 = [](const float* v, float* o) { float inv = 1.0f / sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); o[0] = v[0] * inv; o[1] = v[1] * inv;
        o[2] = v[2] * inv; };
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
    U.sphereCount = s->lightCounts[0]; U.distantCount = s->lightCounts[1]; U.rectCount = s->lightCounts[2]; U.diskCount = s->lightCounts[3]; // (uploadLights)
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
    for (int a = 0; a < 3; a++) { view.domeEmission[a] = dl ? dl->baseEmission[a] : 1.0f; view.background[a] = U.background[a];
        view.cameraPosition[a] = params->camera.position[a]; }
    view.frame = rs.frame;
  }
  if (ensurePathState(&D, 1, 1, 1) != GI_C_OK) return GI_C_ERROR; // counters / pinned mirror exist even for AOV-only renders
  if (colorRb) {
    // --- work decomposition (DESIGN.md "Persistent path pool"): work item = (pixel, sample); the frame is cut into batches of
    // consecutive samples whose per-sample colour buffer fits the budget; a pool of `slots` paths is kept full from a running
    // work counter until the batch's items run out.
    auto envU64 = [](const char* key, uint64_t def) { return optionSet(key) ? (uint64_t)optionValue(key, 0) : def; };
    // Memory plan (r04). The per-sample colour buffer wants to hold the whole frame's samples (every batch ends in a drain / a kernel tail: C2's 34 GB for 1024
    // spp at 1080p in one batch 213.4 ms per step, in four 215.5) and scenes beyond LDS want a 64 Mi-slot pool (17 GB with its queues) -- on an empty 288 GB
    // device. A Hydra plugin shares the device with other scenes, other processes and the host application, so the plan starts from what is FREE now (plus what
    // this scene already holds in these buffers, which is reused), and an allocation that still fails (someone else was faster) is answered with a smaller plan
    // -- more batches first, then a smaller pool -- never with a failed render while a workable plan exists. Results do not depend on the plan
    // (test_pool_and_batch_invariance).
    size_t memFree = 0, memTotal = 0; (void)hipMemGetInfo(&memFree, &memTotal);
    // tests: plan as if this much were free (a planner overtaken by another allocation: the fallback below must recover)
    if (optionSet("assume_free_mb")) memFree = (size_t)optionValue("assume_free_mb", 0) << 20;
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
    // tail (measured on C3 at spp 64 / 256: 4 Mi slots 630, 16 Mi 790, 32 Mi 831 / 810, 64 Mi - / 868 Msamples/s); LDS-resident scenes have uniform, short
    // rays.
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
    // HIT-queue entries (and giCTraceRays) hold a TRACE-queue RECORD index in 30 bits (HIT_INDEX_MASK); records run up to shardCapacity * NSHARD, which exceeds
    // the slot count by the shards' slack -- a pinned pool near 2^30 would push indices past the mask and k_shade would gather the wrong record
    while (!fused && (uint64_t)shardCapacity(slots, wideBlocks, traceBlocks) * NSHARD > 0x3fffffffull /* HIT_INDEX_MASK, gi_queues.h */) { slots -= slots / 8;
        sizeGrids(); }
    const uint32_t mediaStride = rs.mediumStackSize ? rs.mediumStackSize * MEDIUM_FLOATS + 4u : 0u;
    // what a plan costs: the slot pool with its queues (per slot: the Slot, the medium stack, and a share of every queue's records) and the sample buffer
    auto planBytes = [&](size_t nSlots, uint64_t nBatch) -> uint64_t {
      const uint64_t cap = shardCapacity(nSlots, wideBlocks, traceBlocks);
      const uint64_t perQueueEntry = 4ull * Q_COUNT + 32ull * (2 + 1) + 16ull + 8ull * 2;
      return (fused ? 0ull : (uint64_t)nSlots * (sizeof(Slot) + 4ull * mediaStride) + cap * NSHARD * perQueueEntry) + (uint64_t)pixels * nBatch * 16ull
             + (uint64_t)pixels * 16ull;
    };
    // the caller's sizes are taken as given
    const bool pinnedPlan = optionSet("pool_slots") || s->optPoolSlots || optionSet("sample_buffer_mb") || s->optSampleBufferMb;
    auto shrink = [&]() -> bool { // the next smaller plan: halve the sample buffer down to 64 MiB (more batches), then the pool down to 64 Ki slots
      if (batchSamples > 1 && (uint64_t)pixels * batchSamples * 16ull > (64ull << 20)) { batchSamples = std::max<uint64_t>(1, batchSamples / 2);
          if (!fused) slots = (size_t)std::min<uint64_t>(slots, (uint64_t)pixels * batchSamples); return true; }
      if (!fused && slots > (64u << 10)) { slots /= 2; return true; }
      return false;
    };
    // (leave 512 MiB, or half of a tiny remainder)
    if (!pinnedPlan) while (planBytes(slots, batchSamples) > (availMb << 20) - std::min<uint64_t>(availMb << 19, 512ull << 20) && shrink()) sizeGrids();
    for (int attempt = 0;; attempt++) {
      int rc = fused ? GI_C_OK : ensurePathState(&D, slots, wideBlocks, traceBlocks);
      if (rc == GI_C_OK) rc = D.sampleBuf.alloc(pixels * batchSamples);
      if (rc == GI_C_OK) rc = D.accum.alloc(pixels);
      if (rc == GI_C_OK && mediaStride) rc = D.media.alloc(slots * mediaStride);
      // (a smaller plan fitted: the "out of memory" of the larger ones is not this render's error)
      if (rc == GI_C_OK) { if (attempt > 0) t_lastError.clear(); break; }
      if (rc != GI_C_OUT_OF_MEMORY_INTERNAL) return GI_C_ERROR;
      // out of memory: drop what this scene holds in the resizable buffers (a half-grown plan must not stand in the way of the smaller one) and try the next
      // plan
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
    // Deferred Slot initialisation (r04): k_raygen hands a camera ray its (rng, work item) beside the ray record instead of writing the path's 64-byte Slot;
    // the slot is written where the first segment hits (k_route / k_trace) and a camera ray that leaves the scene retires there without ever touching one. The
    // debug AOVs that follow whole paths read the slot when a sample retires (NEE / Bounces / ClockCycles): renders that bind them keep the eager form.
    if (!fused && optionValue("defer_slot", 1) != 0 && !ps.neeKey && !ps.bouncesAov && !ps.pathSegments) U.flags |= FLAG_DEFER_SLOT;
    QueueSet qs = makeQueueSet(&D);
    F4* colorOut = reinterpret_cast<F4*>(rbMem(colorRb, D.slot));
    const bool nee = rs.nextEventEstimation != 0;
    const uint32_t dynRefill = traceDynRefill(s);
    // (GATLING_OPTIONS=shadow_order=0|1 pins it)
    const int32_t shadowOrderNow = optionSet("shadow_order") ? (int32_t)optionValue("shadow_order", -1) : s->shadowOrder.load();
    // Bounds retire (r04n): on the k_trace_dyn path a deferred-slot camera ray that cannot reach the scene's bounds is retired by k_raygen itself (C4: 58 % of
    // the camera rays, C3: ~45 %) -- same sample, same segment count, no ray record, no traversal step, no routing. Not with a dome image / medium stack (a
    // miss needs the slot), not in counting builds (the root visit of such a ray is part of nodes-per-ray), not on the two-level layout (bounds of the TLAS
    // root: not kept).
    {
      SceneView v0 = view; uint32_t ln, lt, ldsBytes; traceLdsLayout(v0, ln, lt, ldsBytes);
      const bool allLds = ln == v0.nodeCount && lt == v0.triCount && v0.triCount > 0u;
      if ((U.flags & FLAG_DEFER_SLOT) && !allLds && dynRefill && !view.twoLevel && view.domeTexture == 0u && rs.mediumStackSize == 0u && !s->countTraversal
          && s->boundsValid &&
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
      if (timers && (curIter % timerStride) == 0u) { (void)hipEventRecord(poolEvent(&D, ev), on); fn(); (void)hipEventRecord(poolEvent(&D, ev + 1), on);
          ev += 2; evKind.push_back(kind); }
      else fn();
    };
    auto timed = [&](int kind, auto&& fn) { timedOn(st, kind, fn); };
    // Two streams (the reference's default frame is ONE sample per pixel, renderDelegate.cpp:93-110). In a batch
    // whose work fits the pool every path starts in iteration 0, so from iteration 1 on k_raygen only FINISHES samples and the closest-hit launch of iteration
    // i + 1 needs nothing from the shadow launch of iteration i -- which k_raygen(i + 1) (it reads the radiance of paths that ended) and k_shade(i + 1) (it
    // goes on adding to it: the float order of rp_main.rgen:397-480) do need. Such batches run
    //     main stream:    Z(i)  [R(0)]  T(i) + route(i)   <wait for Sh(i-1)>   [R(i), i > 0]   S(i)
    //     second stream:                                  <wait for S(i)>  Sh(i)
    // so that Sh(i) runs beside T(i + 1): an iteration lasts max(trace, shadow) + raygen + shade instead of their sum. Per-path arithmetic and per-pixel sample
    // order are untouched (same kernels, same records); what changes is who zeroes which queue counter (gi_queues.h zero_next_counters / zero_closest_counters:
    // Z = k_zero_closest). Not with a dome image (a miss adds the dome's radiance to the Slot in k_route while the previous bounce's shadow launch may still be
    // adding to it: two float additions in an order that would depend on timing) or a medium stack. GATLING_OPTIONS=two_stream=0 switches it off;
    // two_stream_delay=1|2 (tests) holds the main | the second stream back for 0.3 ms per iteration so that the other one runs ahead.
    const bool twoStreamOk = nee && rs.mediumStackSize == 0u && view.domeTexture == 0u && optionValue("two_stream", 1) != 0 && ctx.stream2 != nullptr;
    const long twoStreamDelay = optionValue("two_stream_delay", 0);
    hipStream_t st2 = ctx.stream2;
    if (twoStreamOk && !D.evShade) { HIP_TRY(hipEventCreateWithFlags(&D.evShade, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&D.evShadow, hipEventDisableTiming)); }
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
        // (a wave's last chunk is the launch's tail: 16 claims per wave keep it at ~6 % of a small frame -- C1 5 895 -> 6 360 Msamples/s; C2 does not care, 256
        // ... 2048 measure the same)
        chunk = (uint32_t)std::min<uint64_t>(chunk, std::max<uint64_t>(64u, ((uint64_t)U.workTotal / (waves * 16u)) & ~63ull));
        curIter = totalIters; if (timers) sampledIters++;
        if (timers) { (void)hipEventRecord(poolEvent(&D, ev), st); }
        // which fused kernel: k_path (one path per lane, in registers) unless the wave-local wavefront k_path_bw is asked for (GI_C_SCENE_OPTION_FUSED_PATH = 1
        // / GATLING_OPTIONS=path_bw=1). Measured r03 on C2 (1080p, spp 256, SLP vectorisation off): k_path 55.4 ms per batch, k_path_bw 57.3 -- k_path's 114
        // VGPRs give 4 resident waves per SIMD (3 blocks per CU cost 11 %), k_path_bw's 168 VGPRs and 50 KB of LDS per block give 3; at 128 VGPRs k_path_bw
        // spills 43 registers and falls to 84 ms.
        const int envBw = (int)optionValue("path_bw", -1);
        const bool useBw = !nee && (envBw >= 0 ? envBw != 0 : s->optFusedPath == 1);
        if (useBw) launchPathBw(st, (uint32_t)ctx.cuCount, s->classMask, s->classTextured != 0u, s->countTraversal, chunk, U, view, ps, D.dCounters.ptr,
            D.sampleBuf.ptr);
        else launchPath(st, (uint32_t)ctx.cuCount, s->classMask, s->classTextured != 0u, s->countTraversal, chunk, U, view, ps, D.dCounters.ptr,
            D.sampleBuf.ptr);
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
      // A thin batch (its work fits the pool and is small: the delegate's one sample per
      // pixel and call) launches every kernel of an iteration for a few thousand hits:
      // a further k_shade launch per iteration costs it more (C4: 13 x ~9 us of a 2.7 ms call) than the BASE variant saves, so its BASE hits are binned with
      // class 2 (GATLING_OPTIONS=merge_shade_variants=0 | 1: never | always)
      const long mergeOpt = optionValue("merge_shade_variants", -1);
      const bool thinBatch = rounds == 1 && U.workTotal <= (8u << 20);
      if ((mergeOpt < 0 ? thinBatch : mergeOpt != 0) && (s->shadeClassMask & (1u << SHADE_CLASS_OPBR_BASE))
          && (s->shadeClassMask & 4u)) U.flags |= FLAG_MERGE_SHADE_VARIANTS;
      else U.flags &= ~FLAG_MERGE_SHADE_VARIANTS;
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
        // (the last k_raygen of the batch: the test below ends the loop)
        else if (rounds == 1 && it == (uint64_t)std::max(1u, U.maxBounces)) raygenBehindShadow();
        // A batch whose work fits the pool (a low-spp frame: hdGatling renders ONE sample per pixel and call) starts every path in iteration 0, a path traces
        // at most maxBounces segments, one per iteration (the bounce counter, rp_main.rgen:298-304) -- so k_raygen(maxBounces) has just retired the last
        // samples and nothing is in flight: no need to find that out two empty iterations later through the poll below (10 launches of ~90 in a spp-1 call).
        if (rounds == 1 && it == (uint64_t)std::max(1u, U.maxBounces) && rs.mediumStackSize == 0u) { totalIters++; break; }
        if (it >= rounds) {
          // All work cannot be handed out earlier. From here on every iteration snapshots the queue sizes behind its k_raygen (asynchronous copy into a pinned
          // ring) and tests the snapshot of POLL_LAG iterations ago: the wait is for work the GPU finished long ago -- it still holds the iterations in
          // between, so the stream never runs dry -- and the loop stops at most POLL_LAG empty iterations after the pool drained. (Until r03 the loop
          // synchronised every 16th iteration: C4 ran 15 empty iterations of 0.2 ms each, `tools/exp_iter_log.py`.)
          constexpr uint32_t R = SceneDevice::POLL_RING, LAG = SceneDevice::POLL_LAG;
          constexpr size_t snapshot = (size_t)Q_COUNT * NSHARD;
          HIP_TRY(hipMemcpyAsync(D.hPoll + (it % R) * snapshot, D.dCounters.ptr, sizeof(PaddedCounter) * snapshot, hipMemcpyDeviceToHost, st));
          HIP_TRY(hipEventRecord(D.pollEvent[it % R], st));
          if (it >= rounds + LAG) {
            const uint64_t j = it - LAG;
            HIP_TRY(hipEventSynchronize(D.pollEvent[j % R]));
            const PaddedCounter* snap = D.hPoll + (j % R) * snapshot + (size_t)(Q_TRACE_A + (uint32_t)(j & 1u)) * NSHARD;
            // (FLAG_BOUNDS_RETIRE: a k_raygen whose camera rays all miss the scene's bounds queues no ray either, but hands its slots on -- REGEN[(j&1)^1],
            // zero at this point otherwise -- and work is left)
            const PaddedCounter* again = D.hPoll + (j % R) * snapshot + (size_t)(Q_REGEN_A + (uint32_t)((j & 1u) ^ 1u)) * NSHARD;
            uint32_t pending = 0; for (uint32_t k = 0; k < NSHARD; k++) pending += snap[k].v + again[k].v;
            if (pending == 0) { totalIters++; break; } // k_raygen(j) consumed the regen queue and produced no rays: the pool had drained
          }
        }
        if (two && twoStreamDelay == 1) launchSpin(st, 300000ull);
        timed(1,
            [&] { launchTrace(st, traceBlocks, false, s->countTraversal, view, ps, qs, D.dCounters.ptr, Q_TRACE_A + par, Q_REGEN_A + (par ^ 1u), dynRefill,
            wideBlocks, U, D.sampleBuf.ptr); });
        traceLaunches++;
        if (two && it > 0) raygenBehindShadow();
        // one launch per shade class in use (scattering events inside a medium are routed to class 2, k_route: it is launched whenever a medium stack exists
        // and OpenPBR does)
        uint32_t shadeMask = s->shadeClassMask | ((rs.mediumStackSize != 0u && (s->shadeClassMask & (1u << SHADE_CLASS_OPBR_BASE))) ? 4u : 0u);
        // BASE hits sit in class 2's queue
        if (U.flags & FLAG_MERGE_SHADE_VARIANTS) shadeMask = (shadeMask & ~(1u << SHADE_CLASS_OPBR_BASE)) | ((shadeMask >> SHADE_CLASS_OPBR_BASE) & 1u) << 2;
        for (uint32_t klass = 0; klass < MAT_CLASS_COUNT; klass++)
          if (shadeMask & (1u << klass)) timed(2,
              [&] { launchShade(st, wideBlocks, klass, (s->shadeClassTextured & (1u << klass)) != 0u, rs.mediumStackSize != 0u, U, view, ps, qs,
              D.dCounters.ptr, par); });
        if (nee) {
          // (the slot-order flag belongs to k_trace_dyn: with dynamic refill off -- TRACE_DYNAMIC 0 -- dynRefill stays 0 so that launchTrace picks the
          // block-synchronous k_trace the grid was sized for, and there is no order to measure) not chosen yet: alternate, and count (below)
          const int32_t order = (dynRefill & 0xffu) == 0u ? 0 : (shadowOrderNow >= 0 ? shadowOrderNow : (int32_t)(totalIters & 1u));
          hipStream_t on = st;
          if (two) { // the shadow launch moves to the second stream, behind this iteration's k_shade
            HIP_TRY(hipEventRecord(D.evShade, st)); HIP_TRY(hipStreamWaitEvent(st2, D.evShade, 0));
            if (twoStreamDelay == 2) launchSpin(st2, 300000ull);
            on = st2;
          }
          timedOn(on, 3,
              [&] { launchTrace(on, traceBlocks, true, s->countTraversal, view, ps, qs, D.dCounters.ptr, Q_SHADOW, Q_SHADOW,
              dynRefill | (order ? TRACE_DYN_SLOT_ORDER : 0u), wideBlocks, U, D.sampleBuf.ptr); });
          if (two) { HIP_TRY(hipEventRecord(D.evShadow, st2)); shadowInFlight = true; }
        }
        // (GATLING_ITER_LOG, with kernel timers on every iteration: what each iteration's queues held -- one sync per iteration, for measurements only)
        if (iterLog) {
          HIP_TRY(hipMemcpyAsync(D.hCounters, D.dCounters.ptr, sizeof(PaddedCounter) * Q_COUNT * NSHARD, hipMemcpyDeviceToHost, st));
          HIP_TRY(hipStreamSynchronize(st));
          auto total = [&](uint32_t q) { uint64_t n = 0; for (uint32_t k = 0; k < NSHARD; k++) n += D.hCounters->count[q][k].v; return n; };
          uint64_t hits = 0; for (uint32_t c = 0; c < MAT_CLASS_COUNT; c++) hits += total(Q_HIT + c);
          iterRows.push_back({ev, total(Q_TRACE_A + par), hits, total(Q_SHADOW), total(Q_REGEN_A + (par ^ 1u)), total(Q_TRACE_A + (par ^ 1u))});
        }
        iters++; totalIters++;
      }
      // (a batch that ended through the poll: its last shadow launches were empty)
      if (shadowInFlight) { (void)hipStreamWaitEvent(st, D.evShadow, 0); shadowInFlight = false; }
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
  S.segments = D.hCounters->segments; S.shadowRays = D.hCounters->shadowRays; S.nodesVisited = D.hCounters->nodesVisited;
      S.trisTested = D.hCounters->trisTested;
  S.shadowNodesVisited = D.hCounters->shadowNodesVisited; S.shadowTrisTested = D.hCounters->shadowTrisTested;
  // (colorRb: an AOV-only render never ran k_init -- the counters would be the previous render's) choose the shadow walks' order once both have been measured
  // on enough rays of this scene: fewer node visits per ray wins
  if (D.slot == 0u && colorRb && s->shadowOrder.load() < 0) {
    for (int m = 0; m < 2; m++) {
      s->shadowOrderRays[m] += D.hCounters->shadowOrderRays[m];
      for (int k = 0; k < 16; k++) s->shadowOrderSteps[m] += D.hCounters->shadowOrderSteps[m][k].v;
    }
    constexpr uint64_t ENOUGH = 1u << 16;
    if (s->shadowOrderRays[0] >= ENOUGH && s->shadowOrderRays[1] >= ENOUGH)
    {
      s->shadowOrder = (double)s->shadowOrderSteps[1] * (double)s->shadowOrderRays[0] < (double)s->shadowOrderSteps[0] * (double)s->shadowOrderRays[1] ? 1 : 0;
      if (getenv("GATLING_BUILD_TIMING")) fprintf(stderr,
          "[gatling_gi] shadow walks: near-to-far %.3f node visits per ray (%llu rays), slot order %.3f (%llu rays) -> %s\n",
                                                  (double)s->shadowOrderSteps[0] / (double)s->shadowOrderRays[0], (unsigned long long)s->shadowOrderRays[0],
                                                  (double)s->shadowOrderSteps[1] / (double)s->shadowOrderRays[1], (unsigned long long)s->shadowOrderRays[1],
                                                      s->shadowOrder.load() ? "slot order" : "near-to-far");
    }
  }
  if (s->countTraversal && D.hCounters->phaseTrips && optionValue("phase_stats", 0)) { // k_path's phase split (counting build)
    const Counters& c = *D.hCounters; const double tot = (double)(c.phaseCycles[0] + c.phaseCycles[1] + c.phaseCycles[2] + c.phaseCycles[3]);
    static const char* names[4] = {"regen", "trace", "shade", "shadow+finish"};
    for (int k = 0; k < 4; k++) fprintf(stderr, "[gatling_gi] k_path phase %-14s %5.1f %% of wave cycles, %5.1f of 64 lanes busy per trip\n", names[k],
        100.0 * (double)c.phaseCycles[k] / tot, (double)c.phaseLanes[k] / (double)c.phaseTrips);
    fprintf(stderr, "[gatling_gi] k_path trips %llu, %.0f cycles per trip and wave\n", (unsigned long long)c.phaseTrips, tot / (double)c.phaseTrips);
  }
  if (s->countTraversal && D.hCounters->dynStats[0] && optionValue("phase_stats", 0)) { // k_trace_dyn's lane accounting (counting build, closest-hit launches)
    const unsigned long long* d = D.hCounters->dynStats; const double st = (double)d[0];
    fprintf(stderr, "[gatling_gi] k_trace_dyn<closest> %llu wave steps: per step %.1f lanes hold a ray, %.1f run the node test, %.1f wait for the triangle "
                    "ring; %.3f batches per step of %.1f pairs; "
                    "a refill every %.2f steps, %.1f lanes each\n", d[0], (double)d[1] / st, (double)d[2] / st, (double)d[3] / st, (double)d[4] / st, d[4]
                        ? (double)d[5] / (double)d[4] : 0.0,
            d[6] ? st / (double)d[6] : 0.0, d[6] ? (double)d[7] / (double)d[6] : 0.0);
  }
  if (D.hCounters->overflow) { setError("giCRender: a work-queue shard overflowed its capacity (internal sizing error); the image is invalid");
      return GI_C_ERROR; }
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
        for (; 2 * k < iterRows[r].evEnd && k < evKind.size(); k++) { float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, D.eventPool[2 * k], D.eventPool[2 * k + 1]); ms4[evKind[k]] += ms; }
        fprintf(stderr, "[gatling_gi] iter %3zu: rays %9llu hits %9llu shadow %9llu ended %9llu continuing %9llu | raygen %7.3f trace+route %7.3f shade %7.3f "
                        "shadow %7.3f ms\n", r,
                (unsigned long long)iterRows[r].traced, (unsigned long long)iterRows[r].hits, (unsigned long long)iterRows[r].shadow,
                    (unsigned long long)iterRows[r].ended,
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
        HIP_TRY(hipMemcpy2DAsync((uint8_t*)rb->stageMem + off, pitch, (uint8_t*)rb->replicaMem[d - 1u] + off, pitch, rowBytes, rows, hipMemcpyDeviceToHost,
            g_ctx.devs[d].stream));
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
  if (rs.mediumStackSize > MAX_MEDIUM_STACK) {
    setError("giCRender: mediumStackSize > 15 cannot be addressed (the payload's medium index has four bits, rp_main_payload.glsl:4-5)");
    return GI_C_ERROR;
  }
  const GiCRenderBuffer* sizeRb = (colorBinding ? colorBinding : &params->aovBindings[0])->renderBuffer;
  const uint32_t width = sizeRb->width, height = sizeRb->height;
  if (width == 0 || height == 0) return GI_C_OK; // Render.Empty-style degenerate target: nothing to do
  if (width > 65535u || height > 65535u) { setError("giCRender: image dimensions exceed 65535 (imageDims packing, rp_main.h:38)"); return GI_C_ERROR; }
  // a camera the ray generation can use (the reference passes whatever Hydra hands it, Gi.cpp:2373-2426; a NaN there is a NaN image): refused, with the field
  // named
  {
    const GiCCameraDesc& c = params->camera;
    const float fields[] = {c.position[0], c.position[1], c.position[2], c.forward[0], c.forward[1], c.forward[2], c.up[0], c.up[1], c.up[2],
                            c.vfov, c.fStop, c.focusDistance, c.focalLength, c.clipStart, c.clipEnd, c.exposure};
    for (float f : fields) if (!std::isfinite(f)) { setError("giCRender: the camera has a non-finite field"); return GI_C_ERROR; }
    const float f2 = (c.forward[0] * c.forward[0] + c.forward[1] * c.forward[1]) + c.forward[2] * c.forward[2],
        u2 = (c.up[0] * c.up[0] + c.up[1] * c.up[1]) + c.up[2] * c.up[2];
    if (!(f2 > 0.0f) || !(u2 > 0.0f) || !std::isfinite(f2) || !std::isfinite(u2) || !std::isfinite(1.0f / sqrtf(f2)) || !std::isfinite(1.0f / sqrtf(u2))) {
      setError("giCRender: the camera's forward or up vector cannot be normalised (zero, denormal or overflowing length)");
      return GI_C_ERROR;
    }
    if (!(c.vfov > 0.0f && c.vfov < 3.14159265f)) { setError("giCRender: the camera's vertical field of view must lie inside (0, pi) radians");
        return GI_C_ERROR; }
    if (!std::isfinite(1.0f / (2.0f * tanf(c.vfov * 0.5f)))) {
      setError("giCRender: the camera's vertical field of view is too small for the image plane distance to be finite");
      return GI_C_ERROR;
    }
  }
  if (const GiCDomeLight* dl = params->domeLight) { // (the dome light's setters take whatever they are given, like the reference's)
    const float fields[] = {dl->rotation[0], dl->rotation[1], dl->rotation[2], dl->rotation[3], dl->baseEmission[0], dl->baseEmission[1], dl->baseEmission[2],
        dl->diffuse, dl->specular};
    for (float f : fields) if (!std::isfinite(f)) { setError("giCRender: the dome light has a non-finite field"); return GI_C_ERROR; }
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
      memcmp(s->oldClear, clear, sizeof(clear)) != 0 || s->oldRowBegin != rowBegin || s->oldRowEnd != rowEnd || s->oldRowStride != rowStride
          || s->oldDome != params->domeLight ||
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

