// gi_kernels.hip -- the wavefront path tracer's stage kernels for gfx950 (CDNA4, wave64).
//
// Replaces the Vulkan ray-tracing megakernel of the reference:
//   rp_main.rgen  (/root/reference/src/gi/shaders/rp_main.rgen:185-521)  -> k_raygen + the host bounce loop
//   traceRayEXT   (rp_main.rgen:381-393, 412-424; HW BVH traversal)        -> k_trace<closest>, k_trace<any>
//   rp_main.chit  (rp_main.chit:132-493)                                   -> k_shade
//   rp_main.miss  (:55-86), rp_main_shadow.miss, the NEE add (rgen:426-429) -> k_raygen (miss term), k_trace<any> epilogue
// One slot per pixel of the tile walks its samples in order, so the per-pixel float accumulation order of
// rp_main.rgen:498 is preserved exactly while different slots sit in different stages/queues.
//
// Data movement (DESIGN.md "Data layout"): rays, hits and shadow rays travel as RECORDS inside the queues -- producers
// write them at the queue position they were allotted, consumers read them back fully coalesced -- and only the
// 64-byte per-pixel Slot (throughput, radiance, accumulator) is gathered/scattered by slot index.
//
// Built with -ffp-contract=off (arithmetic contract, see gi_device_math.h).  Box tests inside the traversal use
// explicit fmaf: they are conservative filters and never influence results.

#include <type_traits>
#include <hip/hip_runtime.h>

#include "gi_device_math.h"
#include "gi_kernels.h"
#include "gi_types.h"
#include "gi_queues.h"
#include "gi_traversal.h"
#include "gi_shading.h"
#include "gi_stages.h"

namespace gi {


// ------------------------------------------------------------------------------------------------
// k_init: every pool slot starts in regen queue A with "no sample in flight"; no work handed out yet
// ------------------------------------------------------------------------------------------------
__global__ void k_init(PathState st, QueueSet qs, Counters* cnt, uint32_t n, uint32_t resetStats)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t per = (n + NSHARD - 1u) / NSHARD; // regen segment s = slots [s*per, min(n,(s+1)*per))
  if (i < Q_COUNT * NSHARD) {
    const uint32_t q = i / NSHARD, sdx = i % NSHARD;
    uint32_t c = 0;
    if (q == Q_REGEN_A) { const uint32_t lo = sdx * per; c = lo < n ? ((n - lo) < per ? (n - lo) : per) : 0u; }
    cnt->count[q][sdx].v = c;
  }
  if (i == 0) {
    if (resetStats) cnt->overflow = 0u; // sticky across the batches of one render: giCRender reads it back once, after the last batch
    cnt->workBase[0].v = 0; cnt->workBase[1].v = 0; for (uint32_t k = 0; k < NCURSOR; k++) { cnt->cursor[0][k].v = 0; cnt->cursor[1][k].v = 0; }
    if (resetStats) { cnt->shadowOrderRays[0] = 0; cnt->shadowOrderRays[1] = 0; for (int k = 0; k < 16; k++) { cnt->shadowOrderSteps[0][k].v = 0; cnt->shadowOrderSteps[1][k].v = 0; }
                      cnt->segments = 0; cnt->shadowRays = 0; cnt->nodesVisited = 0; cnt->trisTested = 0; cnt->shadowNodesVisited = 0; cnt->shadowTrisTested = 0;
                      for (int k = 0; k < 4; k++) { cnt->phaseCycles[k] = 0; cnt->phaseLanes[k] = 0; } cnt->phaseTrips = 0; for (int k = 0; k < 8; k++) cnt->dynStats[k] = 0; }
  }
  for (; i < n; i += gridDim.x * blockDim.x) qs.slot[Q_REGEN_A][(i / per) * qs.cap + (i % per)] = i | REGEN_FRESH; // the slots themselves stay untouched
}

// ------------------------------------------------------------------------------------------------
// k_raygen: persistent-thread ray generation (rp_main.rgen:213-283), the per-sample finish (:483-498) and the miss
// term.  Entry i of the regen queue finishes its sample (if any) into the per-sample colour buffer and takes work item
// w = workBase + i, decoded by work_item (gi_queues.h): by default consecutive entries get consecutive samples of one pixel, pixels in 8x8 blocks.
// ------------------------------------------------------------------------------------------------
// A new path's Slot (rp_main.rgen:274-276): throughput 1, bitfield 0, radiance 0, the rng state after the camera draws, its work item
__device__ __forceinline__ void slot_begin_path(Slot* S, uint32_t rng, uint32_t pixelLocal, uint32_t sLocal)
{
  st4(&S->thr, 1.0f, 1.0f, 1.0f, u2f(0u));
  st4(&S->rad, 0.0f, 0.0f, 0.0f, u2f(rng));
  st4(&S->id, u2f(pixelLocal), u2f(sLocal), u2f(1u), 0.0f);
}
// FLAG_DEFER_SLOT: the first segment of a camera path whose Slot k_raygen did not write has been traced.  A hit writes the Slot now (the path goes on exactly as
// if k_raygen had written it).  A miss retires the sample on the spot -- radiance 0 + throughput 1 x background, then the per-sample finish, the arithmetic of
// k_raygen's finish of a REGEN_MISSED entry (rp_main.miss:68-86, rp_main.rgen:489-496) -- and returns true: the slot goes back to the regen queue as REGEN_FRESH,
// "nothing to finish, memory unwritten".  Scenes with a dome image or medium stacks need the slot at a miss (dome_miss / the scattering test): the caller writes
// it first and takes the ordinary route.
__device__ __forceinline__ void retire_fresh_miss(const FrameUniforms& U, const FreshRec& f, F4* __restrict__ sampleBuf)
{
  uint32_t pixelLocal, sLocal; work_item(U, f.work, pixelLocal, sLocal);
  V3 rad = v3(0.0f, 0.0f, 0.0f) + v3(1.0f, 1.0f, 1.0f) * v3(U.background);
  const float mv = fmax2(rad.x, fmax2(rad.y, rad.z));
  if (mv > U.maxSampleValue) rad = rad * (U.maxSampleValue / mv);
  st4(&sampleBuf[sample_record(U, pixelLocal, sLocal)], fmax2(0.0f, rad.x), fmax2(0.0f, rad.y), fmax2(0.0f, rad.z), 0.0f);
}
__device__ __forceinline__ void begin_fresh_path(const FrameUniforms& U, const PathState& st, uint32_t slot, const FreshRec& f)
{
  uint32_t pixelLocal, sLocal; work_item(U, f.work, pixelLocal, sLocal);
  slot_begin_path(&st.slots[slot], f.rng, pixelLocal, sLocal);
}
// (a first segment that HIT is begun by k_shade, which gathers the ray record and the FreshRec beside it and completes the Slot in one go)

// FLAG_BOUNDS_RETIRE: can the ray reach the scene at all?  A slab test against the root node's bounds (host: padded beyond the dequantised child boxes), widened per
// ray by 30 x the rounding error of (plane - origin) * (1 / d) and decided only by comparisons that a NaN fails -- so it answers "misses" for no ray whose walk
// could accept a triangle (every triangle lies inside its leaf box, every leaf box inside the root's bounds; same contract as the node test).
__device__ __forceinline__ bool ray_misses_bounds(const FrameUniforms& U, const V3& o, const V3& d, float tMin, float tMax)
{
  float tn = tMin, tf = tMax;
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  bool out = false;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float lo = U.sceneLo[a], hi = U.sceneHi[a];
    const float pad = (fabsf(oo[a]) + fmax2(fabsf(lo), fabsf(hi))) * 4.0e-6f;
    if (dd[a] == 0.0f) { out = out || (oo[a] < lo - pad) || (oo[a] > hi + pad); continue; }
    const float inv = 1.0f / dd[a];
    const float t0 = ((lo - pad) - oo[a]) * inv, t1 = ((hi + pad) - oo[a]) * inv;
    float nearT = t0 < t1 ? t0 : t1, farT = t0 < t1 ? t1 : t0; // (the padded planes keep their order; a NaN leaves the interval alone below)
    nearT -= fabsf(nearT) * 1.0e-5f; farT += fabsf(farT) * 1.0e-5f;
    if (nearT > tn) tn = nearT;
    if (farT < tf) tf = farT;
  }
  return out || tn > tf;
}

// RAYGEN_ITEMS regen entries per thread and trip: the trip is otherwise two barriers and an atomic round trip around a chain of dependent loads (entry -> slot ->
// finished sample), and the entries of one thread are independent of each other.
constexpr int RAYGEN_ITEMS = 2; static_assert(RAYGEN_ITEMS <= (int)APPEND_ITEMS_MAX, "shardCapacity's slack"); // (1 -> 2: raygen stage -6 % on C3 / C4; 4: the same, r05x)
// FLAG_TWO_STREAM: the counters k_trace / k_route of iteration `par`'s parity need zeroed (gi_queues.h zero_closest_counters); one wave
__global__ __launch_bounds__(64) void k_zero_closest(Counters* cnt, uint32_t par) { zero_closest_counters(cnt, par); }
// test hook (GATLING_OPTIONS=two_stream_delay): holds a stream for about `ns` nanoseconds so that the other one runs ahead
__global__ __launch_bounds__(64) void k_spin(unsigned long long ns, uint32_t* sink)
{
  const unsigned long long t0 = wall_clock64(); // 100 MHz constant clock
  while ((wall_clock64() - t0) * 10ull < ns) { }
  if (sink && threadIdx.x == 0xffffu) *sink = 0u;
}
__global__ __launch_bounds__(BLOCK) void k_raygen(FrameUniforms U, PathState st, QueueSet qs, Counters* cnt, uint32_t par, F4* __restrict__ sampleBuf)
{
  __shared__ AppendScratch<2> sh;
  const uint32_t qIn = Q_REGEN_A + par, qOut = Q_TRACE_A + par, qAgain = Q_REGEN_A + (par ^ 1u);
  const bool boundsRetire = (U.flags & FLAG_BOUNDS_RETIRE) != 0u;
  QueueReader rd; reader_init(rd, cnt, qIn, qs.cap);
  const uint32_t n = rd.pre[NSHARD];
  const uint32_t workBase = cnt->workBase[par].v;
  uint32_t nRetired = 0u;
  if (blockIdx.x == 0) {
    zero_next_counters(cnt, par, !boundsRetire, (U.flags & FLAG_TWO_STREAM) != 0u);
    if (threadIdx.x == 0) { const uint32_t left = U.workTotal - workBase; cnt->workBase[par ^ 1u].v = workBase + (n < left ? n : left); }
  }
  const uint32_t stride = gridDim.x * BLOCK * RAYGEN_ITEMS;
  const uint32_t qid[2] = {qOut, qAgain};
  uint32_t trip = 0;
  for (uint32_t base = blockIdx.x * BLOCK * RAYGEN_ITEMS; base < n; base += stride, trip++) {
    uint32_t which[RAYGEN_ITEMS], idx[RAYGEN_ITEMS], slotOf[RAYGEN_ITEMS]; FreshRec freshOf[RAYGEN_ITEMS];
    V3 originOf[RAYGEN_ITEMS], dirOf[RAYGEN_ITEMS]; float tMinOf[RAYGEN_ITEMS], tMaxOf[RAYGEN_ITEMS];
    uint32_t entryOf[RAYGEN_ITEMS];
#pragma unroll
    for (int k = 0; k < RAYGEN_ITEMS; k++) { // every item's queue entry first: independent loads in flight
      const uint32_t i = base + (uint32_t)k * BLOCK + threadIdx.x;
      entryOf[k] = i < n ? qs.slot[qIn][reader_index(rd, i)] : 0u;
    }
#pragma unroll
    for (int k = 0; k < RAYGEN_ITEMS; k++) {
      const uint32_t i = base + (uint32_t)k * BLOCK + threadIdx.x;
      bool more = false; uint32_t slot = 0, rng = 0u; FreshRec fresh{0u, 0u};
      V3 origin = v3(0.0f, 0.0f, 0.0f), dir = origin; float tMin = 0.0f, tMax = GI_FLT_MAX;
      if (i < n) {
        const uint32_t entry = entryOf[k];
        slot = entry & ~(REGEN_MISSED | REGEN_FRESH);
        Slot* S = &st.slots[slot];
        F4 id = F4{0.0f, 0.0f, 0.0f, 0.0f};
        if (!(entry & REGEN_FRESH)) id = ld4(&S->id);
        if (f2u(id.z) != 0u) { // finish the sample that just terminated (:489-496) -> per-sample colour buffer
          F4 r = ld4(&S->rad);
          V3 rad = v3(r.x, r.y, r.z);
          if (entry & REGEN_MISSED) {
            // the path left the scene: uniform fallback dome == colour-AOV clear value (rp_main.miss:68-86, Gi.cpp:2184-2199,
            // 2232-2238).  k_trace routes misses straight here; nothing after the miss can change the sample any more.
            F4 tb = ld4(&S->thr);
            rad = rad + v3(tb.x, tb.y, tb.z) * v3(U.background);
            if (st.neeKey && (f2u(tb.w) & 0x00000fffu) == 0u) nee_aov_record(st, slot, false); // primary miss: no light sampled, "not shadowed" (rp_main.rgen:431-435)
          }
          if (st.bouncesAov && U.batchFirstSample + f2u(id.y) == U.spp - 1u) { // Bounces AOV: the pixel's last sample (rp_main.rgen:483-486)
            // a path that left the scene was routed here straight from k_trace, before the loop's bounce++ (rp_main.rgen:480)
            const uint32_t bounces = ((f2u(S->thr.w) + ((entry & REGEN_MISSED) ? 1u : 0u)) & 0x00000fffu), maxB = U.maxBounces < 0x00000fffu ? U.maxBounces : 0x00000fffu;
            const V3 c = gi_colormap_inferno((float)bounces / (float)maxB);
            F4* dst = &st.bouncesAov[tile_to_image_pixel(U, f2u(id.x))];
            dst->x = c.x; dst->y = c.y; dst->z = c.z;
          }
          if (st.pathSegments) atomicAdd(&st.pathSegments[f2u(id.x)], (f2u(S->thr.w) + ((entry & REGEN_MISSED) ? 1u : 0u)) & 0x00000fffu); // integer sum: order-free
          float mv = fmax2(rad.x, fmax2(rad.y, rad.z));
          if (mv > U.maxSampleValue) rad = rad * (U.maxSampleValue / mv);
          // one aligned 16-byte store: a 12-byte record straddles DRAM sectors and costs two read-modify-writes
          st4(&sampleBuf[sample_record(U, f2u(id.x), f2u(id.y))], fmax2(0.0f, rad.x), fmax2(0.0f, rad.y), fmax2(0.0f, rad.z), 0.0f);
        }
        const uint32_t w = workBase + i; // < 2^32 by construction of the batches (host)
        more = (i < U.workTotal - workBase) && (w < U.workTotal);
        if (more) {
          uint32_t pixelLocal, sLocal; work_item(U, w, pixelLocal, sLocal);
          const uint32_t pixelIndex = tile_to_image_pixel(U, pixelLocal); // :195 (global index: RNG is tile independent)
          const uint32_t sampleIndex = U.sampleOffset + U.batchFirstSample + sLocal;
          make_camera_ray(U, pixelIndex, sampleIndex, origin, dir, tMin, tMax, rng);
          fresh.rng = rng; fresh.work = w;
          if (!(U.flags & FLAG_DEFER_SLOT)) slot_begin_path(S, rng, pixelLocal, sLocal); // :274-276 (deferred: written when the first segment hits, route_fresh)
        }
      }
      // a camera ray that cannot reach the scene: what k_route does with a fresh miss (retire_fresh_miss; the segment is counted below), minus the 52-byte record,
      // the traversal step and the routing pass -- the slot goes straight to the next k_raygen
      bool again = false;
      if (boundsRetire && more && ray_misses_bounds(U, origin, dir, tMin, tMax)) { retire_fresh_miss(U, fresh, sampleBuf); more = false; again = true; nRetired++; }
      which[k] = more ? 0u : (again ? 1u : 2u);
      slotOf[k] = slot; freshOf[k] = fresh; originOf[k] = origin; dirOf[k] = dir; tMinOf[k] = tMin; tMaxOf[k] = tMax;
    }
    block_append_items<2, RAYGEN_ITEMS>(sh, trip, which, qid, qs.cap, cnt, idx);
#pragma unroll
    for (int k = 0; k < RAYGEN_ITEMS; k++) {
      if (which[k] == 0u) {
        const bool defer = (U.flags & FLAG_DEFER_SLOT) != 0u;
        qs.slot[qOut][idx[k]] = defer ? (slotOf[k] | TRACE_FRESH) : slotOf[k];
        st4(&qs.a[qOut][idx[k]], originOf[k].x, originOf[k].y, originOf[k].z, tMinOf[k]);
        st4(&qs.b[qOut][idx[k]], dirOf[k].x, dirOf[k].y, dirOf[k].z, tMaxOf[k]);
        if (defer) qs.fresh[par][idx[k]] = freshOf[k]; // 8 bytes beside the record, written and read in queue order
      } else if (which[k] == 1u) qs.slot[qAgain][idx[k]] = slotOf[k] | REGEN_FRESH;
    }
  }
  if (boundsRetire) { // the retired camera rays are segments of their paths (Counters::segments equals the oracle's count): one atomic per wave
    unsigned long long c = nRetired;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if (__lane_id() == 0u && c) atomicAdd(&cnt->segments, c);
  }
}

// ------------------------------------------------------------------------------------------------
// k_accumulate: folds one batch of per-sample colours into the per-pixel running sum IN SAMPLE ORDER
// (pixel_color += sample_color * invSpp, rp_main.rgen:498) and, after the last batch, writes the colour AOV with
// the progressive blend of rp_main.rgen:506-515.  One thread per pixel; reads are coalesced across pixels (sample-major buffer) or whole lines per thread (pixel-major).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_accumulate(FrameUniforms U, const F4* __restrict__ sampleBuf, F4* __restrict__ accum, F4* __restrict__ colorOut,
                                                      uint32_t firstBatch, uint32_t lastBatch)
{
  const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= U.pixelCount) return;
  V3 pixelColor = v3(0.0f, 0.0f, 0.0f);
  if (!firstBatch) { const F4 a = ld4(&accum[p]); pixelColor = v3(a.x, a.y, a.z); }
  if (U.flags & FLAG_PIXEL_MAJOR) { // the pixel's samples are one contiguous run: a thread streams its own lines, eight records (one 128-byte line) per round
    const F4* src = sampleBuf + (size_t)p * U.batchSamples;
    uint32_t s = 0;
    for (; s + 8u <= U.batchSamples; s += 8u) {
      F4 r[8];
#pragma unroll
      for (uint32_t k = 0; k < 8u; k++) r[k] = ld4(&src[s + k]);
#pragma unroll
      for (uint32_t k = 0; k < 8u; k++) pixelColor = pixelColor + v3(r[k].x, r[k].y, r[k].z) * U.invSpp;
    }
    for (; s < U.batchSamples; s++) { const F4 r = ld4(&src[s]); pixelColor = pixelColor + v3(r.x, r.y, r.z) * U.invSpp; }
  } else {
    for (uint32_t s = 0; s < U.batchSamples; s++) {
      const F4 src = ld4(&sampleBuf[(size_t)s * U.pixelCount + p]);
      pixelColor = pixelColor + v3(src.x, src.y, src.z) * U.invSpp;
    }
  }
  if (!lastBatch) { st4(&accum[p], pixelColor.x, pixelColor.y, pixelColor.z, 0.0f); return; }
  const uint32_t pixelIndex = tile_to_image_pixel(U, p);
  V3 prev = pixelColor;
  if ((U.flags & FLAG_PROGRESSIVE) && U.sampleOffset > 0u) { const F4 q = ld4(&colorOut[pixelIndex]); prev = v3(q.x, q.y, q.z); }
  const V3 c = (prev * U.sampleOffsetF + pixelColor * U.sppF) * U.invTotalSampleCount;
  st4(&colorOut[pixelIndex], c.x, c.y, c.z, 1.0f);
}

__global__ void k_resolve_nee(FrameUniforms U, const unsigned long long* __restrict__ key, F4* __restrict__ aov, uint32_t pixelCount)
{
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pixelCount) return;
  const unsigned long long k = key[p];
  if (k == 0ull) return; // no sample reached the shadow test (maxBounces == 0): the AOV keeps its clear value
  F4* dst = &aov[tile_to_image_pixel(U, p)];
  dst->x = (k & 1ull) ? 1.0f : 0.0f; dst->y = (k & 1ull) ? 0.0f : 1.0f; dst->z = 0.0f;
}

template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool ALL_LDS, bool CUTOUT, bool DOME>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace(SceneView sc, PathState st, QueueSet qs, Counters* cnt, uint32_t qIn, uint32_t qMiss, uint32_t ldsNodes, uint32_t ldsTris,
                                                       FrameUniforms U, F4* __restrict__ sampleBuf)
{
  // dynamic LDS, sized by the launch to what this scene actually stages: [stack | nodes | triangles]
  extern __shared__ uint4 s_dyn[];
  uint2 (*s_stack)[TRACE_BLOCK] = reinterpret_cast<uint2 (*)[TRACE_BLOCK]>(s_dyn);
  uint4* s_nodes = s_dyn + (STACK * TRACE_BLOCK * sizeof(uint2)) / sizeof(uint4);
  uint4* s_tris = s_nodes + ldsNodes * 5u;
  __shared__ AppendScratch<1 + MAT_CLASS_COUNT> sh;
  __shared__ WaveTri s_wave[TRACE_BLOCK / 64];
  WaveTri& W = s_wave[threadIdx.x >> 6];
  QueueReader rd; reader_init(rd, cnt, qIn, qs.cap);
  const uint32_t n = rd.pre[NSHARD];
  if (blockIdx.x == 0 && threadIdx.x == 0) { if (ANYHIT) cnt->shadowRays += n; else cnt->segments += n; } // single writer per launch
  if (blockIdx.x * TRACE_BLOCK >= n) return; // whole block idle (uniform)
  for (uint32_t i = threadIdx.x; i < ldsNodes * 5u; i += TRACE_BLOCK) s_nodes[i] = reinterpret_cast<const uint4*>(sc.nodes)[i];
  for (uint32_t i = threadIdx.x; i < ldsTris * 3u; i += TRACE_BLOCK) s_tris[i] = reinterpret_cast<const uint4*>(sc.tris)[(i / 3u) * 4u + (i % 3u)];
  __syncthreads();

  TraceCounters tc{0u, 0u};
  RayTrav R; trav_init(R, v3(0.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, 1.0f), 0.0f, 0.0f);
  uint2 overflow[OVERFLOW ? OVF_STACK : 1];
  const uint32_t stride = gridDim.x * TRACE_BLOCK;
  uint32_t trip = 0;
  for (uint32_t base = blockIdx.x * TRACE_BLOCK; base < n; base += stride, trip++) {
    const uint32_t i = base + threadIdx.x;
    bool hit = false, miss = false; uint32_t slot = 0, mat = 0;
    float t = 0.0f, u = 0.0f, v = 0.0f; uint32_t tri = MISS; F4 rdir = F4{0.0f, 0.0f, 0.0f, 0.0f}, ro = rdir;
    bool alive = false, fresh = false; uint32_t r = 0u, rng = 0u;
    if (i < n) {
      r = reader_index(rd, i);
      slot = qs.slot[qIn][r];
      if (!ANYHIT) { fresh = (slot & TRACE_FRESH) != 0u; slot &= ~TRACE_FRESH; } // camera ray of a path whose Slot is still unwritten (FLAG_DEFER_SLOT)
      ro = ld4(&qs.a[qIn][r]);
      rdir = ld4(&qs.b[qIn][r]);
      rng = CUTOUT ? (ANYHIT ? f2u(rdir.w) : (fresh ? qs.fresh[qIn - Q_TRACE_A][r].rng : f2u(st.slots[slot].rad.w))) : 0u; // the any-hit test needs the path's rng state (shadow rays carry their copy)
      // shadow ray (rp_main.rgen:397-429): origin = next ray origin, tMin 0.01, tMax = distance to the light sample
      if (!ANYHIT) trav_init(R, v3(ro.x, ro.y, ro.z), v3(rdir.x, rdir.y, rdir.z), ro.w, rdir.w);
      else trav_init(R, v3(ro.x, ro.y, ro.z), v3(rdir.x, rdir.y, rdir.z), 0.01f, ro.w);
      wave_ray_begin(W, R.tBest);
      alive = true;
    }
    while (__ballot(alive)) {
      if (wave_step<ANYHIT, COUNT, STACK, OVERFLOW, ALL_LDS, CUTOUT>(R, alive, W, sc, s_nodes, ldsNodes, s_tris, ldsTris, s_stack, overflow, tc, rng)) alive = false;
    }
    if (i < n) {
      wave_ray_end(W, R);
      if (!ANYHIT) {
        hit = R.found; miss = !hit; t = R.tBest; u = R.bestU; v = R.bestV; tri = R.bestTri; mat = R.bestMat;
      } else {
        F4 nc = F4{0.0f, 0.0f, 0.0f, 0.0f}; // (neeContrib, 1 = emitted at bounce 0)
        if (!R.found || st.neeKey) nc = ld4(&qs.c[qIn][r]);
        if (!R.found) {
          Slot* S = &st.slots[slot];
          F4 rr = ld4(&S->rad);
          st4(&S->rad, rr.x + nc.x, rr.y + nc.y, rr.z + nc.z, rr.w);
        }
        if (st.neeKey && nc.w != 0.0f) nee_aov_record(st, slot, R.found);
      }
    }
    if (!ANYHIT) {
      // sort by outcome and material class: hits go to their class's shade queue as (slot, hit, direction) records, misses
      // straight to k_raygen
      uint32_t klass = (mat >> 24) & 0xfu;
      if (klass == SHADE_CLASS_OPBR_BASE && (U.flags & FLAG_MERGE_SHADE_VARIANTS)) klass = 2u; // thin batches: one OpenPBR launch (same bits: gi_shading.h "BASE variant")
      bool retired = false, freshHit = false;
      if (fresh) { // first segment of a path k_raygen did not write: k_shade begins it (hit), or it is begun / retired here (miss)
        const FreshRec f = qs.fresh[qIn - Q_TRACE_A][r];
        if (hit) freshHit = true;
        else if (DOME && (sc.domeTexture != 0u || sc.mediumStackSize != 0u)) begin_fresh_path(U, st, slot, f);
        else { retire_fresh_miss(U, f, sampleBuf); retired = true; }
      }
      bool volMiss = false; // the segment ended inside a medium: a scattering event for k_shade<2>, not a miss (rp_main.miss:57-66)
      if (DOME && miss && sc.mediumStackSize) {
        volMiss = payload_medium_idx(f2u(st.slots[slot].thr.w), sc.mediumStackSize < MAX_MEDIUM_STACK ? sc.mediumStackSize : MAX_MEDIUM_STACK) > 0u;
        if (volMiss) { miss = false; klass = 2u; }
      }
      bool pred[1 + MAT_CLASS_COUNT]; uint32_t qid[1 + MAT_CLASS_COUNT]; uint32_t idx[1 + MAT_CLASS_COUNT];
      pred[0] = miss; qid[0] = qMiss;
#pragma unroll
      for (uint32_t c = 0; c < MAT_CLASS_COUNT; c++) { pred[1 + c] = (hit || volMiss) && klass == c; qid[1 + c] = Q_HIT + c; }
      block_append<1 + MAT_CLASS_COUNT>(sh, trip, pred, qid, qs.cap, cnt, idx);
      if (hit || volMiss) { // the result stays in the ray's record (the form k_trace_dyn leaves), the class queue gets its index
        if (!volMiss) st4(&qs.a[qIn][r], t, u, v, u2f(tri | (klass << 28)));
        else { st4(&qs.a[qIn][r], rdir.w, ro.x, ro.y, u2f(MISS)); reinterpret_cast<float*>(&qs.b[qIn][r])[3] = ro.z; } // (tMax, origin) for the scattering event
        qs.slot[Q_HIT + klass][idx[1 + klass]] = r | (freshHit ? HIT_FRESH : 0u) | (volMiss ? HIT_VOLUME : 0u);
      }
      if (miss) {
        if (DOME && sc.domeTexture) { dome_miss(sc, st, slot, v3(rdir.x, rdir.y, rdir.z)); qs.slot[qMiss][idx[0]] = slot; } // scene has a dome light image
        else qs.slot[qMiss][idx[0]] = slot | (retired ? REGEN_FRESH : REGEN_MISSED);
      }
    }
  }
  if (COUNT) { // measurement builds only: one atomic pair per wave
    unsigned long long a = tc.nodes, b = tc.tris;
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    if (__lane_id() == 0) { atomicAdd(ANYHIT ? &cnt->shadowNodesVisited : &cnt->nodesVisited, a); atomicAdd(ANYHIT ? &cnt->shadowTrisTested : &cnt->trisTested, b); }
  }
}

// ------------------------------------------------------------------------------------------------
// k_trace_dyn: traversal for scenes that do not fit LDS.  Ray cost has a long tail there (a ray through dense geometry
// visits several times the average node count), so "one ray per lane until the whole block is done" leaves most lanes
// idle.  Here every wave is persistent and independent: lanes that finish write their result IN PLACE over the ray
// record (a = (t, u, v, triangle | class << 28) or (tMax, origin.xy, MISS)) and, once `refill` lanes of the wave are idle, the wave hands
// them new rays.  No barriers, no appends; k_route then streams the results into the per-class shade queues / the regen queue.
// Per-ray arithmetic is trav_step's, i.e. identical to k_trace's.
//
// The kernel is bound by instruction issue along each wave's dependent chain (its time follows the instruction count of a
// step one to one), so the loop is written for few instructions per step:
//   * a lane keeps only what the walk needs (RayWalk); the nearest hit lives in the wave's LDS record, where the winning lane of a triangle batch
//     leaves the FINISHED 16-byte result -- ending a ray is one ds_read_b128 + one global store (the record is pre-set to the miss result when
//     the ray begins);
//   * the launch's rays are the queue's NSHARD shards, and shard k IS cursor range k: a claimed ray's record index is `shard * cap + position`,
//     no search through the shard prefix sums;
//   * everything wave-uniform (claims, chunk and ring bookkeeping) is forced into SGPRs with readfirstlane;
//   * shadow walks (ANYHIT) end at their first hit, so near-to-far order is optional for them: the SLOT instantiation visits children in slot order (no octant flip
//     in its node test) and the host launches whichever order the scene's shadow walks have been cheaper in (trace_dyn_body, gi_render.cpp shadowOrder).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t DYN_SLOT_ORDER = 0x200u; // bit in k_trace_dyn's `refill` argument (shadow launches): the launch is the slot-order instantiation (the prologue counts its rays as such)
constexpr uint32_t DYN_CLAIM = 128;   // rays per cursor atomic (a device-scope atomic on one line completes ~88 times per microsecond; 64 / 256 / 512 measured: r04x)
constexpr uint32_t DYN_FLUSH_AT = 8;  // the triangle ring is flushed below 64 pairs once this many finished walks wait for it (0 / 2 / 24 measured: r04c)
constexpr int DYN_WAVES = 5; // resident waves per SIMD the register allocation aims for (84 - 96 VGPRs).  6 waves (80 VGPRs, 3 - 8 dwords spilled) do not pay:
                             // C3 trace 55.3 -> 56.2 ms, C5 119 -> 123 (profiles/r05r_six_waves_variants.txt) -- the SIMD's issue rate is shared, more waves do not raise it
constexpr uint32_t DYN_THIN_WALKERS = 8; // the ring is flushed at the end of every step while this few lanes walk (16: the same, r05d)

template <bool TWO> struct DynRay { using type = RayWalk; };
template <> struct DynRay<true> { using type = RayTrav2; };
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); } // wave-uniform by construction: keep it in an SGPR

// HELP: the instantiation with helper lanes (thin launches, below); a full launch runs the one without -- the bookkeeping alone (record and ring addresses computed from a
// register instead of the lane number, the stack base) costs the full launches 3 - 6 % of their traversal time when it is compiled into their loop (r05g).
template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool CUTOUT, bool TWO, bool HELP, bool SLOT = false>
__device__ __forceinline__ void trace_dyn_body(const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t qIn, uint32_t refill, WaveTri& W,
                                               uint32_t shardCount, uint32_t claim)
{
  // Shadow walks end at their first hit, so near-to-far order is not needed for the result -- and which order finds an occluder sooner depends on the scene (C3's soup:
  // slot order visits 7 % fewer nodes; C5's interior, where the occluders sit near the shaded surface: 28 % more, r05c).  The host tells the launch which one to use
  // (the SLOT instantiation: no octant flip in the node test, children are visited in slot order) and the kernel counts the walks' node visits for it to choose by.
  constexpr bool slotOrder = ANYHIT && !TWO && SLOT;
  uint32_t walkSteps = 0u;
  extern __shared__ uint4 s_dyn[];
  uint2 (*s_stack)[TRACE_BLOCK] = reinterpret_cast<uint2 (*)[TRACE_BLOCK]>(s_dyn);
  const uint32_t lane = __lane_id();
  const uint32_t cap = qs.cap;
  PaddedCounter* cursors = cnt->cursor[ANYHIT ? 1 : 0];
  TraceCounters tc{0u, 0u};
  typename DynRay<TWO>::type R;
  auto ray_init = [&](V3 o, V3 d, float tMin, float tMax) { if constexpr (TWO) trav2_init(R, o, d, tMin, tMax); else walk_init(R, o, d, tMin, tMax); };
  ray_init(v3(0.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, 1.0f), 0.0f, 0.0f);
  uint2 overflow[OVERFLOW ? OVF_STACK : 1];
  bool alive = false, draining = false;
  uint32_t rec = 0u, rng = 0u, lastEnd = 0u;
  uint32_t ringHead = 0u, ringTail = 0u; // wave-uniform: the triangle ring persists across steps
  refill = uni(refill & 0xffu);
  // The wave claims rays DYN_CLAIM at a time from the cursor of a shard and takes them 64 at a time: lane j prefetches ray j of the chunk into registers; lanes
  // that run idle are then handed the chunk's rays in order with register shuffles, so a refill never waits on memory.  A wave starts on the shard of its
  // index and moves on when a shard runs dry, so the shards also balance each other at the end of the launch.
  F4 pro = F4{0.0f, 0.0f, 0.0f, 0.0f}, prd = F4{0.0f, 0.0f, 0.0f, 0.0f}; uint32_t prec = 0u, prng = 0u;
  uint32_t chunkCount = 0u, chunkUsed = 0u;
  uint32_t range = uni((blockIdx.x * (TRACE_BLOCK / 64u) + (threadIdx.x >> 6)) % NCURSOR), rangesTried = 0u, rangeEnd = 0u;
  uint32_t claimBase = 0u, claimLeft = 0u;
  // (End of a launch: every wave finds its shard dry and walks the other cursors, 8 atomics per wave.  Publishing "dry" bits on a line of their own and reading them first
  // saves those atomics and changed no launch time, r05k: not built in.)
  auto next_chunk = [&]() {
    while (claimLeft == 0u && rangesTried < NCURSOR) {
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)shardCount, (int)range);
      uint32_t b = 0xffffffffu;
      if (lane == 0u && hi != 0u) b = atomicAdd(&cursors[range].v, claim);
      b = uni(b);
      if (b < hi) { claimBase = b; claimLeft = (hi - b) < claim ? ((hi - b + 63u) & ~63u) : claim; rangeEnd = hi; rangesTried = 0u; }
      else { range = (range + 1u) % NCURSOR; rangesTried++; }
    }
    chunkCount = 0u; chunkUsed = 0u;
    if (claimLeft) {
      const uint32_t base = claimBase;
      claimBase += 64u; claimLeft -= 64u;
      chunkCount = rangeEnd - base < 64u ? rangeEnd - base : 64u; // (the last chunk of a shard may be partial)
      if (lane < chunkCount) {
        prec = range * cap + base + lane;
        pro = ld4(&qs.a[qIn][prec]);
        prd = ld4(&qs.b[qIn][prec]);
        if (CUTOUT) { // the any-hit test needs the path's rng state (shadow rays carry their copy; a camera ray whose Slot is still unwritten has it beside the record)
          if (ANYHIT) prng = f2u(prd.w);
          else { const uint32_t sw = qs.slot[qIn][prec]; prng = (sw & TRACE_FRESH) ? qs.fresh[qIn - Q_TRACE_A][prec].rng : f2u(st.slots[sw].rad.w); }
        }
      }
    }
  };
  // A ray's walk can be shared (flat layout): once the wave has nothing left to claim, lanes without a ray HELP the longest walks instead of idling -- a helper takes
  // the bottom entry of a walking lane's traversal stack (the oldest deferred group, i.e. the largest subtree still to do), copies the ray and walks that part.  All
  // lanes working on a ray report to the same LDS record (`key`: the lane that owns the ray -- the atomicMin hit key is order-independent, so the result does not
  // change), pairs in the triangle ring name the owner (whose registers hold the ray until the end), and the owner's record counts its live helpers: the ray is
  // finished when the owner's own walk has drained and that count is zero.  Why: a launch ends with its slowest ray, a 100-step ray outlives the average one six
  // times over, and the thin launches of a low-spp frame (one sample per pixel and call is hdGatling's default) are NOTHING BUT that tail.  Only in thin launches:
  // a helper walks far subtrees before the near hit that would have culled them is known, and in the tail of a full launch that extra work costs the waves that
  // still have rays more than the tail shortens (r05f: trace +2.5 ... +5 % on C3 / C4 / C5 with helpers everywhere; a spp-1 frame -9 ... -16 % with them).
  uint32_t keyReg = lane, base = 0u; // the lane whose LDS record this walk reports to; first stack entry that is still this walk's
  bool helper = false;
#define key (HELP ? keyReg : lane)
  unsigned long long ds[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}; // (COUNT) lane accounting, Counters::dynStats
  next_chunk();
  for (;;) {
    const unsigned long long idle = __ballot(!alive);
    const uint32_t nIdle = (uint32_t)__popcll(idle);
    if (nIdle >= refill && chunkUsed < chunkCount) {
      const uint32_t avail = chunkCount - chunkUsed, take = nIdle < avail ? nIdle : avail;
      const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
      const int src = (int)((chunkUsed + rank) & 63u);
      const F4 ro = F4{__shfl(pro.x, src), __shfl(pro.y, src), __shfl(pro.z, src), __shfl(pro.w, src)};
      const F4 rdir = F4{__shfl(prd.x, src), __shfl(prd.y, src), __shfl(prd.z, src), __shfl(prd.w, src)};
      const uint32_t srec = (uint32_t)__shfl((int)prec, src);
      const uint32_t srng = CUTOUT ? (uint32_t)__shfl((int)prng, src) : 0u;
      if (!alive && rank < take) {
        rec = srec; rng = srng;
        if (!ANYHIT) ray_init(v3(ro.x, ro.y, ro.z), v3(rdir.x, rdir.y, rdir.z), ro.w, rdir.w);
        else ray_init(v3(ro.x, ro.y, ro.z), v3(rdir.x, rdir.y, rdir.z), 0.01f, ro.w); // shadow ray (rp_main.rgen:397-429)
        wave_ray_begin(W, R.tBest);
        wt_hit_put(W, lane, f2u(ro.x), f2u(ro.y), MISS, 0u); // the result if nothing is hit: (tMax, origin.xy, MISS) -- k_route needs the origin for scattering events (medium stacks only); no helpers
        alive = true; draining = false; lastEnd = ringHead; // no pair of this ray is pending
        keyReg = lane; base = 0u; helper = false;
      }
      if (COUNT) { ds[6]++; ds[7] += take; }
      chunkUsed += take;
      if (chunkUsed == chunkCount) next_chunk(); // loads complete while the wave keeps traversing
    } else if (nIdle == 64u) break; // nothing in flight and nothing left to claim (an exhausted chunk is replaced at once, so chunkUsed == chunkCount means there is none)
    else if (HELP && chunkCount == 0u && nIdle != 0u) {
      const unsigned long long donors = __ballot(alive && !draining && R.sp > base && base < STACK);
      if (donors) {
        // the k-th idle lane helps the k-th donor: donors leave their lane number in lane k (forward permute; the other lanes aim at lane 63, which no helper reads:
        // with a non-donor in the wave there are at most 63 donors, ranks 0 .. 62)
        const bool donor = alive && !draining && R.sp > base && base < STACK; // (entries beyond STACK live in the lane's scratch: OVERFLOW variants)
        const uint32_t dRank = (uint32_t)__popcll(donors & ((1ull << lane) - 1ull)), tRank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
        const uint32_t compact = (uint32_t)__builtin_amdgcn_ds_permute((int)((donor ? dRank : 63u) << 2), (int)lane);
        const uint32_t nPairs = (uint32_t)__popcll(donors) < nIdle ? (uint32_t)__popcll(donors) : nIdle;
        const bool thief = !alive && tRank < nPairs;
        const int from = (int)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(tRank << 2), (int)compact); // (all lanes: wave-uniform control flow; only thieves use it)
        const uint32_t dBase = (uint32_t)__shfl((int)base, from), dKey = (uint32_t)__shfl((int)key, from);
        const V3 o = v3(__shfl(R.o.x, from), __shfl(R.o.y, from), __shfl(R.o.z, from)), d = v3(__shfl(R.d.x, from), __shfl(R.d.y, from), __shfl(R.d.z, from));
        const float idx = __shfl(R.idx, from), idy = __shfl(R.idy, from), idz = __shfl(R.idz, from), tMin = __shfl(R.tMin, from), tBest = __shfl(R.tBest, from);
        const uint32_t octinv = (uint32_t)__shfl((int)R.octinv, from), drng = CUTOUT ? (uint32_t)__shfl((int)rng, from) : 0u;
        // a donor that was paired gives up its bottom entry (its rank is below the number of pairs)
        if (donor && dRank < nPairs) base++;
        if (thief) {
          R.o = o; R.d = d; R.idx = idx; R.idy = idy; R.idz = idz; R.tMin = tMin; R.tBest = tBest; R.octinv = octinv; rng = drng;
          R.G = s_stack[dBase][(threadIdx.x & ~63u) + (uint32_t)from]; // (bottom entries live in LDS for every STACK / OVERFLOW variant)
          R.sp = 0u; base = 0u; keyReg = dKey; helper = true;
          wt_helpers_add(W, key, 1u);
          alive = true; draining = false; lastEnd = ringHead;
        }
      }
    }
    bool done = false;
    if constexpr (TWO) done = wave_step2<ANYHIT, COUNT, CUTOUT>(R, alive, W, sc, s_stack, tc, rng);
    else {
      // The triangle ring is carried from step to step: a node step yields fewer pairs than a batch holds (C3: 23 per step, C4: 18), so flushing at the end of every step
      // ran the ~110-instruction batch at a third of its lanes.  A batch runs when 64 pairs are pending; the rest waits.  A ray whose walk has ended while pairs of it are
      // still pending is DRAINING: its lane keeps the ray (a pending pair fetches the ray from its owner lane at batch time) and sits out the node phases until the ring
      // has moved past its last pair (the ring is FIFO: `head` has reached `lastEnd`).  The ring is flushed below 64 pairs when DYN_FLUSH_AT or more lanes are blocked like
      // that, or when few lanes walk.  Results do not depend on any of this (the hit key under atomicMin does not depend on when a pair is tested); only the culling distance
      // a walking ray sees may lag by a step or two.
      auto batch = [&](uint32_t n) { wave_tri_batch<COUNT, false, CUTOUT, true>(W, ringHead, n, R, rng, sc, nullptr, 0u, tc); ringHead += n; if (COUNT) { ds[4]++; ds[5] += n; } };
      const bool walking = alive && !draining;
      if (COUNT) { ds[0]++; ds[1] += (unsigned long long)__popcll(__ballot(alive)); ds[2] += (unsigned long long)__popcll(__ballot(walking)); ds[3] += (unsigned long long)__popcll(__ballot(alive && draining)); }
      uint2 Gt = make_uint2(0u, 0u);
      if (walking) { Gt = trav_node<COUNT, STACK, OVERFLOW, false, !slotOrder>(R, sc, nullptr, 0u, s_stack, overflow, tc); if (ANYHIT && !TWO) walkSteps++; }
      // positions from a wave prefix sum over the per-lane pair counts, then every lane writes its own pairs
      const uint32_t cntL = (uint32_t)__popc(Gt.y);
      const uint32_t scan = wave_scan_inclusive(cntL);
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)scan, 63);
      if (total != 0u) {
        const uint32_t tag = key << TRI_ID_BITS;
        if ((ringTail - ringHead) + total <= 128u) {
          uint32_t pos = ringTail + scan - cntL;
          if (cntL) lastEnd = pos + cntL;
          while (Gt.y) {
            const uint32_t k = (uint32_t)__ffs((int)Gt.y) - 1u;
            Gt.y &= Gt.y - 1u;
            wt_queue_put(W, pos & 127u, tag | (Gt.x + k));
            pos++;
          }
          ringTail += total;
          while (ringTail - ringHead >= 64u) batch(64u);
        } else for (;;) { // (more pairs than the ring has room for) one ballot round per triangle; a batch as soon as 64 pairs are pending (<= 63 + 64 <= the ring's 128 entries)
          const unsigned long long m = __ballot(Gt.y != 0u);
          if (!m) break;
          const bool push = Gt.y != 0u;
          if (push) {
            const uint32_t k = (uint32_t)__ffs((int)Gt.y) - 1u;
            Gt.y &= Gt.y - 1u;
            wt_queue_put(W, (ringTail + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & 127u, tag | (Gt.x + k));
          }
          ringTail += (uint32_t)__popcll(m);
          if (push) lastEnd = ringTail;
          if (ringTail - ringHead >= 64u) batch(64u);
        }
      }
      // the walk moves on before the ring is looked at (a closest-hit walk's pop does not depend on tBest)
      if (!ANYHIT && walking && trav_pop<STACK, OVERFLOW>(R, s_stack, overflow, HELP ? base : 0u)) draining = true;
      if (ringTail != ringHead) {
        const unsigned long long blocked = __ballot(alive && draining && (int)(ringHead - lastEnd) < 0);
        // (a wave with few walks left -- the tail of a launch, or all of a thin launch -- fills the ring slowly: waiting for 64 pairs there only delays the
        // distance its walks cull with, and with it the launch's last ray; r05d: C3 trace -3 %, shadow -8 %, a spp-1 frame -20 %)
        const uint32_t walkers = (uint32_t)__popcll(__ballot(alive && !draining));
        if ((uint32_t)__popcll(blocked) >= DYN_FLUSH_AT || walkers <= DYN_THIN_WALKERS) batch(ringTail - ringHead);
      }
      // every ray picks up what the batches of this step found
      if (alive) {
        if (!ANYHIT) R.tBest = u2f(wt_best_t(W, key));
        else if (!draining && (wt_best_id(W, key) != 0u || trav_pop<STACK, OVERFLOW>(R, s_stack, overflow, HELP ? base : 0u))) draining = true; // a shadow walk ends at the first hit
        done = draining && (int)(ringHead - lastEnd) >= 0;
      }
    }
    if (alive && done) {
      if (HELP && helper) { alive = false; wt_helpers_add(W, key, 0xffffffffu); } // this part of the ray is done; the owner writes the result
      else {
        const uint4 h = wt_hit_get(W, lane); // (u, v, triangle | class << 28) or (origin.xy, MISS); .w: helpers still walking parts of this ray
        if (!HELP || h.w == 0u) { // the ray's result, written in place over its record
          alive = false;
          if (!ANYHIT) {
            // ONE 16-byte store per finished ray: the batches left the finished record in LDS (material class in the top four bits of the triangle word; until r04 the
            // material word went into b.w as a second store into another line: C3 85 GB of write traffic per frame for 26 GB of results)
            uint32_t word = h.z;
            if constexpr (TWO) { if (word != MISS) word = sc.flatOfOrig[word & 0x0fffffffu] | (word & 0xf0000000u); } // scene-order id -> index of the hit's TriRec
            st4(&qs.a[qIn][rec], R.tBest, u2f(h.x), u2f(h.y), u2f(word));
            if (sc.mediumStackSize && word == MISS) { V3 wo = R.o; if constexpr (TWO) wo = R.wo; reinterpret_cast<float*>(&qs.b[qIn][rec])[3] = wo.z; }
          } else {
            const bool found = wt_best_id(W, lane) != 0u;
            const uint32_t slot = qs.slot[qIn][rec];
            F4 nc = F4{0.0f, 0.0f, 0.0f, 0.0f}; // (neeContrib, 1 = emitted at bounce 0)
            if (!found || st.neeKey) nc = ld4(&qs.c[qIn][rec]);
            if (!found) {
              Slot* S = &st.slots[slot];
              F4 rr = ld4(&S->rad);
              st4(&S->rad, rr.x + nc.x, rr.y + nc.y, rr.z + nc.z, rr.w);
            }
            if (st.neeKey && nc.w != 0.0f) nee_aov_record(st, slot, found);
          }
        }
      }
    }
  }
  if (ANYHIT && !TWO) { // node visits of this wave's shadow walks, for the host's choice of their order (one atomic per wave, on one of 16 lines)
    uint32_t n = walkSteps;
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
    if (lane == 0 && n) atomicAdd(&cnt->shadowOrderSteps[slotOrder ? 1 : 0][(blockIdx.x * (TRACE_BLOCK / 64u) + (threadIdx.x >> 6)) & 15u].v, n);
  }
  if (COUNT) { // measurement builds only: one atomic pair per wave
    unsigned long long a = tc.nodes, b = tc.tris;
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    if (lane == 0) { atomicAdd(ANYHIT ? &cnt->shadowNodesVisited : &cnt->nodesVisited, a); atomicAdd(ANYHIT ? &cnt->shadowTrisTested : &cnt->trisTested, b); }
    if (!ANYHIT && !TWO && lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&cnt->dynStats[k], ds[k]);
  }
}

#undef key
// what every wave of a k_trace_dyn launch does first: the ray counts of the queue's shards (lane k keeps shard k's; read back with readlane where a range is entered),
// the launch's total (one writer adds it to the frame's statistics), and the rays per cursor atomic -- DYN_CLAIM while every wave of the launch can have a claim of its own;
// below that 64, so that the rays spread over twice as many waves: a launch lasts as long as its slowest wave, and in the thin launches of a low-spp frame
// (hdGatling's default is ONE sample per pixel and call) that is all it lasts
template <bool ANYHIT>
__device__ __forceinline__ bool trace_dyn_prologue(const QueueSet& qs, Counters* cnt, uint32_t qIn, uint32_t refill, uint32_t& shardCount, uint32_t& claim)
{
  const uint32_t lane = __lane_id(), cap = qs.cap;
  shardCount = 0u;
  if (lane < NSHARD) { const uint32_t c = cnt->count[qIn][lane].v; shardCount = c < cap ? c : cap; } // (clamped: see Counters::overflow)
  uint32_t nRays = shardCount;
  for (int off = 4; off > 0; off >>= 1) nRays += __shfl_down(nRays, off);
  nRays = uni(nRays);
  if (blockIdx.x == 0 && threadIdx.x == 0) { if (ANYHIT) { cnt->shadowRays += nRays; cnt->shadowOrderRays[(refill & DYN_SLOT_ORDER) ? 1 : 0] += nRays; } else cnt->segments += nRays; } // single writer per launch
  claim = nRays >= gridDim.x * (TRACE_BLOCK / 64u) * DYN_CLAIM ? DYN_CLAIM : 64u;
  // Waves the launch has no chunk for leave at once, without touching the cursors: every wave that stays walks all NSHARD cursors before it gives up, and a device-scope
  // atomic on one line completes ~88 times per microsecond -- 8 192 waves x 8 cursors were a 0.1 ms floor under every launch that held any ray at all, i.e. under each of
  // the 13 thin launches of a spp-1 frame (r05: C4 233 rays, 0.146 ms).  ceil(n / 64) chunks + one ragged chunk per shard; the waves that stay claim until every shard is dry.
  return (blockIdx.x * (TRACE_BLOCK / 64u) + (threadIdx.x >> 6)) < (nRays + 63u) / 64u + NSHARD;
}
template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool CUTOUT, bool SLOT = false>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(DYN_WAVES, 8))) void k_trace_dyn(SceneView sc, PathState st, QueueSet qs, Counters* cnt, uint32_t qIn, uint32_t refill)
{
  __shared__ WaveTri s_wave[TRACE_BLOCK / 64];
  uint32_t shardCount, claim;
  if (!trace_dyn_prologue<ANYHIT>(qs, cnt, qIn, refill, shardCount, claim)) return;
  // a THIN launch -- fewer rays than two chunks per wave -- lasts as long as its slowest ray, not as its throughput allows: its idle lanes help (trace_dyn_body)
  if (claim == 64u) trace_dyn_body<ANYHIT, COUNT, STACK, OVERFLOW, CUTOUT, false, true, SLOT>(sc, st, qs, cnt, qIn, refill, s_wave[threadIdx.x >> 6], shardCount, claim);
  else trace_dyn_body<ANYHIT, COUNT, STACK, OVERFLOW, CUTOUT, false, false, SLOT>(sc, st, qs, cnt, qIn, refill, s_wave[threadIdx.x >> 6], shardCount, claim);
}
// the two-level layout (wave_step2): 16 LDS stack entries, world + object-space ray in registers
template <bool ANYHIT, bool COUNT, bool CUTOUT>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_trace_dyn2(SceneView sc, PathState st, QueueSet qs, Counters* cnt, uint32_t qIn, uint32_t refill)
{
  __shared__ WaveTri s_wave[TRACE_BLOCK / 64];
  uint32_t shardCount, claim;
  if (!trace_dyn_prologue<ANYHIT>(qs, cnt, qIn, refill, shardCount, claim)) return;
  trace_dyn_body<ANYHIT, COUNT, 16, false, CUTOUT, true, false>(sc, st, qs, cnt, qIn, refill, s_wave[threadIdx.x >> 6], shardCount, claim);
}

// k_route: sorts k_trace_dyn's in-place results by outcome and material class (same routing as k_trace's epilogue).  A streaming pass whose trip is two barriers and
// one atomic round trip: ROUTE_ITEMS results per thread and trip (independent loads in flight, a quarter of the trips).
constexpr int ROUTE_ITEMS = 4; static_assert(ROUTE_ITEMS <= (int)APPEND_ITEMS_MAX, "shardCapacity's slack"); // (1 -> 4: trace + route stage -3.5 % on C4 / C5, -0.5 % on C3, r05w)
__global__ __launch_bounds__(BLOCK) void k_route(SceneView sc, PathState st, QueueSet qs, Counters* cnt, uint32_t qIn, uint32_t qMiss, FrameUniforms U, F4* __restrict__ sampleBuf)
{
  constexpr uint32_t NQ = 1 + MAT_CLASS_COUNT, NONE = NQ;
  __shared__ AppendScratch<NQ> sh;
  QueueReader rd; reader_init(rd, cnt, qIn, qs.cap);
  const uint32_t n = rd.pre[NSHARD];
  if (blockIdx.x == 0 && (U.flags & FLAG_BOUNDS_RETIRE) && !(U.flags & FLAG_TWO_STREAM)) zero_consumed_regen(cnt, qIn - Q_TRACE_A); // (k_raygen no longer zeroes it: it appends to it; two streams: k_zero_closest does)
  const uint32_t stride = gridDim.x * BLOCK * ROUTE_ITEMS;
  uint32_t qid[NQ]; qid[0] = qMiss;
#pragma unroll
  for (uint32_t c = 0; c < MAT_CLASS_COUNT; c++) qid[1 + c] = Q_HIT + c;
  uint32_t trip = 0;
  for (uint32_t base = blockIdx.x * BLOCK * ROUTE_ITEMS; base < n; base += stride, trip++) {
    uint32_t which[ROUTE_ITEMS], entry[ROUTE_ITEMS], idx[ROUTE_ITEMS]; F4 rdir[ROUTE_ITEMS]; uint32_t slotOf[ROUTE_ITEMS];
    uint32_t rec[ROUTE_ITEMS], sw[ROUTE_ITEMS]; F4 h[ROUTE_ITEMS];
#pragma unroll
    for (int k = 0; k < ROUTE_ITEMS; k++) { // the loads of all items first: independent requests in flight
      const uint32_t i = base + (uint32_t)k * BLOCK + threadIdx.x;
      rec[k] = 0u; sw[k] = 0u; h[k] = F4{0.0f, 0.0f, 0.0f, u2f(MISS)};
      if (i < n) { rec[k] = reader_index(rd, i); sw[k] = qs.slot[qIn][rec[k]]; h[k] = ld4(&qs.a[qIn][rec[k]]); }
    }
#pragma unroll
    for (int k = 0; k < ROUTE_ITEMS; k++) {
      const uint32_t i = base + (uint32_t)k * BLOCK + threadIdx.x;
      which[k] = NONE; entry[k] = 0u; rdir[k] = F4{0.0f, 0.0f, 0.0f, 0.0f}; slotOf[k] = 0u;
      if (i < n) {
        const uint32_t r = rec[k];
        uint32_t slot = sw[k];
        const bool fresh = (slot & TRACE_FRESH) != 0u; slot &= ~TRACE_FRESH; // camera ray of a path whose Slot is still unwritten (FLAG_DEFER_SLOT)
        bool hit = f2u(h[k].w) != MISS, miss = !hit, volMiss = false, retired = false, freshHit = false;
        uint32_t klass = 0u;
        // the direction is needed at a miss by the dome lookup only (hits stay in place: k_shade gathers them)
        if (miss && sc.domeTexture != 0u) rdir[k] = ld4(&qs.b[qIn][r]);
        if (hit) klass = f2u(h[k].w) >> 28; // k_trace_dyn's result word: triangle index | material class << 28
        if (klass == SHADE_CLASS_OPBR_BASE && (U.flags & FLAG_MERGE_SHADE_VARIANTS)) klass = 2u; // thin batches: one OpenPBR launch (same bits: gi_shading.h "BASE variant")
        if (fresh) { // k_shade begins the path (hit); a miss that needs the slot (dome image / medium stack) begins it here, any other retires the sample without a Slot
          if (hit) freshHit = true;
          else {
            const FreshRec f = qs.fresh[qIn - Q_TRACE_A][r];
            if (sc.domeTexture != 0u || sc.mediumStackSize != 0u) begin_fresh_path(U, st, slot, f);
            else { retire_fresh_miss(U, f, sampleBuf); retired = true; }
          }
        }
        if (miss && sc.mediumStackSize) { // the segment ended inside a medium: scattering event for k_shade<2> (rp_main.miss:57-66)
          volMiss = payload_medium_idx(f2u(st.slots[slot].thr.w), sc.mediumStackSize < MAX_MEDIUM_STACK ? sc.mediumStackSize : MAX_MEDIUM_STACK) > 0u;
          if (volMiss) { miss = false; klass = 2u; }
        }
        slotOf[k] = slot;
        if (hit || volMiss) { which[k] = 1u + klass; entry[k] = r | (freshHit ? HIT_FRESH : 0u) | (volMiss ? HIT_VOLUME : 0u); } // 4 bytes per hit: the record stays where it is
        else { which[k] = 0u; entry[k] = sc.domeTexture ? slot : (slot | (retired ? REGEN_FRESH : REGEN_MISSED)); }
      }
    }
    block_append_items<NQ, ROUTE_ITEMS>(sh, trip, which, qid, qs.cap, cnt, idx);
#pragma unroll
    for (int k = 0; k < ROUTE_ITEMS; k++) {
      if (which[k] == 0u) { if (sc.domeTexture) dome_miss(sc, st, slotOf[k], v3(rdir[k].x, rdir[k].y, rdir[k].z)); qs.slot[qMiss][idx[k]] = entry[k]; }
      else if (which[k] < NQ) qs.slot[Q_HIT + which[k] - 1u][idx[k]] = entry[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_shade: closest-hit shading + the post-trace part of the bounce loop, over the HIT queue only
// (rp_main.chit:132-493, rp_main.rgen:397-480).  Misses never get here (k_trace routes them to k_raygen).
// ------------------------------------------------------------------------------------------------
// minimum resident waves per SIMD asked of the register allocator for the plain OpenPBR variants: without NEE 4 (128 VGPRs, 3 spilled: the natural 3 waves measured
// slower, r04j); with NEE 1, i.e. what its 164 VGPRs allow -- 3 (squeezed to 4 waves it spills 14 and is slower, r04c)
constexpr int SHADE_OPENPBR_PLAIN_WAVES = 4, SHADE_OPENPBR_NEE_WAVES = 1;
#ifndef GI_SHADE_BASE_WAVES      // experiment knobs (tools/build_variant.py): minimum waves per SIMD asked for the OpenPBR BASE variant, without / with NEE
#define GI_SHADE_BASE_WAVES 1
#endif
#ifndef GI_SHADE_BASE_NEE_WAVES
#define GI_SHADE_BASE_NEE_WAVES 1
#endif
template <uint32_t KLASS, bool TEXTURED, bool VOLUME, bool NEE, bool PACKED>
// (forcing the plain variant to 5 waves/SIMD -- amdgpu_waves_per_eu((...) ? 5 : 1, 8): 90 VGPRs, no spills -- is SLOWER: C2 shade 178 -> 190 ms;
// the stage is bound by the memory pipeline's scattered 16-byte requests, not by latency hiding)
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((KLASS == 2u && !TEXTURED && !VOLUME) ? (NEE ? SHADE_OPENPBR_NEE_WAVES : SHADE_OPENPBR_PLAIN_WAVES) : (KLASS == SHADE_CLASS_OPBR_BASE ? (NEE ? GI_SHADE_BASE_NEE_WAVES : GI_SHADE_BASE_WAVES) : 1), 8))) void k_shade(FrameUniforms U, SceneView sc, PathState st, QueueSet qs, Counters* cnt, uint32_t par, uint32_t hitClass /* the HIT queue read: KLASS, or a variant's whose hits this launch shades with the full kernel */)
{
  __shared__ AppendScratch<3> sh;
  const uint32_t qNext = Q_TRACE_A + (par ^ 1u), qRegen = Q_REGEN_A + (par ^ 1u), qHit = Q_HIT + hitClass;
  QueueReader rdr; reader_init(rdr, cnt, qHit, qs.cap);
  const uint32_t n = rdr.pre[NSHARD];
  const uint32_t stride = gridDim.x * BLOCK;
  uint32_t trip = 0;
  for (uint32_t base = blockIdx.x * BLOCK; base < n; base += stride, trip++) {
    const uint32_t i = base + threadIdx.x;
    bool cont = false, ended = false, shadow = false, shadowFirst = false; uint32_t slot = 0, rngShadow = 0u;
    V3 no = v3(0.0f, 0.0f, 0.0f), k2 = no, sdir = no, nee = no; float ld = 0.0f, tMaxNext = GI_FLT_MAX;
    if (i < n) {
      const uint32_t e = qs.slot[qHit][reader_index(rdr, i)];
      const uint32_t ri = e & HIT_INDEX_MASK, qT = Q_TRACE_A + par; // the ray's record in the queue it was traced from (untouched until the next iteration's producers)
      const bool fresh = (e & HIT_FRESH) != 0u;
      slot = qs.slot[qT][ri] & ~TRACE_FRESH;
      F4 h = ld4(&qs.a[qT][ri]);
      const F4 rd = ld4(&qs.b[qT][ri]);
      h.w = (e & HIT_VOLUME) ? u2f(VOLUME_MISS) : u2f(f2u(h.w) & 0x0fffffffu); // strip the class bits / mark the scattering event for shade_segment
      Slot* S = &st.slots[slot];
      F4 tb = F4{1.0f, 1.0f, 1.0f, u2f(0u)}, rr = F4{0.0f, 0.0f, 0.0f, 0.0f}; // rp_main.rgen:274-276
      if (fresh) { // first hit of a deferred path: its rng / work item are beside the ray record; the Slot is completed here (work item now, throughput / radiance below)
        const FreshRec f = qs.fresh[par][ri];
        rr.w = u2f(f.rng);
        uint32_t pixelLocal, sLocal; work_item(U, f.work, pixelLocal, sLocal);
        st4(&S->id, u2f(pixelLocal), u2f(sLocal), u2f(1u), 0.0f);
      } else { tb = ld4(&S->thr); rr = ld4(&S->rad); }
      ShadeIO io; io.throughput = v3(tb.x, tb.y, tb.z); io.radiance = v3(rr.x, rr.y, rr.z); io.bitfield = f2u(tb.w); io.rng = f2u(rr.w);
      float* M = VOLUME ? st.media + (size_t)slot * st.mediaStride : nullptr; // this path's medium stack + walkSegmentPdf
      shade_segment<KLASS, TEXTURED, VOLUME, NEE, PACKED>(U, sc, M, h, rd, io);
      cont = io.cont; ended = !cont; shadow = io.shadow; shadowFirst = io.shadowFirst; no = io.no; k2 = io.k2; tMaxNext = io.tMaxNext;
      sdir = io.sdir; nee = io.nee; ld = io.ld; rngShadow = io.rngShadow;
      if (NEE && st.neeKey && shadowFirst && !shadow) nee_aov_record(st, slot, false); // NEE AOV (rp_main.rgen:431-435): an untraced shadow ray counts as "not shadowed"
      st4(&S->thr, io.throughput.x, io.throughput.y, io.throughput.z, u2f(io.bitfield));
      st4(&S->rad, io.radiance.x, io.radiance.y, io.radiance.z, u2f(io.rng));
    }
    const bool pred[3] = {cont, ended, shadow}; const uint32_t qid[3] = {qNext, qRegen, Q_SHADOW}; uint32_t idx[3];
    block_append<3>(sh, trip, pred, qid, qs.cap, cnt, idx);
    if (cont) {
      qs.slot[qNext][idx[0]] = slot;
      st4(&qs.a[qNext][idx[0]], no.x, no.y, no.z, 0.0f);
      st4(&qs.b[qNext][idx[0]], k2.x, k2.y, k2.z, tMaxNext);
    }
    if (ended) qs.slot[qRegen][idx[1]] = slot;
    if (shadow) {
      qs.slot[Q_SHADOW][idx[2]] = slot;
      st4(&qs.a[Q_SHADOW][idx[2]], no.x, no.y, no.z, ld);
      st4(&qs.b[Q_SHADOW][idx[2]], sdir.x, sdir.y, sdir.z, u2f(rngShadow)); // .w: rng state for the any-hit test of cutouts
      st4(&qs.c[Q_SHADOW][idx[2]], nee.x, nee.y, nee.z, shadowFirst ? 1.0f : 0.0f);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_aov: the non-colour AOVs (rp_main.rgen:132-183, 517-520; rp_main.chit:192-290).  They depend only on the primary hit
// of each sample and are overwritten sample after sample, so they are produced by a separate per-pixel pass that
// replays the camera rays of samples 0..spp-1 in order (same RNG streams) -- exact, and off the colour path's hot loop.
// ------------------------------------------------------------------------------------------------
__device__ inline V3 bsdf_albedo(const MaterialRec* m, const ShState& st, V3 k1)
{
  float nk1 = fmax2(dot(st.normal, k1), 1e-4f);
  if (m->klass == 0u) return v3(m->p[0], m->p[1], m->p[2]);
  if (m->klass == 1u) {
    UpsParams u = ups_params(m, st);
    float Fc = u.coat * (0.04f + 0.96f * schlick_w(nk1));
    V3 Fs = schlick3(u.F0, nk1);
    V3 diffuse = (u.albedo * (v3(1.0f, 1.0f, 1.0f) - Fs)) * (1.0f - Fc);
    V3 glossy = v3(Fc, Fc, Fc) + Fs * (1.0f - Fc);
    return diffuse + glossy;
  }
  OpbrParams o = opbr_params(m, st);
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
    diffuse = ((o.albedo * o.coatTint) * (v3(1.0f, 1.0f, 1.0f) - Fdf)) * (base * diel * (1.0f - o.tw));
    glossy = v3(Fc, Fc, Fc) + ((opbr_film_metal(o, nk1, schlick_f82(o.albedo, o.metalTint, nk1)) * o.specWeight) * o.coatTint) * (base * o.metalness)
             + ((o.specColor * o.coatTint) * Fdf) * (base * diel);
  }
  if (o.fuzzWeight > 0.0f) { // the fuzz layer keeps P = fuzz_weight * min(E, 1) of the light (tinted), what is beneath gets 1 - P
    const float Pf = o.fuzzWeight * fmin2(fuzz_albedo(nk1, o.fuzzAlpha), 1.0f);
    return (diffuse + glossy) * (1.0f - Pf) + o.fuzzColor * Pf;
  }
  return diffuse + glossy;
}

template <uint32_t STACK, bool OVERFLOW, bool PACKED>
__global__ __launch_bounds__(TRACE_BLOCK) void k_aov(FrameUniforms U, SceneView sc, AovTargets A, uint32_t ldsNodes, uint32_t ldsTris)
{
  extern __shared__ uint4 s_dyn[];
  uint2 (*s_stack)[TRACE_BLOCK] = reinterpret_cast<uint2 (*)[TRACE_BLOCK]>(s_dyn);
  uint4* s_nodes = s_dyn + (STACK * TRACE_BLOCK * sizeof(uint2)) / sizeof(uint4);
  uint4* s_tris = s_nodes + ldsNodes * 5u;
  for (uint32_t i = threadIdx.x; i < ldsNodes * 5u; i += TRACE_BLOCK) s_nodes[i] = reinterpret_cast<const uint4*>(sc.nodes)[i];
  for (uint32_t i = threadIdx.x; i < ldsTris * 3u; i += TRACE_BLOCK) s_tris[i] = reinterpret_cast<const uint4*>(sc.tris)[(i / 3u) * 4u + (i % 3u)];
  __syncthreads();
  const uint32_t p = blockIdx.x * TRACE_BLOCK + threadIdx.x;
  if (p >= U.pixelCount) return;
  const uint32_t pixelIndex = tile_to_image_pixel(U, p);
  auto put3 = [&](F4* buf, V3 v) { if (buf) { float* d = reinterpret_cast<float*>(&buf[pixelIndex]); d[0] = v.x; d[1] = v.y; d[2] = v.z; } };
  auto clr3 = [&](F4* buf, int id) { put3(buf, v3(A.clear[id][0], A.clear[id][1], A.clear[id][2])); };
  clr3(A.barycentrics, 3); clr3(A.texcoords, 4); clr3(A.opacity, 7); clr3(A.tangents, 8); clr3(A.bitangents, 9); clr3(A.thinWalled, 10);
  if (A.objectId) A.objectId[pixelIndex] = (int)f2u(A.clear[11][0]);
  if (A.depth) A.depth[pixelIndex] = A.clear[12][0];
  if (A.faceId) A.faceId[pixelIndex] = (int)f2u(A.clear[13][0]);
  if (A.instanceId) A.instanceId[pixelIndex] = (int)f2u(A.clear[14][0]);
  clr3(A.doubleSided, 15);
  V3 curNormal = v3(0.0f, 0.0f, 0.0f), curAlbedo = curNormal;
  if (U.sampleOffset == 0u) { clr3(A.normal, 1); clr3(A.albedo, 16); curNormal = v3(A.clear[1][0], A.clear[1][1], A.clear[1][2]); curAlbedo = v3(A.clear[16][0], A.clear[16][1], A.clear[16][2]); }
  else {
    if (A.normal) { const F4 q = ld4(&A.normal[pixelIndex]); curNormal = v3(q.x, q.y, q.z); }
    if (A.albedo) { const F4 q = ld4(&A.albedo[pixelIndex]); curAlbedo = v3(q.x, q.y, q.z); }
  }
  TraceCounters tc{0u, 0u};
  const bool blend = (U.flags & FLAG_PROGRESSIVE) && U.sampleOffset > 0u;
  for (uint32_t s = 0; s < U.spp; s++) {
    V3 origin, dir; float tMin, tMax; uint32_t rng;
    make_camera_ray(U, pixelIndex, U.sampleOffset + s, origin, dir, tMin, tMax, rng);
    float t, u, v; uint32_t tri;
    uint32_t matWord;
    if (!traverse<false, false, STACK, OVERFLOW, false, true>(sc, s_nodes, ldsNodes, s_tris, ldsTris, s_stack, origin, dir, tMin, tMax, t, u, v, tri, matWord, tc, rng)) continue;
    ShState ss;
    setup_shading_state<PACKED>(sc, tri, u, v, dir, ss);
    const uint4* tp = reinterpret_cast<const uint4*>(sc.tris) + (size_t)tri * 4u;
    const uint32_t instIdx = tp[2].z;
    if (A.opacity) {
      // rp_main.chit:199-205 writes (1,0,0) for materials without cutout transparency; for the others the any-hit shader has written
      // viridis(opacity) (white for 0) of its last candidate (rp_main.ahit:45-49) -- restated as the ACCEPTED primary hit's opacity
      V3 c = v3(1.0f, 0.0f, 0.0f);
      if (matWord & (1u << 28)) { const float op = cutout_opacity_at(sc, matWord, tri, u, v); c = (op == 0.0f) ? v3(1.0f, 1.0f, 1.0f) : gi_colormap_viridis(op); }
      put3(A.opacity, c);
    }
    put3(A.tangents, (ss.tangentU + v3(1.0f, 1.0f, 1.0f)) * 0.5f);
    put3(A.bitangents, (ss.tangentV + v3(1.0f, 1.0f, 1.0f)) * 0.5f);
    put3(A.barycentrics, v3(1.0f - u - v, u, v));
    if (A.texcoords) {
      const uint4 td = tp[3];
      const float bx = 1.0f - u - v;
      if (PACKED) { const TriShade& q = sc.triShade[td.x]; put3(A.texcoords, v3((bx * q.uv[0][0] + u * q.uv[1][0]) + v * q.uv[2][0], (bx * q.uv[0][1] + u * q.uv[1][1]) + v * q.uv[2][1], 0.0f)); }
      else {
      const FVertex* va = &sc.verts[td.x]; const FVertex* vb = &sc.verts[td.y]; const FVertex* vc = &sc.verts[td.z];
      put3(A.texcoords, v3((bx * va->u + u * vb->u) + v * vc->u, (bx * va->v + u * vb->v) + v * vc->v, 0.0f));
      }
    }
    { const MaterialRec* tm = &sc.materials[ss.material]; put3(A.thinWalled, (tm->klass == 2u && ((uint32_t)tm->p[MP_FEATURES] & MATF_THIN_WALLED) != 0u) ? v3(1.0f, 0.0f, 0.0f) : v3(0.0f, 1.0f, 0.0f)); } // rp_main.chit:218-220
    if (A.objectId) A.objectId[pixelIndex] = (int)sc.instances[instIdx].pad;
    if (A.depth) A.depth[pixelIndex] = 2.0f * gi_logf(t / U.clipNear) / gi_logf(U.clipFar / U.clipNear) - 1.0f;
    if (A.faceId) A.faceId[pixelIndex] = sc.triFaceId[tri];
    if (A.instanceId) A.instanceId[pixelIndex] = sc.instances[instIdx].instanceId;
    put3(A.doubleSided, (ss.meshFlags & 2u) ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f));
    if (A.normal) {
      const V3 pos = (ss.normal + v3(1.0f, 1.0f, 1.0f)) * 0.5f;
      const V3 prev = blend ? curNormal : pos;
      curNormal = (prev * U.sampleOffsetF + pos * U.sppF) * U.invTotalSampleCount;
    }
    if (A.albedo) {
      const MaterialRec* am = &sc.materials[ss.material];
      if (am->flags & MAT_FLAG_TEXTURED) resolve_material_textures(sc, am, dir, ss);
      const V3 al = bsdf_albedo(am, ss, -dir);
      const V3 prev = blend ? curAlbedo : al;
      curAlbedo = (prev * U.sampleOffsetF + al * U.sppF) * U.invTotalSampleCount;
    }
  }
  put3(A.albedo, curAlbedo);
  if (A.normal) { // rp_main.rgen:517-520
    const V3 n = curNormal * 2.0f - v3(1.0f, 1.0f, 1.0f);
    put3(A.normal, (normalize(n) + v3(1.0f, 1.0f, 1.0f)) * 0.5f);
  }
}

// ------------------------------------------------------------------------------------------------
// k_debug_bsdf: the closed-form BSDF entry points on explicit shading frames (device-side known-answer tests)
// ------------------------------------------------------------------------------------------------
__global__ void k_debug_bsdf(const MaterialRec* mat, uint32_t shadeClass, uint32_t count, const float* __restrict__ in, float* __restrict__ out)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float* p = in + 22 * (size_t)i; float* o = out + 15 * (size_t)i;
  ShState st; st.normal = v3(p); st.tangentU = v3(p + 3); st.tangentV = v3(p + 6); st.geomNormal = v3(p + 9);
  st.position = v3(0.0f, 0.0f, 0.0f); st.frontFace = (p[21] < 0.5f); st.meshFlags = 0u; st.material = 0u;
  st.u = 0.0f; st.v = 0.0f; st.texMask = 0u; st.ior1 = 0.0f; st.ior2 = 0.0f; st.thinWalled = mat->klass == 2u && ((uint32_t)mat->p[MP_FEATURES] & MATF_THIN_WALLED) != 0u; st.sssVolume = false; st.hasCoatFrame = false; st.mesh = 0u; st.prim = 0u; st.vi[0] = st.vi[1] = st.vi[2] = 0u; st.instanceId = 0; st.hu = st.hv = 0.0f;
  BsdfSample bs; BsdfEval ev;
  if (shadeClass == SHADE_CLASS_OPBR_BASE) { bsdf_sample<SHADE_CLASS_OPBR_BASE>(mat, st, v3(p + 12), p[18], p[19], p[20], bs); bsdf_evaluate<SHADE_CLASS_OPBR_BASE>(mat, st, v3(p + 12), v3(p + 15), ev); }
  else { bsdf_sample<KLASS_DYNAMIC>(mat, st, v3(p + 12), p[18], p[19], p[20], bs); bsdf_evaluate<KLASS_DYNAMIC>(mat, st, v3(p + 12), v3(p + 15), ev); }
  o[0] = bs.k2.x; o[1] = bs.k2.y; o[2] = bs.k2.z; o[3] = bs.overPdf.x; o[4] = bs.overPdf.y; o[5] = bs.overPdf.z; o[6] = bs.pdf; o[7] = (float)bs.event;
  o[8] = ev.diffuse.x; o[9] = ev.diffuse.y; o[10] = ev.diffuse.z; o[11] = ev.glossy.x; o[12] = ev.glossy.y; o[13] = ev.glossy.z; o[14] = ev.pdf;
}

// k_debug_tex: the MDL runtime's remaining texture entry points on explicit queries (same layout as the oracle's orc_tex_runtime)
__global__ void k_debug_tex(const float* texels, uint32_t w, uint32_t h, uint32_t d, uint32_t count, const float* __restrict__ queries, float* __restrict__ out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float* q = queries + 8 * (size_t)i;
  const TextureRec t2{texels, w, h}; const TextureRec3 t3{texels, w, h, d};
  const int kind = (int)q[0]; const bool valid = q[1] != 0.0f;
  F4 r = F4{0.0f, 0.0f, 0.0f, 0.0f};
  if (kind == 0) r = tex_texel_float4_2d(t2, valid, (int)q[2], (int)q[3]);
  else if (kind == 1) { int rw, rh; tex_resolution_2d(t2, valid, rw, rh); r = F4{(float)rw, (float)rh, 0.0f, 0.0f}; }
  else if (kind == 2) r = tex_lookup_float4_3d(t3, valid, q[2], q[3], q[4], (uint32_t)q[5], (uint32_t)q[6], (uint32_t)q[7]);
  else r = tex_texel_float4_3d(t3, valid, (int)q[2], (int)q[3], (int)q[4]);
  st4(reinterpret_cast<F4*>(out) + i, r.x, r.y, r.z, r.w);
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------------
void launchInit(hipStream_t s, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t n, bool resetStats)
{
  uint32_t blocks = (n + 255u) / 256u; if (blocks > 4096u) blocks = 4096u; if (blocks == 0u) blocks = 1u;
  hipLaunchKernelGGL(k_init, dim3(blocks), dim3(256), 0, s, st, qs, cnt, n, resetStats ? 1u : 0u);
}
void launchRaygen(hipStream_t s, uint32_t blocks, const FrameUniforms& U, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t par, F4* sampleBuf)
{
  hipLaunchKernelGGL(k_raygen, dim3(blocks), dim3(BLOCK), 0, s, U, st, qs, cnt, par, sampleBuf);
}
void launchAccumulate(hipStream_t s, const FrameUniforms& U, const F4* sampleBuf, F4* accum, F4* colorOut, bool firstBatch, bool lastBatch)
{
  hipLaunchKernelGGL(k_accumulate, dim3((U.pixelCount + BLOCK - 1u) / BLOCK), dim3(BLOCK), 0, s, U, sampleBuf, accum, colorOut, firstBatch ? 1u : 0u, lastBatch ? 1u : 0u);
}
static bool sceneFitsLds(const SceneView& sc) { return sc.nodeCount <= LDS_NODES && sc.triCount <= LDS_TRIS && sc.triCount > 0u; }
static uint32_t traceStackEntries(const SceneView& sc) { return (sc.bvhDepth <= 4u && sceneFitsLds(sc)) ? 4u : (sc.bvhDepth <= 8u ? 8u : 16u); }
uint32_t traceStaticLdsBytes() { return (uint32_t)(sizeof(WaveTri) * (TRACE_BLOCK / 64) + sizeof(AppendScratch<1 + MAT_CLASS_COUNT>)); }
void traceLdsLayout(const SceneView& sc, uint32_t& ldsNodes, uint32_t& ldsTris, uint32_t& bytes)
{
  ldsNodes = sc.nodeCount < LDS_NODES ? sc.nodeCount : LDS_NODES;
  ldsTris = sc.triCount <= LDS_TRIS ? sc.triCount : 0u;
  bytes = traceStackEntries(sc) * TRACE_BLOCK * (uint32_t)sizeof(uint2) + ldsNodes * 80u + ldsTris * 48u;
  // (+ the kernels' static LDS: WaveTri per wave and the append scratch, see traceStaticLdsBytes)
}
template <bool ANYHIT, bool COUNT, bool CUTOUT>
static void launchTraceVariant(hipStream_t s, uint32_t blocks, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t qIn, uint32_t qMiss,
                               uint32_t dynRefill, uint32_t routeBlocks, const FrameUniforms& U, F4* sampleBuf)
{
  uint32_t ln, lt, bytes; traceLdsLayout(sc, ln, lt, bytes);
  const bool allLds = ln == sc.nodeCount && lt == sc.triCount && sc.triCount > 0u; // the whole scene is staged in LDS
  if (!allLds && dynRefill) { // big scene: persistent waves with dynamic ray fetch, results routed by a streaming pass
    // persistent waves pay the scratch set-up once, so trees deeper than 8 levels may keep 8 entries in LDS (more
    // resident waves) and spill the rest (TRACE_DYN_SPILL8), or keep 16 in LDS
    const uint32_t refill = (dynRefill & 0xffu) | (ANYHIT ? (dynRefill & DYN_SLOT_ORDER) : 0u);
    if (sc.twoLevel) { // instanced scene: TLAS + shared per-mesh BLASes
      hipLaunchKernelGGL((k_trace_dyn2<ANYHIT, COUNT, CUTOUT>), dim3(blocks), dim3(TRACE_BLOCK), 16u * TRACE_BLOCK * (uint32_t)sizeof(uint2), s, sc, st, qs, cnt, qIn, refill);
      if (!ANYHIT) hipLaunchKernelGGL(k_route, dim3(routeBlocks), dim3(BLOCK), 0, s, sc, st, qs, cnt, qIn, qMiss, U, sampleBuf);
      return;
    }
    const bool spill8 = (dynRefill & TRACE_DYN_SPILL8) != 0u;
    // 8 entries (16 KB per block), 12 (24 KB: 5 blocks per CU still fit next to the 8 KB of WaveTri) or 16 (32 KB: 4 blocks -- one wave per SIMD fewer)
    const uint32_t entries = (sc.bvhDepth <= 8u || spill8) ? 8u : (sc.bvhDepth <= 12u ? 12u : 16u);
    const uint32_t stackBytes = entries * TRACE_BLOCK * (uint32_t)sizeof(uint2);
#define GI_LAUNCH_DYN(STACK, OVF) do { \
      if (ANYHIT && (refill & DYN_SLOT_ORDER)) hipLaunchKernelGGL((k_trace_dyn<ANYHIT, COUNT, STACK, OVF, CUTOUT, ANYHIT>), dim3(blocks), dim3(TRACE_BLOCK), stackBytes, s, sc, st, qs, cnt, qIn, refill); \
      else hipLaunchKernelGGL((k_trace_dyn<ANYHIT, COUNT, STACK, OVF, CUTOUT, false>), dim3(blocks), dim3(TRACE_BLOCK), stackBytes, s, sc, st, qs, cnt, qIn, refill); } while (0)
    if (sc.bvhDepth <= 8u) GI_LAUNCH_DYN(8, false);
    else if (spill8) GI_LAUNCH_DYN(8, true);
    else if (sc.bvhDepth <= 12u) GI_LAUNCH_DYN(12, false);
    else if (sc.bvhDepth <= 16u) GI_LAUNCH_DYN(16, false);
    else GI_LAUNCH_DYN(16, true);
#undef GI_LAUNCH_DYN
    if (!ANYHIT) hipLaunchKernelGGL(k_route, dim3(routeBlocks), dim3(BLOCK), 0, s, sc, st, qs, cnt, qIn, qMiss, U, sampleBuf);
    return;
  }
  const bool dome = !ANYHIT && (sc.domeTexture != 0u || sc.mediumStackSize != 0u); // misses need the slot: dome image lookup / scattering events
#define GI_LAUNCH_TRACE(STACK, OVF, LDS) do { \
    if (dome) hipLaunchKernelGGL((k_trace<ANYHIT, COUNT, STACK, OVF, LDS, CUTOUT, !ANYHIT>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, sc, st, qs, cnt, qIn, qMiss, ln, lt, U, sampleBuf); \
    else hipLaunchKernelGGL((k_trace<ANYHIT, COUNT, STACK, OVF, LDS, CUTOUT, false>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, sc, st, qs, cnt, qIn, qMiss, ln, lt, U, sampleBuf); } while (0)
  if (allLds && sc.bvhDepth <= 4u) GI_LAUNCH_TRACE(4, false, true);
  else if (allLds && sc.bvhDepth <= 8u) GI_LAUNCH_TRACE(8, false, true);
  else if (sc.bvhDepth <= 8u) GI_LAUNCH_TRACE(8, false, false);
  else if (sc.bvhDepth <= 16u) GI_LAUNCH_TRACE(16, false, false);
  else GI_LAUNCH_TRACE(16, true, false);
#undef GI_LAUNCH_TRACE
}
template <bool ANYHIT, bool COUNT>
static void launchTraceCutout(hipStream_t s, uint32_t blocks, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t qIn, uint32_t qMiss,
                              uint32_t dynRefill, uint32_t routeBlocks, const FrameUniforms& U, F4* sampleBuf)
{
  if (sc.hasCutouts) launchTraceVariant<ANYHIT, COUNT, true>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks, U, sampleBuf);
  else launchTraceVariant<ANYHIT, COUNT, false>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks, U, sampleBuf);
}
// U / sampleBuf: the frame's uniforms and per-sample colour buffer -- read only when the queue holds camera rays flagged TRACE_FRESH (FLAG_DEFER_SLOT)
void launchTrace(hipStream_t s, uint32_t blocks, bool anyHit, bool count, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt,
                 uint32_t qIn, uint32_t qMiss, uint32_t dynRefill, uint32_t routeBlocks, const FrameUniforms& U, F4* sampleBuf)
{
  if ((dynRefill & 0xffu) > 64u) dynRefill = (dynRefill & ~0xffu) | 64u;
  if (!anyHit) { if (count) launchTraceCutout<false, true>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks, U, sampleBuf); else launchTraceCutout<false, false>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks, U, sampleBuf); }
  else { if (count) launchTraceCutout<true, true>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks, U, sampleBuf); else launchTraceCutout<true, false>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks, U, sampleBuf); }
}
void launchAov(hipStream_t s, const FrameUniforms& U, const SceneView& sc, const AovTargets& A)
{
  uint32_t ln, lt, bytes; traceLdsLayout(sc, ln, lt, bytes);
  bytes = (sc.bvhDepth <= 8u ? 8u : 16u) * TRACE_BLOCK * (uint32_t)sizeof(uint2) + ln * 80u + lt * 48u; // k_aov has no 4-entry variant
  const uint32_t blocks = (U.pixelCount + TRACE_BLOCK - 1u) / TRACE_BLOCK;
#define GI_LAUNCH_AOV(P) do { \
  if (sc.bvhDepth <= 8u) hipLaunchKernelGGL((k_aov<8, false, P>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, U, sc, A, ln, lt); \
  else if (sc.bvhDepth <= 16u) hipLaunchKernelGGL((k_aov<16, false, P>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, U, sc, A, ln, lt); \
  else hipLaunchKernelGGL((k_aov<16, true, P>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, U, sc, A, ln, lt); } while (0)
  if (sc.shadePacked) GI_LAUNCH_AOV(true); else GI_LAUNCH_AOV(false);
#undef GI_LAUNCH_AOV
}
void launchShade(hipStream_t s, uint32_t blocks, uint32_t klass, bool textured, bool volume, const FrameUniforms& U, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t par)
{
  // `klass` is a SHADE class (gi_types.h): the HIT queue to read.  The OpenPBR BASE variant's kernel exists for untextured materials in renders without a medium stack;
  // its hits go through the full OpenPBR kernel otherwise (same bits: gi_shading.h "BASE variant")
  const uint32_t hitClass = klass;
  if (klass == SHADE_CLASS_OPBR_BASE && (textured || volume)) klass = 2u;
#define GI_LAUNCH_SHADE4(K, T, V, N) do { if (sc.shadePacked) hipLaunchKernelGGL((k_shade<K, T, V, N, true>), dim3(blocks), dim3(BLOCK), 0, s, U, sc, st, qs, cnt, par, hitClass); \
    else hipLaunchKernelGGL((k_shade<K, T, V, N, false>), dim3(blocks), dim3(BLOCK), 0, s, U, sc, st, qs, cnt, par, hitClass); } while (0)
#define GI_LAUNCH_SHADE3(K, T, V) do { if (nee) GI_LAUNCH_SHADE4(K, T, V, true); else GI_LAUNCH_SHADE4(K, T, V, false); } while (0)
#define GI_LAUNCH_SHADE(K) do { \
    if (volume) { if (textured) GI_LAUNCH_SHADE3(K, true, true); else GI_LAUNCH_SHADE3(K, false, true); } \
    else if (textured) GI_LAUNCH_SHADE3(K, true, false); \
    else GI_LAUNCH_SHADE3(K, false, false); } while (0)
  const bool nee = (U.flags & FLAG_NEE) != 0u;
  if (klass == 0u) GI_LAUNCH_SHADE(0u); else if (klass == 1u) GI_LAUNCH_SHADE(1u); else if (klass == 2u) GI_LAUNCH_SHADE(2u); else GI_LAUNCH_SHADE3(SHADE_CLASS_OPBR_BASE, false, false);
#undef GI_LAUNCH_SHADE4
#undef GI_LAUNCH_SHADE3
#undef GI_LAUNCH_SHADE
}

void launchZeroClosest(hipStream_t s, Counters* cnt, uint32_t par) { hipLaunchKernelGGL(k_zero_closest, dim3(1), dim3(64), 0, s, cnt, par); }
void launchSpin(hipStream_t s, unsigned long long ns) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ns, (uint32_t*)nullptr); }
void launchResolveNee(hipStream_t s, const FrameUniforms& U, const unsigned long long* key, F4* aov, uint32_t pixelCount)
{
  hipLaunchKernelGGL(k_resolve_nee, dim3((pixelCount + 255u) / 256u), dim3(256), 0, s, U, key, aov, pixelCount);
}

void launchDebugTex(hipStream_t s, const float* texels, uint32_t w, uint32_t h, uint32_t d, uint32_t count, const float* queries, float* out)
{
  hipLaunchKernelGGL(k_debug_tex, dim3((count + 63u) / 64u), dim3(64), 0, s, texels, w, h, d, count, queries, out);
}

void launchDebugBsdf(hipStream_t s, const MaterialRec* mat, uint32_t shadeClass, uint32_t count, const float* in, float* out)
{
  hipLaunchKernelGGL(k_debug_bsdf, dim3((count + 63u) / 64u), dim3(64), 0, s, mat, shadeClass, count, in, out);
}

} // namespace gi
