// gi_kernels.hip -- the wavefront path tracer's stage kernels for gfx950 (CDNA4, wave64): this unit holds the STREAM kernels (k_init, k_raygen, k_route,
// k_accumulate, the counter and debug helpers); the traversal kernels are in
// gi_trace.hip, k_shade in gi_shade.hip, k_aov in gi_aov.hip (one unit until round 6).
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
    if (resetStats) { cnt->shadowOrderRays[0] = 0; cnt->shadowOrderRays[1] = 0;
        for (int k = 0; k < 16; k++) { cnt->shadowOrderSteps[0][k].v = 0; cnt->shadowOrderSteps[1][k].v = 0; }
                      cnt->segments = 0; cnt->shadowRays = 0; cnt->nodesVisited = 0; cnt->trisTested = 0; cnt->shadowNodesVisited = 0;
                          cnt->shadowTrisTested = 0;
                      for (int k = 0; k < 4; k++) { cnt->phaseCycles[k] = 0; cnt->phaseLanes[k] = 0; } cnt->phaseTrips = 0;
                          for (int k = 0; k < 8; k++) cnt->dynStats[k] = 0; }
  }
  for (; i < n; i += gridDim.x * blockDim.x) qs.slot[Q_REGEN_A][(i / per) * qs.cap + (i % per)] = i | REGEN_FRESH; // the slots themselves stay untouched
}

// ------------------------------------------------------------------------------------------------
// k_raygen: persistent-thread ray generation (rp_main.rgen:213-283), the per-sample finish (:483-498) and the miss
// term.  Entry i of the regen queue finishes its sample (if any) into the per-sample colour buffer and takes work item
// w = workBase + i, decoded by work_item (gi_queues.h): by default consecutive entries get consecutive samples of one pixel, pixels in 8x8 blocks.
// ------------------------------------------------------------------------------------------------
// FLAG_BOUNDS_RETIRE: can the ray reach the scene at all?  A slab test against the root node's bounds (host: padded beyond the dequantised child boxes),
// widened per ray by 30 x the rounding error of (plane - origin) * (1 / d) and decided only by comparisons that a NaN fails -- so it answers "misses" for no
// ray whose walk could accept a triangle (every triangle lies inside its leaf box, every
// leaf box inside the root's bounds; same contract as the node test).
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

// RAYGEN_ITEMS regen entries per thread and trip: the trip is otherwise two barriers and an atomic round trip around a chain of dependent loads (entry -> slot
// -> finished sample), and the entries of one thread are independent of each other.
// (1 -> 2: raygen stage -6 % on C3 / C4; 4: the same, r05x)
constexpr int RAYGEN_ITEMS = 2; static_assert(RAYGEN_ITEMS <= (int)APPEND_ITEMS_MAX, "shardCapacity's slack");
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
            // primary miss: no light sampled, "not shadowed" (rp_main.rgen:431-435)
            if (st.neeKey && (f2u(tb.w) & 0x00000fffu) == 0u) nee_aov_record(st, slot, false);
          }
          if (st.bouncesAov && U.batchFirstSample + f2u(id.y) == U.spp - 1u) { // Bounces AOV: the pixel's last sample (rp_main.rgen:483-486)
            // a path that left the scene was routed here straight from k_trace, before the loop's bounce++ (rp_main.rgen:480)
            const uint32_t bounces = ((f2u(S->thr.w) + ((entry & REGEN_MISSED) ? 1u : 0u)) & 0x00000fffu), maxB = U.maxBounces < 0x00000fffu
                ? U.maxBounces : 0x00000fffu;
            const V3 c = gi_colormap_inferno((float)bounces / (float)maxB);
            F4* dst = &st.bouncesAov[tile_to_image_pixel(U, f2u(id.x))];
            dst->x = c.x; dst->y = c.y; dst->z = c.z;
          }
          // integer sum: order-free
          if (st.pathSegments) atomicAdd(&st.pathSegments[f2u(id.x)], (f2u(S->thr.w) + ((entry & REGEN_MISSED) ? 1u : 0u)) & 0x00000fffu);
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
          // :274-276 (deferred: written when the first segment hits, route_fresh)
          if (!(U.flags & FLAG_DEFER_SLOT)) slot_begin_path(S, rng, pixelLocal, sLocal);
        }
      }
      // a camera ray that cannot reach the scene: what k_route does with a fresh miss (retire_fresh_miss; the segment is counted below), minus the 52-byte
      // record, the traversal step and the routing pass -- the slot goes straight to the next k_raygen
      bool again = false;
      if (boundsRetire && more && ray_misses_bounds(U, origin, dir, tMin, tMax)) { retire_fresh_miss(U, fresh, sampleBuf); more = false; again = true;
          nRetired++; }
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
// the progressive blend of rp_main.rgen:506-515.  One thread per pixel; reads are
// coalesced across pixels (sample-major buffer) or whole lines per thread (pixel-major).
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

// k_route: sorts k_trace_dyn's in-place results by outcome and material class (same routing as k_trace's epilogue).  A streaming pass whose trip is two
// barriers and one atomic round trip: ROUTE_ITEMS results per thread and trip (independent loads in flight, a quarter of the trips).
// (1 -> 4: trace + route stage -3.5 % on C4 / C5, -0.5 % on C3, r05w)
constexpr int ROUTE_ITEMS = 4; static_assert(ROUTE_ITEMS <= (int)APPEND_ITEMS_MAX, "shardCapacity's slack");
__global__ __launch_bounds__(BLOCK) void k_route(SceneView sc, PathState st, QueueSet qs, Counters* cnt, uint32_t qIn, uint32_t qMiss, FrameUniforms U,
    F4* __restrict__ sampleBuf)
{
  constexpr uint32_t NQ = 1 + MAT_CLASS_COUNT, NONE = NQ;
  __shared__ AppendScratch<NQ> sh;
  QueueReader rd; reader_init(rd, cnt, qIn, qs.cap);
  const uint32_t n = rd.pre[NSHARD];
  // (k_raygen no longer zeroes it: it appends to it; two streams: k_zero_closest does)
  if (blockIdx.x == 0 && (U.flags & FLAG_BOUNDS_RETIRE) && !(U.flags & FLAG_TWO_STREAM)) zero_consumed_regen(cnt, qIn - Q_TRACE_A);
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
        // thin batches: one OpenPBR launch (same bits: gi_shading.h "BASE variant")
        if (klass == SHADE_CLASS_OPBR_BASE && (U.flags & FLAG_MERGE_SHADE_VARIANTS)) klass = 2u;
        // k_shade begins the path (hit); a miss that needs the slot (dome image / medium stack) begins it here, any other retires the sample without a Slot
        if (fresh) {
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
        // 4 bytes per hit: the record stays where it is
        if (hit || volMiss) { which[k] = 1u + klass; entry[k] = r | (freshHit ? HIT_FRESH : 0u) | (volMiss ? HIT_VOLUME : 0u); }
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
  hipLaunchKernelGGL(k_accumulate, dim3((U.pixelCount + BLOCK - 1u) / BLOCK), dim3(BLOCK), 0, s, U, sampleBuf, accum, colorOut, firstBatch ? 1u : 0u, lastBatch
      ? 1u : 0u);
}
// k_route behind a k_trace_dyn launch (gi_trace.hip launchTrace)
void launchRoute(hipStream_t s, uint32_t blocks, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t qIn, uint32_t qMiss,
    const FrameUniforms& U,
                 F4* sampleBuf)
{
  hipLaunchKernelGGL(k_route, dim3(blocks), dim3(BLOCK), 0, s, sc, st, qs, cnt, qIn, qMiss, U, sampleBuf);
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

} // namespace gi
