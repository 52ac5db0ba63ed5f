// gi_queues.h -- work queues of the wavefront loop: block-aggregated sharded appends, the reader side, record helpers.
// Included by gi_kernels.hip only (device code, namespace gi).
#pragma once

#include <hip/hip_runtime.h>

#include "gi_device_math.h"
#include "gi_types.h"

namespace gi {

// ------------------------------------------------------------------------------------------------
// Stream compaction.  wave64 ballot + popcount prefix inside a wave, LDS aggregation over the 4 waves of a block,
// ONE atomic per block, queue and loop trip -- on the block's own shard of the queue (see gi_types.h: NSHARD).
// All stage kernels run block-uniform loops so the two barriers per trip are legal.  Returns, per queue, the index
// at which the calling lane must write its record (valid where pred is set).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t BLOCK = 256;
constexpr uint32_t WAVES = BLOCK / 64;

template <int NQ>
struct AppendScratch { uint32_t wcount[2][NQ][WAVES]; uint32_t base[2][NQ]; };

template <int NQ>
__device__ __forceinline__ void block_append(AppendScratch<NQ>& sh, uint32_t trip, const bool (&pred)[NQ], const uint32_t (&qid)[NQ], uint32_t cap,
                                             Counters* cnt, uint32_t (&outIdx)[NQ])
{
  const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6, par = trip & 1u, shard = blockIdx.x % NSHARD;
  unsigned long long m[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    m[q] = __ballot(pred[q]);
    if (lane == 0) sh.wcount[par][q][wave] = (uint32_t)__popcll(m[q]);
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    const uint32_t q = threadIdx.x;
    uint32_t total = 0;
#pragma unroll
    for (uint32_t w = 0; w < WAVES; w++) total += sh.wcount[par][q][w];
    uint32_t b = total ? atomicAdd(&cnt->count[qid[q]][shard].v, total) : 0u;
    if (b + total > cap) { cnt->overflow = 1u; b = 0u; } // never write outside the shard; the host reports the render as failed
    sh.base[par][q] = b;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    uint32_t off = sh.base[par][q] + (uint32_t)__popcll(m[q] & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < wave; w++) off += sh.wcount[par][q][w];
    outIdx[q] = shard * cap + off;
  }
}

// The same with ITEMS records per thread and trip (k_route, whose trip is otherwise the two barriers and the atomic's round trip: 0.33 ms per launch on C4 for
// 8 GB of traffic): `which[k]` names the queue item k of this thread goes to (>= NQ:
// none).  One atomic per block, queue and trip as before, now for ITEMS * 256 records.
template <int NQ, int ITEMS>
__device__ __forceinline__ void block_append_items(AppendScratch<NQ>& sh, uint32_t trip, const uint32_t (&which)[ITEMS], const uint32_t (&qid)[NQ],
    uint32_t cap,
                                                   Counters* cnt, uint32_t (&outIdx)[ITEMS])
{
  const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6, par = trip & 1u, shard = blockIdx.x % NSHARD;
  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t within[ITEMS]; // position of item k among this wave's appends to its queue
#pragma unroll
  for (int k = 0; k < ITEMS; k++) within[k] = 0u;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    uint32_t total = 0u;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const unsigned long long m = __ballot(which[k] == (uint32_t)q);
      if (which[k] == (uint32_t)q) within[k] = total + (uint32_t)__popcll(m & below);
      total += (uint32_t)__popcll(m);
    }
    if (lane == 0) sh.wcount[par][q][wave] = total;
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    const uint32_t q = threadIdx.x;
    uint32_t total = 0;
#pragma unroll
    for (uint32_t w = 0; w < WAVES; w++) total += sh.wcount[par][q][w];
    uint32_t b = total ? atomicAdd(&cnt->count[qid[q]][shard].v, total) : 0u;
    if (b + total > cap) { cnt->overflow = 1u; b = 0u; } // never write outside the shard; the host reports the render as failed
    sh.base[par][q] = b;
  }
  __syncthreads();
  uint32_t start[NQ]; // where this wave's records of queue q begin
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    uint32_t off = sh.base[par][q];
    for (uint32_t w = 0; w < wave; w++) off += sh.wcount[par][q][w];
    start[q] = shard * cap + off;
  }
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    uint32_t s0 = 0u;
#pragma unroll
    for (int q = 0; q < NQ; q++) s0 = which[k] == (uint32_t)q ? start[q] : s0;
    outIdx[k] = s0 + within[k];
  }
}

// Reader side: a queue is the concatenation of its NSHARD segments; maps a flat index to the record index.
struct QueueReader { uint32_t pre[NSHARD + 1]; uint32_t cap; };
__device__ __forceinline__ void reader_init(QueueReader& r, const Counters* cnt, uint32_t q, uint32_t cap)
{
  r.pre[0] = 0;
#pragma unroll
  // (clamped: see Counters::overflow)
  for (uint32_t s = 0; s < NSHARD; s++) { const uint32_t c = cnt->count[q][s].v; r.pre[s + 1] = r.pre[s] + (c < cap ? c : cap); }
  r.cap = cap;
}
__device__ __forceinline__ uint32_t reader_index(const QueueReader& r, uint32_t i)
{
  uint32_t s = 0, p = 0;
#pragma unroll
  for (uint32_t k = 1; k < NSHARD; k++) { const bool ge = i >= r.pre[k]; s += ge ? 1u : 0u; p = ge ? r.pre[k] : p; }
  return s * r.cap + (i - p);
}

__device__ __forceinline__ F4 ld4(const F4* p) { float4 v = *reinterpret_cast<const float4*>(p); return F4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void st4(F4* p, float x, float y, float z, float w) { *reinterpret_cast<float4*>(p) = make_float4(x, y, z, w); }
// tile-local pixel (row-major over the tile's rows) -> pixel index in the full image (rp_main.rgen:195); the tile's rows are
// rowBegin, rowBegin + rowStride, ...
__device__ __forceinline__ uint32_t tile_to_image_pixel(const FrameUniforms& U, uint32_t pixelLocal)
{
  const uint32_t row = pixelLocal / U.imageWidth, x = pixelLocal - row * U.imageWidth;
  return (U.rowBegin + row * U.rowStride) * U.imageWidth + x;
}
// Work item w of a batch -> (tile-local pixel, sample of the batch).  Any bijection gives the same image (a sample's RNG stream is a function of its pixel and
// sample index alone, and k_accumulate sums a pixel's samples in sample order); it decides which rays sit next to each other in the queues.
//   sample-major (w = sample * P + pixel): a wave holds 64 adjacent pixels of one sample index;
//   pixel-major (FLAG_PIXEL_MAJOR, w = visit(pixel) * S + sample): a wave holds consecutive samples of ONE pixel -- camera rays that differ by the sub-pixel
//   jitter only, so its lanes ask for the same nodes and triangles (one request per distinct line) -- and pixels are visited in 8x8 blocks.
__device__ __forceinline__ uint32_t visit_to_pixel(const FrameUniforms& U, uint32_t q)
{
  const uint32_t W = U.imageWidth, rows = U.pixelCount / W;
  const uint32_t band = q / (8u * W), j = q - band * 8u * W;
  const uint32_t left = rows - band * 8u, r = left < 8u ? left : 8u; // rows of this band
  const uint32_t nFull = W >> 3, t = j / (8u * r);
  uint32_t x, y;
  if (t < nFull) { const uint32_t k = j - t * 8u * r; x = 8u * t + (k & 7u); y = k >> 3; }
  else { const uint32_t wr = W - 8u * nFull, k = j - nFull * 8u * r; y = k / wr; x = 8u * nFull + (k - y * wr); }
  return (band * 8u + y) * W + x;
}
__device__ __forceinline__ void work_item(const FrameUniforms& U, uint32_t w, uint32_t& pixelLocal, uint32_t& sLocal)
{
  if (U.flags & FLAG_PIXEL_MAJOR) { const uint32_t q = w / U.batchSamples; sLocal = w - q * U.batchSamples; pixelLocal = visit_to_pixel(U, q); }
  else { sLocal = w / U.pixelCount; pixelLocal = w - sLocal * U.pixelCount; }
}
// where sample `sLocal` of tile pixel `pixelLocal` lives in the per-sample colour buffer: next to the other samples of its pixel when the work order is
// pixel-major (the lanes of a wave finish consecutive samples of one pixel: one contiguous run), next to the same sample of the neighbouring pixels otherwise
__device__ __forceinline__ size_t sample_record(const FrameUniforms& U, uint32_t pixelLocal, uint32_t sLocal)
{
  return (U.flags & FLAG_PIXEL_MAJOR) ? (size_t)pixelLocal * U.batchSamples + sLocal : (size_t)sLocal * U.pixelCount + pixelLocal;
}
constexpr uint32_t MISS = 0xffffffffu;
// HIT queues (round 4): an entry is the INDEX of the ray's record in the TRACE queue it was traced from (k_trace_dyn / k_trace leave the result in place there:
// a = (t, u, v, triangle | class << 28) or (tMax, origin.xy, MISS), b = (direction, - | origin.z)) plus two flags; k_shade gathers the record.  Until then
// k_route copied 36 bytes per hit into the class queue.
constexpr uint32_t HIT_FRESH = 0x80000000u;    // first hit of a path whose Slot is still unwritten: FreshRec beside the ray record has its rng / work item
constexpr uint32_t HIT_VOLUME = 0x40000000u;   // not a hit: the segment ended inside a medium (scattering event, rp_main.miss:57-66)
constexpr uint32_t HIT_INDEX_MASK = 0x3fffffffu;
// flag on a TRACE-queue slot word (FLAG_DEFER_SLOT): a camera ray whose Slot is still unwritten -- QueueSet::fresh holds its rng / work item
constexpr uint32_t TRACE_FRESH = 0x80000000u;
constexpr uint32_t REGEN_MISSED = 0x80000000u; // flag on a regen-queue entry: the path left the scene (k_trace -> k_raygen)
// flag on a regen-queue entry written by k_init: the slot carries no sample yet and its memory is uninitialised -- k_raygen
// must not read it (this replaces a 64-byte write per slot in k_init: 4 GB and 2 ms per batch for the 64 Mi-slot pool)
constexpr uint32_t REGEN_FRESH = 0x40000000u;

// Zeroes the counters of the queues that the producers of iteration `it` will append to.  Called by one thread of
// k_raygen(it): none of these queues is read or appended by k_raygen(it) itself (it reads REGEN[it&1] and appends
// TRACE[it&1]), and their previous consumers finished in iteration it-1 (stream order).
// `zeroRegen` false (FLAG_BOUNDS_RETIRE): k_raygen(it) itself appends to REGEN[(it&1)^1], so that counter is zeroed one kernel earlier, by k_route(it-1)
// (zero_consumed_regen), which runs after its last reader k_raygen(it-1).
// FLAG_TWO_STREAM (`twoStream`): iteration `it` runs  k_zero_closest, [k_raygen if it == 0,] k_trace, k_route, <wait for the shadow launch of it-1>, [k_raygen
// if it > 0,] k_shade  on the main stream and its shadow launch on the second one.  k_raygen(it) then zeroes only what lies between it and the end of the
// iteration: TRACE[(it&1)^1] (appended by k_shade(it); its count was last read by k_trace / k_route(it-1)), the SHADOW queue's counter and the shadow cursors
// (read by the shadow launch of it-1, which k_raygen(it) has waited for; appended / used by k_shade(it) and its shadow launch).  What k_trace / k_route(it)
// append to or claim from is zeroed by k_zero_closest(it) in front of them (zero_closest_counters).
__device__ __forceinline__ void zero_next_counters(Counters* cnt, uint32_t par, bool zeroRegen, bool twoStream = false)
{
  const uint32_t t = threadIdx.x;
  if (t < NSHARD) {
    cnt->count[Q_TRACE_A + (par ^ 1u)][t].v = 0;
    if (zeroRegen && !twoStream) cnt->count[Q_REGEN_A + (par ^ 1u)][t].v = 0;
    if (!twoStream) for (uint32_t c = 0; c < MAT_CLASS_COUNT; c++) cnt->count[Q_HIT + c][t].v = 0;
    cnt->count[Q_SHADOW][t].v = 0;
  }
  if (t < 2u * NCURSOR && (!twoStream || t >= NCURSOR)) cnt->cursor[t / NCURSOR][t % NCURSOR].v = 0; // k_trace_dyn's ray cursors (closest, shadow)
}
// k_zero_closest(it) (FLAG_TWO_STREAM), in front of k_trace(it): the HIT queues (last read by k_shade(it-1), appended by k_route(it)), REGEN[(it&1)^1] (last
// read by k_raygen(it-1); appended by k_raygen(0)'s bounds retire, k_route(it) and k_shade(it)) and the closest-hit cursors (k_trace(it-1))
__device__ __forceinline__ void zero_closest_counters(Counters* cnt, uint32_t par)
{
  const uint32_t t = threadIdx.x;
  if (t < NSHARD) {
    cnt->count[Q_REGEN_A + (par ^ 1u)][t].v = 0;
    for (uint32_t c = 0; c < MAT_CLASS_COUNT; c++) cnt->count[Q_HIT + c][t].v = 0;
  }
  if (t < NCURSOR) cnt->cursor[0][t].v = 0;
}

// k_route(it), block 0: REGEN[it&1] was read by k_raygen(it) and is next appended by k_raygen(it+1) (bounds retire) and k_route(it+1)
__device__ __forceinline__ void zero_consumed_regen(Counters* cnt, uint32_t par)
{
  if (threadIdx.x < NSHARD) cnt->count[Q_REGEN_A + par][threadIdx.x].v = 0;
}

} // namespace gi
