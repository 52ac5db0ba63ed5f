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

#include <hip/hip_runtime.h>

#include "gi_device_math.h"
#include "gi_kernels.h"
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
    sh.base[par][q] = total ? atomicAdd(&cnt->count[qid[q]][shard].v, total) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    uint32_t off = sh.base[par][q] + (uint32_t)__popcll(m[q] & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < wave; w++) off += sh.wcount[par][q][w];
    outIdx[q] = shard * cap + off;
  }
}

// Reader side: a queue is the concatenation of its NSHARD segments; maps a flat index to the record index.
struct QueueReader { uint32_t pre[NSHARD + 1]; uint32_t cap; };
__device__ __forceinline__ void reader_init(QueueReader& r, const Counters* cnt, uint32_t q, uint32_t cap)
{
  r.pre[0] = 0;
#pragma unroll
  for (uint32_t s = 0; s < NSHARD; s++) r.pre[s + 1] = r.pre[s] + cnt->count[q][s].v;
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
constexpr uint32_t MISS = 0xffffffffu;
constexpr uint32_t REGEN_MISSED = 0x80000000u; // flag on a regen-queue entry: the path left the scene (k_trace -> k_raygen)

// Zeroes the counters of the queues that the producers of iteration `it` will append to.  Called by one thread of
// k_raygen(it): none of these queues is read or appended by k_raygen(it) itself (it reads REGEN[it&1] and appends
// TRACE[it&1]), and their previous consumers finished in iteration it-1 (stream order).
__device__ __forceinline__ void zero_next_counters(Counters* cnt, uint32_t par)
{
  const uint32_t t = threadIdx.x;
  if (t < NSHARD) {
    cnt->count[Q_TRACE_A + (par ^ 1u)][t].v = 0;
    cnt->count[Q_REGEN_A + (par ^ 1u)][t].v = 0;
    for (uint32_t c = 0; c < MAT_CLASS_COUNT; c++) cnt->count[Q_HIT + c][t].v = 0;
    cnt->count[Q_SHADOW][t].v = 0;
  }
  if (t < 2u) cnt->cursor[t].v = 0; // k_trace_dyn's ray cursors (closest, shadow)
}

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
    cnt->workBase[0].v = 0; cnt->workBase[1].v = 0; cnt->cursor[0].v = 0; cnt->cursor[1].v = 0;
    if (resetStats) { cnt->segments = 0; cnt->shadowRays = 0; cnt->nodesVisited = 0; cnt->trisTested = 0; cnt->shadowNodesVisited = 0; cnt->shadowTrisTested = 0; }
  }
  for (; i < n; i += gridDim.x * blockDim.x) {
    st4(&st.slots[i].id, 0.0f, 0.0f, 0.0f, 0.0f);
    qs.slot[Q_REGEN_A][(i / per) * qs.cap + (i % per)] = i;
  }
}

// Camera ray of (pixel, sample): RNG init, pixel jitter / filter importance sampling, thin lens, clip range
// (rp_main.rgen:215-288).  Returns the RNG state after the draws the reference makes here.
__device__ __forceinline__ void make_camera_ray(const FrameUniforms& U, uint32_t pixelIndex, uint32_t sampleIndex, V3& origin, V3& dir, float& tMin, float& tMax, uint32_t& rng)
{
  const uint32_t px = pixelIndex % U.imageWidth, py = pixelIndex / U.imageWidth;
  rng = gi_hash_init(pixelIndex * (sampleIndex + 1u)); // :223, common.glsl:121-124
  float r0 = gi_next1f(rng), r1 = gi_next1f(rng);     // :224 (always drawn)
  float sox = 0.5f, soy = 0.5f;
  if (U.flags & FLAG_JITTER) {
    if (U.flags & FLAG_FIS) { float gx, gy; gi_fis_gauss(r0, r1, gx, gy); sox = 0.5f + gx; soy = 0.5f + gy; }
    else { sox = r0; soy = r1; }
  }
  V3 camRight = v3(U.camRight), camUp = v3(U.camUp), camPos = v3(U.camPos);
  V3 P = (v3(U.L) + (camRight * ((float)px + sox)) * U.WX) + (camUp * ((float)py + soy)) * U.HY; // :239-242
  origin = camPos;
  dir = normalize(P - origin);
  if ((U.flags & FLAG_DOF) && U.lensRadius > 0.0f) { // :249-263
    float z0 = gi_next1f(rng), z1 = gi_next1f(rng);
    V3 focal = origin + dir * U.focusDistance;
    V3 ap = gi_sample_hemisphere(z0, z1);
    origin = origin + camRight * (ap.x * U.lensRadius);
    origin = origin + camUp * (ap.y * U.lensRadius);
    dir = normalize(focal - origin);
  }
  if (dir.x == 0.0f) dir.x += GI_FLT_MIN; // :271
  if (dir.y == 0.0f) dir.y += GI_FLT_MIN;
  if (dir.z == 0.0f) dir.z += GI_FLT_MIN;
  tMin = 0.0f; tMax = GI_FLT_MAX;
  if (U.flags & FLAG_CLIP) { // :287-288, 308-314 (bounce 0 only)
    float cosCone = fmax2(1e-5f, dot(dir, v3(U.camFwd)));
    tMin = U.clipNear / cosCone; tMax = U.clipFar / cosCone;
  }
}

// ------------------------------------------------------------------------------------------------
// k_raygen: persistent-thread ray generation (rp_main.rgen:213-283), the per-sample finish (:483-498) and the miss
// term.  Entry i of the regen queue finishes its sample (if any) into the per-sample colour buffer and takes work item
// workBase + i = (pixel w % P, sample w / P) -- consecutive entries get adjacent pixels of the same sample index.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_raygen(FrameUniforms U, PathState st, QueueSet qs, Counters* cnt, uint32_t par, F4* __restrict__ sampleBuf)
{
  __shared__ AppendScratch<1> sh;
  const uint32_t qIn = Q_REGEN_A + par, qOut = Q_TRACE_A + par;
  QueueReader rd; reader_init(rd, cnt, qIn, qs.cap);
  const uint32_t n = rd.pre[NSHARD];
  const uint32_t workBase = cnt->workBase[par].v;
  if (blockIdx.x == 0) {
    zero_next_counters(cnt, par);
    if (threadIdx.x == 0) { const uint32_t left = U.workTotal - workBase; cnt->workBase[par ^ 1u].v = workBase + (n < left ? n : left); }
  }
  const uint32_t stride = gridDim.x * BLOCK;
  uint32_t trip = 0;
  for (uint32_t base = blockIdx.x * BLOCK; base < n; base += stride, trip++) {
    const uint32_t i = base + threadIdx.x;
    bool more = false; uint32_t slot = 0;
    V3 origin = v3(0.0f, 0.0f, 0.0f), dir = origin; float tMin = 0.0f, tMax = GI_FLT_MAX;
    if (i < n) {
      const uint32_t entry = qs.slot[qIn][reader_index(rd, i)];
      slot = entry & ~REGEN_MISSED;
      Slot* S = &st.slots[slot];
      const F4 id = ld4(&S->id);
      if (f2u(id.z) != 0u) { // finish the sample that just terminated (:489-496) -> per-sample colour buffer
        F4 r = ld4(&S->rad);
        V3 rad = v3(r.x, r.y, r.z);
        if (entry & REGEN_MISSED) {
          // the path left the scene: uniform fallback dome == colour-AOV clear value (rp_main.miss:68-86, Gi.cpp:2184-2199,
          // 2232-2238).  k_trace routes misses straight here; nothing after the miss can change the sample any more.
          F4 tb = ld4(&S->thr);
          rad = rad + v3(tb.x, tb.y, tb.z) * v3(U.background);
        }
        if (st.bouncesAov && U.batchFirstSample + f2u(id.y) == U.spp - 1u) { // Bounces AOV: the pixel's last sample (rp_main.rgen:483-486)
          // a path that left the scene was routed here straight from k_trace, before the loop's bounce++ (rp_main.rgen:480)
          const uint32_t bounces = ((f2u(S->thr.w) + ((entry & REGEN_MISSED) ? 1u : 0u)) & 0x00000fffu), maxB = U.maxBounces < 0x00000fffu ? U.maxBounces : 0x00000fffu;
          const V3 c = gi_colormap_inferno((float)bounces / (float)maxB);
          F4* dst = &st.bouncesAov[U.rowBegin * U.imageWidth + f2u(id.x)];
          dst->x = c.x; dst->y = c.y; dst->z = c.z;
        }
        float mv = fmax2(rad.x, fmax2(rad.y, rad.z));
        if (mv > U.maxSampleValue) rad = rad * (U.maxSampleValue / mv);
        // one aligned 16-byte store: a 12-byte record straddles DRAM sectors and costs two read-modify-writes
        st4(&sampleBuf[(size_t)f2u(id.y) * U.pixelCount + f2u(id.x)], fmax2(0.0f, rad.x), fmax2(0.0f, rad.y), fmax2(0.0f, rad.z), 0.0f);
      }
      const uint32_t w = workBase + i; // < 2^32 by construction of the batches (host)
      more = (i < U.workTotal - workBase) && (w < U.workTotal);
      if (more) {
        const uint32_t pixelLocal = w % U.pixelCount, sLocal = w / U.pixelCount;
        const uint32_t pixelIndex = U.rowBegin * U.imageWidth + pixelLocal; // :195 (global index: RNG is tile independent)
        const uint32_t sampleIndex = U.sampleOffset + U.batchFirstSample + sLocal;
        uint32_t rng;
        make_camera_ray(U, pixelIndex, sampleIndex, origin, dir, tMin, tMax, rng);
        st4(&S->thr, 1.0f, 1.0f, 1.0f, u2f(0u)); // :274-276
        st4(&S->rad, 0.0f, 0.0f, 0.0f, u2f(rng));
        st4(&S->id, u2f(pixelLocal), u2f(sLocal), u2f(1u), 0.0f);
      }
    }
    const bool pred[1] = {more}; const uint32_t qid[1] = {qOut}; uint32_t idx[1];
    block_append<1>(sh, trip, pred, qid, qs.cap, cnt, idx);
    if (more) {
      qs.slot[qOut][idx[0]] = slot;
      st4(&qs.a[qOut][idx[0]], origin.x, origin.y, origin.z, tMin);
      st4(&qs.b[qOut][idx[0]], dir.x, dir.y, dir.z, tMax);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_accumulate: folds one batch of per-sample colours into the per-pixel running sum IN SAMPLE ORDER
// (pixel_color += sample_color * invSpp, rp_main.rgen:498) and, after the last batch, writes the colour AOV with
// the progressive blend of rp_main.rgen:506-515.  One thread per pixel; reads are coalesced across pixels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_accumulate(FrameUniforms U, const F4* __restrict__ sampleBuf, F4* __restrict__ accum, F4* __restrict__ colorOut,
                                                      uint32_t firstBatch, uint32_t lastBatch)
{
  const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= U.pixelCount) return;
  V3 pixelColor = v3(0.0f, 0.0f, 0.0f);
  if (!firstBatch) { const F4 a = ld4(&accum[p]); pixelColor = v3(a.x, a.y, a.z); }
  for (uint32_t s = 0; s < U.batchSamples; s++) {
    const F4 src = ld4(&sampleBuf[(size_t)s * U.pixelCount + p]);
    pixelColor = pixelColor + v3(src.x, src.y, src.z) * U.invSpp;
  }
  if (!lastBatch) { st4(&accum[p], pixelColor.x, pixelColor.y, pixelColor.z, 0.0f); return; }
  const uint32_t pixelIndex = U.rowBegin * U.imageWidth + p;
  V3 prev = pixelColor;
  if ((U.flags & FLAG_PROGRESSIVE) && U.sampleOffset > 0u) { const F4 q = ld4(&colorOut[pixelIndex]); prev = v3(q.x, q.y, q.z); }
  const V3 c = (prev * U.sampleOffsetF + pixelColor * U.sppF) * U.invTotalSampleCount;
  st4(&colorOut[pixelIndex], c.x, c.y, c.z, 1.0f);
}

// ------------------------------------------------------------------------------------------------
// k_trace: software traversal of the 8-wide quantised BVH, one ray per lane.
//   * persistent blocks stage the top of the tree (and, for small scenes, all triangles) into LDS once
//   * per-lane traversal stack: 8 entries in LDS + scratch overflow
//   * octant-ordered child visits (Ylitie et al. 2017), two-sided Moeller-Trumbore on 48-byte records
// Traversal contract: accept tMin < t < tBest; ties go to the lower scene-order triangle id.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t LDS_NODES = 384;  // upper bound: 30 KiB   (the launch stages min(nodeCount, LDS_NODES) nodes)
constexpr uint32_t LDS_TRIS = 128;   // upper bound: 6 KiB    (all triangles when the scene has <= LDS_TRIS, else none)
constexpr uint32_t OVF_STACK = 40;   // scratch overflow entries of the fallback variant (trees deeper than 16 levels)
constexpr uint32_t TRACE_BLOCK = 256;

struct TraceCounters { uint32_t nodes, tris; };

// STACK = per-lane stack entries kept in LDS.  The traversal pushes at most one entry per tree level, so the host
// picks STACK >= tree depth (8 or 16) and the scratch overflow (OVERFLOW) is compiled in only for deeper trees:
// a kernel that declares scratch pays for it on every wave launch even if it never spills.
// Any-hit randomness (rp_main.ahit:51-60), restated order-independently: stateless hash of the path's rng state and the
// candidate's scene-order triangle id (see oracle cutout_random); the state itself is not advanced.
__device__ __forceinline__ float cutout_random(uint32_t rng, uint32_t triId)
{
  uint32_t st = (rng ^ (triId * 0x9e3779b9u + 0x85ebca6bu)) * 747796405u + 2891336453u;
  uint32_t word = ((st >> ((st >> 28) + 4u)) ^ st) * 277803737u;
  return u2f(0x3f800000u | (((word >> 22) ^ word) >> 9)) - 1.0f;
}

// Per-lane traversal state.  A ray is advanced by trav_step() one "group" at a time (one internal node, then the
// triangles of its leaf children, then a pop) so that k_trace (one ray per lane until it finishes) and k_trace_dyn
// (lanes refill from the queue as they finish) share the same arithmetic.
struct RayTrav {
  V3 o, d; float idx, idy, idz, tMin, tBest; uint32_t octinv;
  uint32_t bestTri, bestOrig, bestMat; float bestU, bestV;
  uint2 G; uint32_t sp; bool found;
};

__device__ __forceinline__ void trav_init(RayTrav& R, V3 o, V3 d, float tMin, float tMax)
{
  R.o = o; R.d = d; R.tMin = tMin; R.tBest = tMax;
  // reciprocal direction for the slab tests only (guard against 0: boxes are padded, a huge finite value is safe)
  const float gx = (fabsf(d.x) < 1e-30f) ? (d.x < 0.0f ? -1e-30f : 1e-30f) : d.x;
  const float gy = (fabsf(d.y) < 1e-30f) ? (d.y < 0.0f ? -1e-30f : 1e-30f) : d.y;
  const float gz = (fabsf(d.z) < 1e-30f) ? (d.z < 0.0f ? -1e-30f : 1e-30f) : d.z;
  // v_rcp_f32 (1 ulp) instead of three IEEE divisions: the reciprocals only feed the box tests, whose far planes are widened by 1e-5
  R.idx = __builtin_amdgcn_rcpf(gx); R.idy = __builtin_amdgcn_rcpf(gy); R.idz = __builtin_amdgcn_rcpf(gz);
  R.octinv = ((d.x >= 0.0f ? 1u : 0u) | (d.y >= 0.0f ? 2u : 0u) | (d.z >= 0.0f ? 4u : 0u)) * 0x01010101u; // replicated into the 4 bytes (trav_node_test)
  R.bestTri = 0xffffffffu; R.bestOrig = 0xffffffffu; R.bestMat = 0u; R.bestU = 0.0f; R.bestV = 0.0f;
  R.G = make_uint2(0u, 0x80000000u); // virtual group holding only the root
  R.sp = 0u; R.found = false;
}

// Node half of a traversal step: takes the nearest unvisited child of the current node group (pushing the rest), tests
// the ray against that node's 8 quantised child boxes and returns the triangle group (base, mask) of its hit leaf
// children; R.G becomes the group of hit internal children.  Caller guarantees R.G has node bits.
// Node half of a traversal step, part 1: takes the nearest unvisited child of the current node group (pushing the rest)
// and returns its node index.  Caller guarantees R.G has node bits.
template <uint32_t STACK, bool OVERFLOW>
__device__ __forceinline__ uint32_t trav_node_pick(RayTrav& R, uint2 (*s_stack)[TRACE_BLOCK], uint2 (&overflow)[OVERFLOW ? OVF_STACK : 1])
{
  const uint32_t tid = threadIdx.x;
  uint2 G = R.G;
  uint32_t sp = R.sp;
  const uint32_t bit = 31u - (uint32_t)__clz((int)(G.y & 0xff000000u));
  G.y &= ~(1u << bit);
  if (G.y & 0xff000000u) {
    if (!OVERFLOW || sp < STACK) s_stack[sp < STACK ? sp : STACK - 1u][tid] = G; else overflow[sp - STACK] = G;
    sp++;
  }
  const uint32_t slot = (bit - 24u) ^ (R.octinv & 7u);
  const uint32_t rel = (uint32_t)__popc((G.y & 0xffu) & ((1u << slot) - 1u));
  R.sp = sp;
  return G.x + rel;
}

// Part 2: tests the ray against the node's 8 quantised child boxes and returns the triangle group (base, mask) of its hit
// leaf children; R.G becomes the group of hit internal children.  The box test is a conservative filter (explicit fma,
// far planes and tBest widened by 1e-5 relative), it never decides a result.  Written for the VALU: the two planes of an
// axis go through one packed fma (v_pk_fma_f32), the per-child meta bytes (slot index, child bits, octant flip of internal
// children) are decoded four at a time with byte-parallel integer ops, and empty slots (meta 0) contribute no bits, so
// the hit mask is assembled without a branch.
typedef float gi_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 trav_node_test(RayTrav& R, const uint4& n0, const uint4& n1, const uint4& n2, const uint4& n3, const uint4& n4)
{
  const V3 o = R.o, d = R.d;
  constexpr float WIDEN = 1.00001f;
  // ray in the node's quantisation frame: t(q) = q * a + b per axis; .x = near plane, .y = far plane (widened)
  const float sx = u2f((n0.w & 0xffu) << 23), sy = u2f(((n0.w >> 8) & 0xffu) << 23), sz = u2f(((n0.w >> 16) & 0xffu) << 23);
  const float ax = sx * R.idx, ay = sy * R.idy, az = sz * R.idz;
  const float bx = (u2f(n0.x) - o.x) * R.idx, by = (u2f(n0.y) - o.y) * R.idy, bz = (u2f(n0.z) - o.z) * R.idz;
  const gi_f2 Ax = {ax, ax * WIDEN}, Ay = {ay, ay * WIDEN}, Az = {az, az * WIDEN};
  const gi_f2 Bx = {bx, bx * WIDEN}, By = {by, by * WIDEN}, Bz = {bz, bz * WIDEN};
  const float tFar = R.tBest * WIDEN, tNear = R.tMin;
  // near/far plane bytes per axis, chosen by direction sign
  const bool nxn = d.x < 0.0f, nyn = d.y < 0.0f, nzn = d.z < 0.0f;
  const uint32_t qlox[2] = {n2.x, n2.y}, qloy[2] = {n2.z, n2.w}, qloz[2] = {n3.x, n3.y};
  const uint32_t qhix[2] = {n3.z, n3.w}, qhiy[2] = {n4.x, n4.y}, qhiz[2] = {n4.z, n4.w};
  const uint32_t metaw[2] = {n1.z, n1.w};
  const uint32_t oct4 = R.octinv;
  uint32_t hitmask = 0u;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const uint32_t nearx = nxn ? qhix[h] : qlox[h], farx = nxn ? qlox[h] : qhix[h];
    const uint32_t neary = nyn ? qhiy[h] : qloy[h], fary = nyn ? qloy[h] : qhiy[h];
    const uint32_t nearz = nzn ? qhiz[h] : qloz[h], farz = nzn ? qloz[h] : qhiz[h];
    // four meta bytes at once: bits 7-5 = child bits (1 = internal, unary count for leaves), bits 4-0 = slot index, where
    // internal children (index 24..31, i.e. bits 4 and 3 set) are flipped by the ray octant
    const uint32_t m4 = metaw[h];
    const uint32_t inner4 = ((m4 & (m4 << 1)) >> 4) & 0x01010101u;
    const uint32_t idx4 = (m4 ^ (oct4 & ((inner4 << 8) - inner4))) & 0x1f1f1f1fu; // (x << 8) - x == x * 0xff per byte, at full rate
    const uint32_t bits4 = (m4 >> 5) & 0x07070707u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t sh = 8u * (uint32_t)k;
      const gi_f2 qx = {(float)((nearx >> sh) & 0xffu), (float)((farx >> sh) & 0xffu)};
      const gi_f2 qy = {(float)((neary >> sh) & 0xffu), (float)((fary >> sh) & 0xffu)};
      const gi_f2 qz = {(float)((nearz >> sh) & 0xffu), (float)((farz >> sh) & 0xffu)};
      const gi_f2 tx = __builtin_elementwise_fma(qx, Ax, Bx), ty = __builtin_elementwise_fma(qy, Ay, By), tz = __builtin_elementwise_fma(qz, Az, Bz);
      const float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, tNear));
      const float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, tFar));
      const uint32_t contrib = ((bits4 >> sh) & 0xffu) << ((idx4 >> sh) & 0xffu);
      hitmask |= (tn <= tf) ? contrib : 0u;
    }
  }
  R.G = make_uint2(n1.x, (hitmask & 0xff000000u) | (n0.w >> 24));
  return make_uint2(n1.y, hitmask & 0x00ffffffu);
}

// The per-lane composition (each lane fetches its own node: from LDS when staged there, else from global memory)
template <bool COUNT, uint32_t STACK, bool OVERFLOW, bool ALL_LDS>
__device__ __forceinline__ uint2 trav_node(RayTrav& R, const SceneView& sc, const uint4* s_nodes, uint32_t ldsNodes,
                                           uint2 (*s_stack)[TRACE_BLOCK], uint2 (&overflow)[OVERFLOW ? OVF_STACK : 1], TraceCounters& tc)
{
  const uint32_t nodeIdx = trav_node_pick<STACK, OVERFLOW>(R, s_stack, overflow);
  uint4 n0, n1, n2, n3, n4;
  if (ALL_LDS || nodeIdx < ldsNodes) { const uint4* p = s_nodes + nodeIdx * 5u; n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3]; n4 = p[4]; }
  else { const uint4* p = reinterpret_cast<const uint4*>(sc.nodes) + (size_t)nodeIdx * sc.nodeStrideU4; n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3]; n4 = p[4]; }
  if (COUNT) tc.nodes++;
  return trav_node_test(R, n0, n1, n2, n3, n4);
}

// End of a step: when the current group has no unvisited internal child left, continue with the stack top.
// Returns true when the traversal is finished.
template <uint32_t STACK, bool OVERFLOW>
__device__ __forceinline__ bool trav_pop(RayTrav& R, uint2 (*s_stack)[TRACE_BLOCK], uint2 (&overflow)[OVERFLOW ? OVF_STACK : 1])
{
  if (R.G.y & 0xff000000u) return false;
  if (R.sp == 0u) return true;
  const uint32_t sp = --R.sp;
  R.G = (!OVERFLOW || sp < STACK) ? s_stack[sp < STACK ? sp : STACK - 1u][threadIdx.x] : overflow[sp - STACK];
  return false;
}

// Two-sided Moeller-Trumbore, operation order == oracle tri_test; evaluated branch-free (a wave almost always has a lane
// that passes each early-out, so predication is cheaper than exec-mask branches).  `inside` excludes the t < tBest test.
__device__ __forceinline__ bool tri_test(V3 o, V3 d, float tMin, const uint4& a, const uint4& b, const uint4& c, float& t, float& u, float& v)
{
  const V3 v0 = v3(u2f(a.x), u2f(a.y), u2f(a.z)), e1 = v3(u2f(a.w), u2f(b.x), u2f(b.y)), e2 = v3(u2f(b.z), u2f(b.w), u2f(c.x));
  const V3 pv = cross(d, e2);
  const float det = dot(e1, pv);
  const float inv = 1.0f / det;
  const V3 tv = o - v0;
  u = dot(tv, pv) * inv;
  const V3 qv = cross(tv, e1);
  v = dot(d, qv) * inv;
  t = dot(e2, qv) * inv;
  return (det != 0.0f) & (u >= 0.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (t > tMin);
}

// Advances the ray by one group (node, then its leaf triangles one after the other, then pop); returns true when the
// traversal is finished.  The per-lane form used by k_aov and the block-synchronous k_trace.
template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool ALL_LDS, bool CUTOUT>
__device__ __forceinline__ bool trav_step(RayTrav& R, const SceneView& sc, const uint4* s_nodes, uint32_t ldsNodes, const uint4* s_tris, uint32_t ldsTris,
                                          uint2 (*s_stack)[TRACE_BLOCK], uint2 (&overflow)[OVERFLOW ? OVF_STACK : 1], TraceCounters& tc, uint32_t rng)
{
  uint2 Gt;
  if (R.G.y & 0xff000000u) Gt = trav_node<COUNT, STACK, OVERFLOW, ALL_LDS>(R, sc, s_nodes, ldsNodes, s_stack, overflow, tc);
  else { Gt = R.G; R.G = make_uint2(0u, 0u); }
  // triangles of this node
  while (Gt.y) {
    const uint32_t k = (uint32_t)__ffs((int)Gt.y) - 1u;
    Gt.y &= Gt.y - 1u;
    const uint32_t triIdx = Gt.x + k;
    uint4 a, b, c;
    if (ALL_LDS || triIdx < ldsTris) { const uint4* p = s_tris + triIdx * 3u; a = p[0]; b = p[1]; c = p[2]; }
    else { const uint4* p = reinterpret_cast<const uint4*>(sc.tris) + (size_t)triIdx * 4u; a = p[0]; b = p[1]; c = p[2]; }
    if (COUNT) tc.tris++;
    const uint32_t orig = c.y;
    float t, u, v;
    const bool inside = tri_test(R.o, R.d, R.tMin, a, b, c, t, u, v);
    const bool better = (t < R.tBest) | ((t == R.tBest) & (R.bestOrig != 0xffffffffu) & (orig < R.bestOrig));
    bool accept = inside & better;
    if (CUTOUT && accept && (c.w & (1u << 28))) { // non-opaque material: stochastic cutout (ignoreIntersectionEXT, rp_main.ahit:57-60)
      const float opacity = sc.materials[c.w & 0x00ffffffu].p[MP_CUTOUT];
      accept = !(cutout_random(rng, orig) > opacity);
    }
    if (accept) {
      R.tBest = t; R.bestU = u; R.bestV = v; R.bestTri = triIdx; R.bestOrig = orig; R.bestMat = c.w; R.found = true;
      if (ANYHIT) { R.G.y = 0u; R.sp = 0u; break; }
    }
  }
  return trav_pop<STACK, OVERFLOW>(R, s_stack, overflow);
}

// ------------------------------------------------------------------------------------------------
// Wave-cooperative triangle stage.  The number of leaf triangles a node step yields varies from 0 to 24 per lane, so a
// per-lane triangle loop runs as long as the busiest lane while most lanes sit idle (measured: 28 % of the lanes active).
// Instead every lane appends its (ray lane, triangle) pairs to a per-wave LDS queue and the wave tests 64 pairs at a
// time, one per lane, fetching the owning lane's ray with ds_bpermute.  The nearest hit of a ray is kept in LDS as the
// 64-bit key (t bits << 32 | scene-order triangle id + 1) under atomicMin, which is exactly the oracle's
// "t < tBest, ties to the lower scene-order id" rule and makes the result independent of the test order.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t TRI_ID_BITS = 26; // queue entry = ray lane << 26 | triangle index (the host refuses scenes with >= 2^26 triangles)
struct WaveTri {
  unsigned long long best[64]; // per ray lane: (t bits << 32) | (scene-order id + 1); low word 0 = no hit yet
  uint4 hit[64];               // per ray lane: (triangle index, u bits, v bits, material word) of that hit
  uint32_t queue[128];         // ring of pending (ray lane, triangle) pairs
};
// Staging buffer of the cooperative fetch (scenes in global memory).  A lane that loads its own 80-byte node issues five
// 16-byte loads to a cache line no other lane touches, so every load instruction costs the L1 64 tag look-ups; measured,
// the texture-address unit was busy 63 % of k_trace's time.  Instead lane i of the wave loads 16-byte piece (i % 5) of
// the node that lane (i / 5) asked for: consecutive lanes read consecutive addresses, an instruction touches ~13-26 lines,
// and the pieces meet again in LDS (conflict-free: 80 B and 48 B lane strides both map 16 lanes onto all 64 banks).
struct WaveStage { uint4 buf[64 * 5]; };
// Measured on C3 (1M-triangle soup): 35.4 ms with the cooperative fetch vs 29.6 ms without (the 20 KiB of staging per
// block cost two resident blocks per CU and the extra LDS round trip outweighs the saved tag look-ups) -> off.
constexpr bool TRACE_DYN_COOP_FETCH = false;

template <bool COUNT, bool ALL_LDS, bool CUTOUT, bool COOP>
__device__ __forceinline__ void wave_tri_batch(WaveTri& W, WaveStage* S, uint32_t head, uint32_t cnt, const RayTrav& R, uint32_t rng, const SceneView& sc,
                                               const uint4* s_tris, uint32_t ldsTris, TraceCounters& tc)
{
  const uint32_t lane = __lane_id();
  const bool act = lane < cnt;
  const uint32_t e = act ? *(volatile uint32_t*)&W.queue[(head + lane) & 127u] : 0u;
  const uint32_t rl = e >> TRI_ID_BITS, triIdx = e & ((1u << TRI_ID_BITS) - 1u);
  // the owning lane's ray (executed by all lanes: wave-uniform control flow)
  const V3 o = v3(__shfl(R.o.x, (int)rl), __shfl(R.o.y, (int)rl), __shfl(R.o.z, (int)rl));
  const V3 d = v3(__shfl(R.d.x, (int)rl), __shfl(R.d.y, (int)rl), __shfl(R.d.z, (int)rl));
  const float tMin = __shfl(R.tMin, (int)rl);
  const uint32_t rrng = CUTOUT ? (uint32_t)__shfl((int)rng, (int)rl) : 0u;
  if (COOP) { // piece (flat % 3) of the triangle of entry (flat / 3), for flat = lane, 64 + lane, 128 + lane
    uint4 piece[3];
#pragma unroll
    for (uint32_t cidx = 0; cidx < 3u; cidx++) {
      const uint32_t flat = cidx * 64u + lane, owner = flat / 3u, part = flat - owner * 3u;
      const uint32_t tIdx = (uint32_t)__shfl((int)triIdx, (int)owner);
      piece[cidx] = make_uint4(0u, 0u, 0u, 0u);
      if (owner < cnt) piece[cidx] = reinterpret_cast<const uint4*>(sc.tris)[(size_t)tIdx * 4u + part];
    }
#pragma unroll
    for (uint32_t cidx = 0; cidx < 3u; cidx++) S->buf[cidx * 64u + lane] = piece[cidx];
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
  }
  if (act) {
    uint4 a, b, c;
    if (COOP) { const uint4* p = S->buf + lane * 3u; a = p[0]; b = p[1]; c = p[2]; }
    else if (ALL_LDS || triIdx < ldsTris) { const uint4* p = s_tris + triIdx * 3u; a = p[0]; b = p[1]; c = p[2]; }
    else { const uint4* p = reinterpret_cast<const uint4*>(sc.tris) + (size_t)triIdx * 4u; a = p[0]; b = p[1]; c = p[2]; }
    if (COUNT) tc.tris++;
    float t, u, v;
    bool accept = tri_test(o, d, tMin, a, b, c, t, u, v);
    if (CUTOUT && accept && (c.w & (1u << 28))) { // non-opaque material: stochastic cutout (ignoreIntersectionEXT, rp_main.ahit:57-60)
      const float opacity = sc.materials[c.w & 0x00ffffffu].p[MP_CUTOUT];
      accept = !(cutout_random(rrng, c.y) > opacity);
    }
    if (accept) {
      const unsigned long long key = ((unsigned long long)f2u(t) << 32) | (unsigned long long)(c.y + 1u);
      atomicMin(&W.best[rl], key);
      if (*(volatile unsigned long long*)&W.best[rl] == key) W.hit[rl] = make_uint4(triIdx, f2u(u), f2u(v), c.w);
    }
  }
}

// One step of all rays of a wave: node phase per lane, then the cooperative triangle stage, then pop.  Wave-uniform
// control flow; lanes without a ray (alive == false) only help testing triangles.  Returns true when this lane's ray is finished.
template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool ALL_LDS, bool CUTOUT, bool COOP>
__device__ __forceinline__ bool wave_step(RayTrav& R, bool alive, WaveTri& W, WaveStage* S, const SceneView& sc, const uint4* s_nodes, uint32_t ldsNodes,
                                          const uint4* s_tris, uint32_t ldsTris, uint2 (*s_stack)[TRACE_BLOCK], uint2 (&overflow)[OVERFLOW ? OVF_STACK : 1],
                                          TraceCounters& tc, uint32_t rng)
{
  const uint32_t lane = __lane_id();
  uint2 Gt = make_uint2(0u, 0u);
  if (!COOP) {
    if (alive) Gt = trav_node<COUNT, STACK, OVERFLOW, ALL_LDS>(R, sc, s_nodes, ldsNodes, s_stack, overflow, tc);
  } else {
    uint32_t nodeIdx = 0xffffffffu;
    if (alive) nodeIdx = trav_node_pick<STACK, OVERFLOW>(R, s_stack, overflow);
    uint4 piece[5]; // piece (flat % 5) of the node lane (flat / 5) asked for, flat = lane, 64 + lane, ...
#pragma unroll
    for (uint32_t cidx = 0; cidx < 5u; cidx++) {
      const uint32_t flat = cidx * 64u + lane, owner = flat / 5u, part = flat - owner * 5u;
      const uint32_t nIdx = (uint32_t)__shfl((int)nodeIdx, (int)owner);
      piece[cidx] = make_uint4(0u, 0u, 0u, 0u);
      if (nIdx != 0xffffffffu) piece[cidx] = reinterpret_cast<const uint4*>(sc.nodes)[(size_t)nIdx * sc.nodeStrideU4 + part];
    }
#pragma unroll
    for (uint32_t cidx = 0; cidx < 5u; cidx++) S->buf[cidx * 64u + lane] = piece[cidx];
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    if (alive) {
      const uint4* p = S->buf + lane * 5u;
      const uint4 n0 = p[0], n1 = p[1], n2 = p[2], n3 = p[3], n4 = p[4];
      if (COUNT) tc.nodes++;
      Gt = trav_node_test(R, n0, n1, n2, n3, n4);
    }
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
  }
  uint32_t head = 0u, tail = 0u; // wave-uniform
  for (;;) {
    const unsigned long long m = __ballot(Gt.y != 0u);
    if (!m) break;
    if (Gt.y) {
      const uint32_t k = (uint32_t)__ffs((int)Gt.y) - 1u;
      Gt.y &= Gt.y - 1u;
      *(volatile uint32_t*)&W.queue[(tail + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & 127u] = (lane << TRI_ID_BITS) | (Gt.x + k);
    }
    tail += (uint32_t)__popcll(m);
    if (tail - head >= 64u) { wave_tri_batch<COUNT, ALL_LDS, CUTOUT, COOP>(W, S, head, 64u, R, rng, sc, s_tris, ldsTris, tc); head += 64u; }
  }
  if (tail != head) wave_tri_batch<COUNT, ALL_LDS, CUTOUT, COOP>(W, S, head, tail - head, R, rng, sc, s_tris, ldsTris, tc);
  bool done = false;
  if (alive) {
    const unsigned long long key = *(volatile unsigned long long*)&W.best[lane];
    R.tBest = u2f((uint32_t)(key >> 32));
    R.found = (uint32_t)key != 0u;
    done = (ANYHIT && R.found) ? true : trav_pop<STACK, OVERFLOW>(R, s_stack, overflow);
  }
  return done;
}

// start of a ray in the cooperative scheme (after trav_init)
__device__ __forceinline__ void wave_ray_begin(WaveTri& W, float tMax) { *(volatile unsigned long long*)&W.best[__lane_id()] = (unsigned long long)f2u(tMax) << 32; }
// result of a finished ray
__device__ __forceinline__ void wave_ray_end(WaveTri& W, RayTrav& R)
{
  __atomic_signal_fence(__ATOMIC_SEQ_CST); // compiler only: the winning lane's store precedes this load in the wave's program order
  if (R.found) { const uint4 h = W.hit[__lane_id()]; R.bestTri = h.x; R.bestU = u2f(h.y); R.bestV = u2f(h.z); R.bestMat = h.w; }
}

template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool ALL_LDS, bool CUTOUT>
__device__ __forceinline__ bool traverse(const SceneView& sc, const uint4* s_nodes, uint32_t ldsNodes, const uint4* s_tris, uint32_t ldsTris,
                                         uint2 (*s_stack)[TRACE_BLOCK], V3 o, V3 d, float tMin, float tMax,
                                         float& outT, float& outU, float& outV, uint32_t& outTri, uint32_t& outMat, TraceCounters& tc, uint32_t rng = 0u)
{
  RayTrav R; trav_init(R, o, d, tMin, tMax);
  uint2 overflow[OVERFLOW ? OVF_STACK : 1];
  while (!trav_step<ANYHIT, COUNT, STACK, OVERFLOW, ALL_LDS, CUTOUT>(R, sc, s_nodes, ldsNodes, s_tris, ldsTris, s_stack, overflow, tc, rng)) {}
  outT = R.tBest; outU = R.bestU; outV = R.bestV; outTri = R.bestTri; outMat = R.bestMat;
  return R.found;
}

// shadeRayPayloadGetMediumIdx / shadeRayPayloadIncrementWalk (rp_main_payload.glsl:60-90), literally (see the oracle's notes:
// the increment lands in bit 0 of the bounce counter, the walk length never grows)
__device__ __forceinline__ uint32_t payload_medium_idx(uint32_t bitfield, uint32_t stackSize)
{
  const uint32_t idx = (bitfield & 0x0f000000u) >> 24, mx = stackSize > 1u ? stackSize : 1u;
  return idx < mx ? idx : mx;
}
__device__ __forceinline__ void payload_increment_walk(uint32_t& bitfield)
{
  uint32_t b = bitfield & 0x00fff000u;
  b = (b + 1u) < 0x00fff000u ? (b + 1u) : 0x00fff000u;
  bitfield &= ~0x00fff000u;
  bitfield |= b;
}

// NEE AOV bookkeeping (see PathState): latest shadow-ray outcome per tile pixel in the reference's (sample, bounce) order
__device__ __forceinline__ void nee_aov_record(const PathState& st, uint32_t slot, bool shadowed)
{
  const Slot* S = &st.slots[slot];
  const F4 id = ld4(&S->id);
  const unsigned long long order = ((unsigned long long)(st.neeSampleBase + f2u(id.y)) << 12) | (unsigned long long)(f2u(S->thr.w) & 0x00000fffu);
  atomicMax(&st.neeKey[f2u(id.x)], (order << 1) | (shadowed ? 1ull : 0ull));
}
__global__ void k_resolve_nee(const unsigned long long* __restrict__ key, F4* __restrict__ aov, uint32_t pixelCount, uint32_t firstPixel)
{
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pixelCount) return;
  const unsigned long long k = key[p];
  if (k == 0ull) return; // no shadow ray traced for this pixel: the AOV keeps its clear value
  F4* dst = &aov[firstPixel + p];
  dst->x = (k & 1ull) ? 1.0f : 0.0f; dst->y = (k & 1ull) ? 0.0f : 1.0f; dst->z = 0.0f;
}

// rp_main.miss:55-86 for scenes with a dome light image (defined with the texture runtime below): adds
// throughput * dome(direction) to the slot's radiance.  Without a dome image the miss term is the constant fallback dome,
// which k_raygen applies when it retires the path (REGEN_MISSED).
__device__ void dome_miss(const SceneView& sc, Slot* S, V3 rayDir);

template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool ALL_LDS, bool CUTOUT, bool DOME>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace(SceneView sc, PathState st, QueueSet qs, Counters* cnt, uint32_t qIn, uint32_t qMiss, uint32_t ldsNodes, uint32_t ldsTris)
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
  for (uint32_t i = threadIdx.x; i < ldsNodes * 5u; i += TRACE_BLOCK) s_nodes[i] = reinterpret_cast<const uint4*>(sc.nodes)[(i / 5u) * sc.nodeStrideU4 + (i % 5u)];
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
    bool alive = false; uint32_t r = 0u, rng = 0u;
    if (i < n) {
      r = reader_index(rd, i);
      slot = qs.slot[qIn][r];
      ro = ld4(&qs.a[qIn][r]);
      rdir = ld4(&qs.b[qIn][r]);
      rng = CUTOUT ? f2u(st.slots[slot].rad.w) : 0u; // the any-hit test needs the path's rng state
      // shadow ray (rp_main.rgen:397-429): origin = next ray origin, tMin 0.01, tMax = distance to the light sample
      if (!ANYHIT) trav_init(R, v3(ro.x, ro.y, ro.z), v3(rdir.x, rdir.y, rdir.z), ro.w, rdir.w);
      else trav_init(R, v3(ro.x, ro.y, ro.z), v3(rdir.x, rdir.y, rdir.z), 0.01f, ro.w);
      wave_ray_begin(W, R.tBest);
      alive = true;
    }
    while (__ballot(alive)) {
      if (wave_step<ANYHIT, COUNT, STACK, OVERFLOW, ALL_LDS, CUTOUT, false>(R, alive, W, nullptr, sc, s_nodes, ldsNodes, s_tris, ldsTris, s_stack, overflow, tc, rng)) alive = false;
    }
    if (i < n) {
      wave_ray_end(W, R);
      if (!ANYHIT) {
        hit = R.found; miss = !hit; t = R.tBest; u = R.bestU; v = R.bestV; tri = R.bestTri; mat = R.bestMat;
      } else {
        if (!R.found) {
          const F4 nc = ld4(&qs.c[qIn][r]);
          Slot* S = &st.slots[slot];
          F4 rr = ld4(&S->rad);
          st4(&S->rad, rr.x + nc.x, rr.y + nc.y, rr.z + nc.z, rr.w);
        }
        if (st.neeKey) nee_aov_record(st, slot, R.found);
      }
    }
    if (!ANYHIT) {
      // sort by outcome and material class: hits go to their class's shade queue as (slot, hit, direction) records, misses
      // straight to k_raygen
      uint32_t klass = (mat >> 24) & 0xfu;
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
      if (hit || volMiss) {
        const uint32_t q = Q_HIT + klass, r = idx[1 + klass];
        qs.slot[q][r] = slot;
        if (!volMiss) { st4(&qs.a[q][r], t, u, v, u2f(tri)); st4(&qs.b[q][r], rdir.x, rdir.y, rdir.z, 0.0f); }
        else { st4(&qs.a[q][r], rdir.w, ro.x, ro.y, u2f(VOLUME_MISS)); st4(&qs.b[q][r], rdir.x, rdir.y, rdir.z, ro.z); } // (tMax, origin) ride along
      }
      if (miss) {
        if (DOME && sc.domeTexture) { dome_miss(sc, &st.slots[slot], v3(rdir.x, rdir.y, rdir.z)); qs.slot[qMiss][idx[0]] = slot; } // scene has a dome light image
        else qs.slot[qMiss][idx[0]] = slot | REGEN_MISSED;
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
// record (a = (t,u,v,tri), b.w = material word) and, once `refill` lanes of the wave are idle, the wave claims that many
// new rays with one atomic on the launch's cursor.  No barriers, no appends; k_route then streams the results into the
// per-class shade queues / the regen queue.  Per-ray arithmetic is trav_step's, i.e. identical to k_trace's.
// ------------------------------------------------------------------------------------------------
template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool CUTOUT>
__device__ __forceinline__ void trace_dyn_body(const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t qIn, uint32_t refill)
{
  extern __shared__ uint4 s_dyn[];
  uint2 (*s_stack)[TRACE_BLOCK] = reinterpret_cast<uint2 (*)[TRACE_BLOCK]>(s_dyn);
  __shared__ WaveTri s_wave[TRACE_BLOCK / 64];
  __shared__ WaveStage s_stage[TRACE_DYN_COOP_FETCH ? TRACE_BLOCK / 64 : 1];
  WaveTri& W = s_wave[threadIdx.x >> 6];
  WaveStage* S = TRACE_DYN_COOP_FETCH ? &s_stage[threadIdx.x >> 6] : nullptr;
  QueueReader rd; reader_init(rd, cnt, qIn, qs.cap);
  const uint32_t n = rd.pre[NSHARD];
  if (blockIdx.x == 0 && threadIdx.x == 0) { if (ANYHIT) cnt->shadowRays += n; else cnt->segments += n; } // single writer per launch
  uint32_t* cursor = &cnt->cursor[ANYHIT ? 1 : 0].v;
  const uint32_t lane = __lane_id();
  TraceCounters tc{0u, 0u};
  RayTrav R; trav_init(R, v3(0.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, 1.0f), 0.0f, 0.0f);
  uint2 overflow[OVERFLOW ? OVF_STACK : 1];
  bool alive = false;
  uint32_t rec = 0u, rng = 0u;
  // The wave claims rays 64 at a time (one atomic per chunk) and lane j prefetches ray j of the chunk into registers; lanes
  // that run idle are then handed the chunk's rays in order with register shuffles, so a refill never waits on memory.
  F4 pro = F4{0.0f, 0.0f, 0.0f, 0.0f}, prd = F4{0.0f, 0.0f, 0.0f, 0.0f}; uint32_t prec = 0u, prng = 0u;
  uint32_t chunkCount = 0u, chunkUsed = 0u; // wave-uniform
  auto next_chunk = [&]() {
    uint32_t base = 0u;
    if (lane == 0u) base = atomicAdd(cursor, 64u);
    base = (uint32_t)__shfl((int)base, 0);
    chunkCount = base < n ? (n - base < 64u ? n - base : 64u) : 0u;
    chunkUsed = 0u;
    if (lane < chunkCount) {
      prec = reader_index(rd, base + lane);
      pro = ld4(&qs.a[qIn][prec]);
      prd = ld4(&qs.b[qIn][prec]);
      if (CUTOUT) prng = f2u(st.slots[qs.slot[qIn][prec]].rad.w); // the any-hit test needs the path's rng state
    }
  };
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
        if (!ANYHIT) trav_init(R, v3(ro.x, ro.y, ro.z), v3(rdir.x, rdir.y, rdir.z), ro.w, rdir.w);
        else trav_init(R, v3(ro.x, ro.y, ro.z), v3(rdir.x, rdir.y, rdir.z), 0.01f, ro.w); // shadow ray (rp_main.rgen:397-429)
        wave_ray_begin(W, R.tBest);
        alive = true;
      }
      chunkUsed += take;
      if (chunkUsed == chunkCount) next_chunk(); // loads complete while the wave keeps traversing
    }
    if (!__ballot(alive)) { if (chunkCount == 0u) break; else continue; }
    const bool done = wave_step<ANYHIT, COUNT, STACK, OVERFLOW, false, CUTOUT, TRACE_DYN_COOP_FETCH>(R, alive, W, S, sc, nullptr, 0u, nullptr, 0u, s_stack, overflow, tc, rng);
    if (alive && done) {
      alive = false;
      wave_ray_end(W, R);
      if (!ANYHIT) {
        if (R.found) { st4(&qs.a[qIn][rec], R.tBest, R.bestU, R.bestV, u2f(R.bestTri)); reinterpret_cast<uint32_t*>(&qs.b[qIn][rec])[3] = R.bestMat; }
        else { st4(&qs.a[qIn][rec], R.tBest, R.o.x, R.o.y, u2f(MISS)); reinterpret_cast<float*>(&qs.b[qIn][rec])[3] = R.o.z; } // (tMax, origin): k_route needs them for scattering events
      } else {
        const uint32_t slot = qs.slot[qIn][rec];
        if (!R.found) {
          const F4 nc = ld4(&qs.c[qIn][rec]);
          Slot* S = &st.slots[slot];
          F4 rr = ld4(&S->rad);
          st4(&S->rad, rr.x + nc.x, rr.y + nc.y, rr.z + nc.z, rr.w);
        }
        if (st.neeKey) nee_aov_record(st, slot, R.found);
      }
    }
  }
  if (COUNT) { // measurement builds only: one atomic pair per wave
    unsigned long long a = tc.nodes, b = tc.tris;
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    if (lane == 0) { atomicAdd(ANYHIT ? &cnt->shadowNodesVisited : &cnt->nodesVisited, a); atomicAdd(ANYHIT ? &cnt->shadowTrisTested : &cnt->trisTested, b); }
  }
}

template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool CUTOUT>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_trace_dyn(SceneView sc, PathState st, QueueSet qs, Counters* cnt, uint32_t qIn, uint32_t refill)
{
  trace_dyn_body<ANYHIT, COUNT, STACK, OVERFLOW, CUTOUT>(sc, st, qs, cnt, qIn, refill);
}

// k_route: sorts k_trace_dyn's in-place results by outcome and material class (same routing as k_trace's epilogue)
__global__ __launch_bounds__(BLOCK) void k_route(SceneView sc, PathState st, QueueSet qs, Counters* cnt, uint32_t qIn, uint32_t qMiss)
{
  __shared__ AppendScratch<1 + MAT_CLASS_COUNT> sh;
  QueueReader rd; reader_init(rd, cnt, qIn, qs.cap);
  const uint32_t n = rd.pre[NSHARD];
  const uint32_t stride = gridDim.x * BLOCK;
  uint32_t trip = 0;
  for (uint32_t base = blockIdx.x * BLOCK; base < n; base += stride, trip++) {
    const uint32_t i = base + threadIdx.x;
    bool hit = false, miss = false, volMiss = false; uint32_t slot = 0, klass = 0;
    F4 h = F4{0.0f, 0.0f, 0.0f, 0.0f}, rdir = F4{0.0f, 0.0f, 0.0f, 0.0f};
    if (i < n) {
      const uint32_t r = reader_index(rd, i);
      slot = qs.slot[qIn][r];
      h = ld4(&qs.a[qIn][r]);
      rdir = ld4(&qs.b[qIn][r]);
      hit = f2u(h.w) != MISS; miss = !hit;
      klass = (f2u(rdir.w) >> 24) & 0xfu;
      if (miss && sc.mediumStackSize) { // the segment ended inside a medium: scattering event for k_shade<2> (rp_main.miss:57-66)
        volMiss = payload_medium_idx(f2u(st.slots[slot].thr.w), sc.mediumStackSize < MAX_MEDIUM_STACK ? sc.mediumStackSize : MAX_MEDIUM_STACK) > 0u;
        if (volMiss) { miss = false; klass = 2u; }
      }
    }
    bool pred[1 + MAT_CLASS_COUNT]; uint32_t qid[1 + MAT_CLASS_COUNT]; uint32_t idx[1 + MAT_CLASS_COUNT];
    pred[0] = miss; qid[0] = qMiss;
#pragma unroll
    for (uint32_t c = 0; c < MAT_CLASS_COUNT; c++) { pred[1 + c] = (hit || volMiss) && klass == c; qid[1 + c] = Q_HIT + c; }
    block_append<1 + MAT_CLASS_COUNT>(sh, trip, pred, qid, qs.cap, cnt, idx);
    if (hit || volMiss) {
      const uint32_t q = Q_HIT + klass, r = idx[1 + klass];
      qs.slot[q][r] = slot;
      if (!volMiss) { st4(&qs.a[q][r], h.x, h.y, h.z, h.w); st4(&qs.b[q][r], rdir.x, rdir.y, rdir.z, 0.0f); }
      else { st4(&qs.a[q][r], h.x, h.y, h.z, u2f(VOLUME_MISS)); st4(&qs.b[q][r], rdir.x, rdir.y, rdir.z, rdir.w); }
    }
    if (miss) {
      if (sc.domeTexture) { dome_miss(sc, &st.slots[slot], v3(rdir.x, rdir.y, rdir.z)); qs.slot[qMiss][idx[0]] = slot; }
      else qs.slot[qMiss][idx[0]] = slot | REGEN_MISSED;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Shading state (mdl_shading_state.glsl:4-98) from flat scene buffers
// ------------------------------------------------------------------------------------------------
struct ShState {
  V3 normal, geomNormal, position, tangentU, tangentV; bool frontFace; uint32_t meshFlags, material;
  float u, v;                    // texture coordinate 0 (mdl_shading_state.glsl:62-65)
  uint32_t mesh, prim, vi[3]; int32_t instanceId; float hu, hv; // renderer state for scene-data lookups (mdl_interface.glsl:281-301)
  float ior1, ior2;              // Bsdf_sample_data.ior1/ior2 (rp_main.chit:188-189): < 0 = the material's own; 0 = empty-stack default
  uint32_t texMask;              // bit per TEX_* slot whose value below replaces the material constant at this hit
  V3 texBaseColor, texEmission; float texRoughness, texMetallic;
};

__device__ __forceinline__ V3 xform_point(const float* a, V3 p, float w)
{
  return v3(((a[0] * p.x + a[1] * p.y) + a[2] * p.z) + a[3] * w,
            ((a[4] * p.x + a[5] * p.y) + a[6] * p.z) + a[7] * w,
            ((a[8] * p.x + a[9] * p.y) + a[10] * p.z) + a[11] * w);
}
__device__ __forceinline__ V3 xform_normal(const float* w, V3 n)
{
  return v3((n.x * w[0] + n.y * w[3]) + n.z * w[6], (n.x * w[1] + n.y * w[4]) + n.z * w[7], (n.x * w[2] + n.y * w[5]) + n.z * w[8]);
}

__device__ __forceinline__ void setup_shading_state(const SceneView& sc, uint32_t triIdx, float hu, float hv, V3 rayDir, ShState& s)
{
  // one dependent step: the triangle record's tail names the instance, the material and the three vertices
  const uint4* tp = reinterpret_cast<const uint4*>(sc.tris) + (size_t)triIdx * 4u;
  const uint4 tc = tp[2], td = tp[3]; // (e2.z, origId, instance, matFlags), (i0, i1, i2, prim)
  s.material = tc.w & 0x00ffffffu; s.meshFlags = tc.w >> 30;
  const float4* ip = reinterpret_cast<const float4*>(&sc.instances[tc.z]);
  const float4* va = reinterpret_cast<const float4*>(&sc.verts[td.x]);
  const float4* vb = reinterpret_cast<const float4*>(&sc.verts[td.y]);
  const float4* vc = reinterpret_cast<const float4*>(&sc.verts[td.z]);
  const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2], r3 = ip[3], r4 = ip[4], r5 = ip[5];
  const float4 a1 = va[0], a2 = va[1], a3 = va[2];
  const float4 b1 = vb[0], b2 = vb[1], b3 = vb[2];
  const float4 c1 = vc[0], c2 = vc[1], c3 = vc[2];
  const float o2w[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
  const float w2o[9] = {r3.x, r3.y, r3.z, r3.w, r4.x, r4.y, r4.z, r4.w, r5.x};
  const float bx = 1.0f - hu - hv, by = hu, bz = hv;                                  // :17
  const V3 pa = v3(a1.x, a1.y, a1.z), pb = v3(b1.x, b1.y, b1.z), pc = v3(c1.x, c1.y, c1.z);
  const V3 localPos = (pa * bx + pb * by) + pc * bz;                                  // :24
  s.position = xform_point(o2w, localPos, 1.0f);                                      // :25
  V3 gn = normalize(cross(pb - pa, pc - pa));                                         // :27
  gn = normalize(xform_normal(w2o, gn));                                              // :28
  const V3 n0 = v3(a2.x, a2.y, a2.z), n1 = v3(b2.x, b2.y, b2.z), n2 = v3(c2.x, c2.y, c2.z); // decoded on the host (:31-33)
  const V3 ln = normalize((n0 * bx + n1 * by) + n2 * bz);                             // :35
  V3 nrm = normalize(xform_normal(w2o, ln));                                          // :36
  s.frontFace = dot(gn, -rayDir) >= 0.0f;                                             // :39
  if (!s.frontFace) { gn = -gn; nrm = -nrm; }                                         // :41-45
  const V3 t0 = v3(a3.x, a3.y, a3.z), t1 = v3(b3.x, b3.y, b3.z), t2 = v3(c3.x, c3.y, c3.z); // decoded on the host (:48-50)
  const V3 lt = normalize((t0 * bx + t1 * by) + t2 * bz);                             // :52
  V3 tg = normalize(xform_point(o2w, lt, 0.0f));                                      // :53
  tg = normalize(tg - nrm * dot(tg, nrm));                                            // :56
  const float bs = (bx * a1.w + by * b1.w) + bz * c1.w;                               // :58
  s.tangentU = tg; s.tangentV = cross(nrm, tg) * bs;                                  // :59
  s.u = (bx * a2.w + by * b2.w) + bz * c2.w; s.v = (bx * a3.w + by * b3.w) + bz * c3.w; // :62-65
  s.normal = nrm; s.geomNormal = gn;
  s.mesh = f2u(r5.y); s.instanceId = (int32_t)f2u(r5.z); s.prim = td.w; s.vi[0] = td.x; s.vi[1] = td.y; s.vi[2] = td.z; s.hu = hu; s.hv = hv;
  s.ior1 = 0.0f; s.ior2 = 0.0f;
  s.texMask = 0u; s.texBaseColor = v3(0.0f, 0.0f, 0.0f); s.texEmission = s.texBaseColor; s.texRoughness = 0.0f; s.texMetallic = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// Texture runtime (mdl_interface.glsl:8-38 apply_wrap_and_crop, :127-145 tex_lookup_float4_2d) over a software sampler:
// bilinear, REPEAT addressing, LOD 0 (the reference's single sampler, Gi.cpp:388-392, CgpuVk.cpp:1985-1990).
// Operation order == oracle sample_bilinear_repeat / tex_lookup_float4_2d.
// ------------------------------------------------------------------------------------------------
__device__ inline F4 sample_bilinear_repeat(const TextureRec& t, float u, float v)
{
  u = u - floorf(u); v = v - floorf(v);
  const float x = u * (float)t.width - 0.5f, y = v * (float)t.height - 0.5f;
  const float x0f = floorf(x), y0f = floorf(y);
  const float fx = x - x0f, fy = y - y0f;
  const int w = (int)t.width, h = (int)t.height;
  int ix0 = (int)x0f, iy0 = (int)y0f;
  if (ix0 < 0) ix0 += w;
  if (iy0 < 0) iy0 += h;
  int ix1 = ix0 + 1; if (ix1 >= w) ix1 -= w;
  int iy1 = iy0 + 1; if (iy1 >= h) iy1 -= h;
  const F4* tx = reinterpret_cast<const F4*>(t.texels);
  const F4 t00 = ld4(&tx[(size_t)iy0 * w + ix0]), t10 = ld4(&tx[(size_t)iy0 * w + ix1]);
  const F4 t01 = ld4(&tx[(size_t)iy1 * w + ix0]), t11 = ld4(&tx[(size_t)iy1 * w + ix1]);
  const float gx = 1.0f - fx, gy = 1.0f - fy;
  F4 o;
  o.x = (t00.x * gx + t10.x * fx) * gy + (t01.x * gx + t11.x * fx) * fy;
  o.y = (t00.y * gx + t10.y * fx) * gy + (t01.y * gx + t11.y * fx) * fy;
  o.z = (t00.z * gx + t10.z * fx) * gy + (t01.z * gx + t11.z * fx) * fy;
  o.w = (t00.w * gx + t10.w * fx) * gy + (t01.w * gx + t11.w * fx) * fy;
  return o;
}
__device__ __forceinline__ float apply_wrap_and_crop(float coord, uint32_t wrap, uint32_t res) // crop = (0, 1)
{
  if (wrap == TEX_WRAP_REPEAT) coord = coord - floorf(coord);
  else {
    if (wrap == TEX_WRAP_MIRRORED_REPEAT) {
      const float tmp = floorf(coord);
      if (((int)tmp & 1) != 0) coord = 1.0f - (coord - tmp); else coord = coord - tmp;
    }
    const float inv_hdim = 0.5f / (float)res;
    coord = fmin2(fmax2(coord, inv_hdim), 1.0f - inv_hdim);
  }
  return coord;
}
__device__ inline F4 tex_lookup_float4_2d(const TextureRec& t, float u, float v, uint32_t wrapU, uint32_t wrapV)
{
  if ((wrapU == TEX_WRAP_CLIP && (u < 0.0f || u > 1.0f)) || (wrapV == TEX_WRAP_CLIP && (v < 0.0f || v > 1.0f))) return F4{0.0f, 0.0f, 0.0f, 0.0f};
  u = apply_wrap_and_crop(u, wrapU, t.width);
  v = apply_wrap_and_crop(v, wrapV, t.height);
  return sample_bilinear_repeat(t, u, v);
}
// mdl_adapt_normal (mdl_interface.glsl:238-256): Iray's shadow-terminator bend of a mapped normal
__device__ __forceinline__ V3 adapt_normal(V3 rayDir, V3 geomNormal, V3 normal)
{
  const float dn = dot(rayDir, normal);
  const V3 r = normalize(rayDir - normal * (2.0f * dn));
  const float a = fmax2(0.0f, dot(r, -geomNormal));
  const float b = dot(normal, geomNormal);
  const V3 tangent = normalize(r + normal * (a / b));
  return normalize(-rayDir + tangent);
}
// Evaluates the material's textured inputs at the hit (== oracle resolve_material); a normal map replaces the shading frame.
__device__ inline void resolve_material_textures(const SceneView& sc, const MaterialRec* m, V3 rayDir, ShState& st)
{
#pragma unroll
  for (uint32_t slot = 0; slot < TEX_SLOT_COUNT; slot++) {
    const TexBindingRec& b = m->tex[slot];
    if (b.tex == 0u) {
      if (!(b.mode & TEX_MODE_PRIMVAR) || slot == TEX_NORMAL) continue;
      // scene_data_lookup_float3 / _float (mdl_interface.glsl:337-371, 398-424; == oracle scene_data_lookup)
      const MeshRec& mr = sc.meshes[st.mesh];
      const uint32_t info = mr.sdInfo[slot];
      if (!(info & 1u)) continue; // SCENE_DATA_INVALID: the input keeps its constant
      const uint32_t stride = ((info >> 1) & 3u) + 1u, interp = (info >> 3) & 3u;
      uint32_t i0, i1, i2;
      if (interp == 2u) i0 = i1 = i2 = st.prim;                       // uniform
      else if (interp == 1u) i0 = i1 = i2 = (uint32_t)st.instanceId;  // instance
      else if (interp == 0u) i0 = i1 = i2 = 0u;                       // constant
      else { i0 = st.vi[0] - mr.vertexOffset; i1 = st.vi[1] - mr.vertexOffset; i2 = st.vi[2] - mr.vertexOffset; } // vertex
      const float* d = sc.sceneData + mr.sdOffset[slot];
      const float bx = 1.0f - st.hu - st.hv, by = st.hu, bz = st.hv;
      const bool vec = slot == TEX_BASE_COLOR || slot == TEX_EMISSION;
      float o[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
      for (uint32_t c = 0; c < 3u; c++) if (c == 0u || vec) o[c] = (d[i0 * stride + c] * bx + d[i1 * stride + c] * by) + d[i2 * stride + c] * bz;
      st.texMask |= 1u << slot;
      if (slot == TEX_BASE_COLOR) st.texBaseColor = v3(o[0], o[1], o[2]);
      else if (slot == TEX_EMISSION) st.texEmission = v3(o[0], o[1], o[2]);
      else if (slot == TEX_ROUGHNESS) st.texRoughness = o[0];
      else st.texMetallic = o[0];
      continue;
    }
    const F4 t = tex_lookup_float4_2d(sc.textures[b.tex - 1u], st.u, st.v, b.mode & 0xffu, (b.mode >> 8) & 0xffu);
    const float val[4] = {t.x * b.scale[0] + b.bias[0], t.y * b.scale[1] + b.bias[1], t.z * b.scale[2] + b.bias[2], t.w * b.scale[3] + b.bias[3]};
    const uint32_t ch = (b.mode >> 16) & 3u;
    const float sel = ch == 0u ? val[0] : (ch == 1u ? val[1] : (ch == 2u ? val[2] : val[3]));
    st.texMask |= 1u << slot;
    if (slot == TEX_BASE_COLOR) st.texBaseColor = v3(val[0], val[1], val[2]);
    else if (slot == TEX_EMISSION) st.texEmission = v3(val[0], val[1], val[2]);
    else if (slot == TEX_ROUGHNESS) st.texRoughness = sel;
    else if (slot == TEX_METALLIC) st.texMetallic = sel;
    else {
      V3 n = normalize((st.tangentU * val[0] + st.tangentV * val[1]) + st.normal * val[2]);
      n = adapt_normal(rayDir, st.geomNormal, n);
      const float hs = dot(cross(st.normal, st.tangentU), st.tangentV) >= 0.0f ? 1.0f : -1.0f;
      const V3 tg = normalize(st.tangentU - n * dot(st.tangentU, n));
      st.normal = n; st.tangentU = tg; st.tangentV = cross(n, tg) * hs;
    }
  }
}

__device__ __forceinline__ V3 quat_rotate_dir(const float* q, V3 dir) // rp_main.miss:38-44
{
  const V3 qv = v3(q[0], q[1], q[2]);
  const V3 a = cross(qv, dir);
  const V3 b = cross(qv, a);
  return dir + ((a * q[3]) + b) * 2.0f;
}
__device__ void dome_miss(const SceneView& sc, Slot* S, V3 rayDir)
{
  const F4 tb = ld4(&S->thr);
  const F4 rr = ld4(&S->rad);
  const bool isPrimaryRay = (f2u(tb.w) & 0x00000fffu) == 0u;
  const bool useFallback = !sc.domeCameraVisible && isPrimaryRay; // :76-80
  V3 texel = v3(sc.background);
  if (!useFallback) {
    const V3 d = normalize(quat_rotate_dir(sc.domeRotation, rayDir)); // :83
    const float u = (gi_atan2f(d.z, d.x) + 0.5f * GI_PI) / (2.0f * GI_PI); // :48-49
    const float v = 1.0f - gi_acosf(d.y) / GI_PI;
    const F4 t = sample_bilinear_repeat(sc.textures[sc.domeTexture - 1u], u, v);
    texel = v3(t.x, t.y, t.z);
  }
  const V3 rad = v3(rr.x, rr.y, rr.z) + v3(tb.x, tb.y, tb.z) * (texel * v3(sc.domeEmission)); // :84-86
  st4(&S->rad, rad.x, rad.y, rad.z, rr.w);
  S->thr.w = u2f(f2u(tb.w) + 1u); // the loop's bounce++ (rp_main.rgen:480): the path retires, only the Bounces AOV reads it
}

// ------------------------------------------------------------------------------------------------
// Closed-form BSDFs (DESIGN.md "Materials"); replace mdl_bsdf_scattering_{sample,evaluate}
// (entry points GlslShaderGen.cpp:181-193; data contracts mdl_types.glsl:158-238)
// ------------------------------------------------------------------------------------------------
enum : uint32_t { EV_ABSORB = 0, EV_DIFFUSE = 1, EV_GLOSSY = 2, EV_SPECULAR = 4, EV_REFLECTION = 8, EV_TRANSMISSION = 16 };

__device__ __forceinline__ V3 to_world(const ShState& s, V3 l) { return (s.tangentU * l.x + s.tangentV * l.y) + s.normal * l.z; }
__device__ __forceinline__ V3 to_local(const ShState& s, V3 w) { return v3(dot(w, s.tangentU), dot(w, s.tangentV), dot(w, s.normal)); }
__device__ __forceinline__ float schlick_w(float c) { float m = 1.0f - c; m = fmin2(fmax2(m, 0.0f), 1.0f); float m2 = m * m; return m2 * m2 * m; }
__device__ __forceinline__ float ggx_lambda_term(float a2, float c) { return sqrtf(a2 + (1.0f - a2) * c * c); }
__device__ __forceinline__ V3 schlick3(V3 F0, float c) { float w = schlick_w(c); return F0 + (v3(1.0f, 1.0f, 1.0f) - F0) * w; }

struct GgxOut { V3 l2; float pdf, g2OverG1, kh; bool valid; };
__device__ inline GgxOut ggx_sample(V3 l1, float alpha, float x0, float x1)
{
  GgxOut o; o.valid = false; o.pdf = 0.0f; o.g2OverG1 = 0.0f; o.kh = 0.0f; o.l2 = v3(0.0f, 0.0f, 0.0f);
  V3 vh = normalize(v3(alpha * l1.x, alpha * l1.y, l1.z));
  float lensq = vh.x * vh.x + vh.y * vh.y;
  V3 T1 = lensq > 0.0f ? v3(-vh.y, vh.x, 0.0f) * (1.0f / sqrtf(lensq)) : v3(1.0f, 0.0f, 0.0f);
  V3 T2 = cross(vh, T1);
  float r = sqrtf(x0);
  float s, c; gi_sincos2pi(x1, &s, &c);
  float t1 = r * c, t2 = r * s;
  float sm = 0.5f * (1.0f + vh.z);
  t2 = (1.0f - sm) * sqrtf(fmax2(0.0f, 1.0f - t1 * t1)) + sm * t2;
  V3 nh = (T1 * t1 + T2 * t2) + vh * sqrtf(fmax2(0.0f, (1.0f - t1 * t1) - t2 * t2));
  V3 h = normalize(v3(alpha * nh.x, alpha * nh.y, fmax2(0.0f, nh.z)));
  float kh = dot(l1, h);
  V3 l2 = h * (2.0f * kh) - l1;
  if (!(l2.z > 0.0f) || !(kh > 0.0f)) return o;
  float a2 = alpha * alpha;
  float nk1 = l1.z, nk2 = l2.z, nh2 = h.z * h.z;
  float dd = nh2 * (a2 - 1.0f) + 1.0f;
  float D = a2 / (GI_PI * dd * dd);
  float L1 = ggx_lambda_term(a2, nk1), L2 = ggx_lambda_term(a2, nk2);
  float G1 = 2.0f * nk1 / (nk1 + L1);
  float G2 = 2.0f * nk1 * nk2 / (nk2 * L1 + nk1 * L2);
  o.l2 = l2; o.kh = kh; o.pdf = G1 * D / (4.0f * nk1); o.g2OverG1 = G2 / G1; o.valid = true;
  return o;
}
__device__ inline void ggx_eval(V3 l1, V3 l2, float alpha, float& fcos, float& pdf, float& kh)
{
  fcos = 0.0f; pdf = 0.0f; kh = 0.0f;
  if (!(l1.z > 0.0f) || !(l2.z > 0.0f)) return;
  V3 h = normalize(l1 + l2);
  kh = dot(l1, h);
  float a2 = alpha * alpha;
  float nk1 = l1.z, nk2 = l2.z, nh2 = h.z * h.z;
  float dd = nh2 * (a2 - 1.0f) + 1.0f;
  float D = a2 / (GI_PI * dd * dd);
  float L1 = ggx_lambda_term(a2, nk1), L2 = ggx_lambda_term(a2, nk2);
  float G1 = 2.0f * nk1 / (nk1 + L1);
  float G2 = 2.0f * nk1 * nk2 / (nk2 * L1 + nk1 * L2);
  fcos = D * G2 / (4.0f * nk1);
  pdf = G1 * D / (4.0f * nk1);
}

// ior2 / ior1 of the interface (== oracle relative_eta): eta entering, 1/eta leaving when the medium stack is empty
__device__ __forceinline__ float relative_eta(const ShState& st, float materialEta)
{
  float e1 = st.ior1 == 0.0f ? (st.frontFace ? 1.0f : -1.0f) : st.ior1;
  float e2 = st.ior2 == 0.0f ? (st.frontFace ? -1.0f : 1.0f) : st.ior2;
  if (e1 < 0.0f) e1 = materialEta;
  if (e2 < 0.0f) e2 = materialEta;
  return e2 / e1;
}
struct UpsParams { V3 albedo, F0; float alpha, coat, coatAlpha; };
// per-material constants are evaluated once on the host (gi_c.cpp: deriveMaterialConstants) with the same fp32
// formulas the oracle evaluates per hit
__device__ __forceinline__ UpsParams ups_params(const MaterialRec* m, const ShState& st)
{
  UpsParams u;
  u.albedo = v3(m->p[MP_ALBEDO], m->p[MP_ALBEDO + 1], m->p[MP_ALBEDO + 2]);
  u.F0 = v3(m->p[MP_F0], m->p[MP_F0 + 1], m->p[MP_F0 + 2]);
  u.alpha = m->p[MP_ALPHA]; u.coat = m->p[MP_COAT]; u.coatAlpha = m->p[MP_COAT_ALPHA];
  if (st.texMask & ((1u << TEX_BASE_COLOR) | (1u << TEX_ROUGHNESS) | (1u << TEX_METALLIC))) { // textured inputs: the oracle's per-hit formulas
    const V3 dc = (st.texMask & (1u << TEX_BASE_COLOR)) ? st.texBaseColor : v3(m->p[0], m->p[1], m->p[2]);
    const float r = (st.texMask & (1u << TEX_ROUGHNESS)) ? st.texRoughness : m->p[11];
    u.alpha = fmax2(r * r, 0.001f);
    if (m->p[6] != 0.0f) { u.F0 = v3(m->p[7], m->p[8], m->p[9]); u.albedo = dc; }
    else {
      const float ior = m->p[16], metal = (st.texMask & (1u << TEX_METALLIC)) ? st.texMetallic : m->p[10];
      const float q = (1.0f - ior) / (1.0f + ior), f0 = q * q;
      u.F0 = v3(f0, f0, f0) * (1.0f - metal) + dc * metal;
      u.albedo = dc * (1.0f - metal);
    }
  }
  return u;
}

struct BsdfSample { V3 k2, overPdf; float pdf; uint32_t event; };
struct BsdfEval { V3 diffuse, glossy; float pdf; };

// ---- class 2: OpenPBR (lobe graph of src/gi/mtlx/open_pbr_surface.mtlx:99-635, closed forms of our own) ----
__device__ __forceinline__ float fresnel_dielectric(float c, float eta)
{
  float sin2t = (1.0f - c * c) / (eta * eta);
  if (!(sin2t < 1.0f)) return 1.0f;
  float ct = sqrtf(1.0f - sin2t);
  float rs = (c - eta * ct) / (c + eta * ct);
  float rp = (eta * c - ct) / (eta * c + ct);
  return 0.5f * (rs * rs + rp * rp);
}
__device__ __forceinline__ V3 schlick_f82(V3 F0, V3 tint, float c)
{
  const float w5 = 0.462664366f, K = 17.6513846f;
  V3 one = v3(1.0f, 1.0f, 1.0f);
  V3 fb = F0 + (one - F0) * w5;
  V3 a = (fb * (one - tint)) * K;
  float m = 1.0f - c; m = fmin2(fmax2(m, 0.0f), 1.0f);
  float m2 = m * m, m5 = m2 * m2 * m, m6 = m5 * m;
  V3 f = (F0 + (one - F0) * m5) - a * (c * m6);
  return v3(fmin2(fmax2(f.x, 0.0f), 1.0f), fmin2(fmax2(f.y, 0.0f), 1.0f), fmin2(fmax2(f.z, 0.0f), 1.0f));
}
struct OpbrParams { V3 albedo, metalTint, specColor, transTint, coatTint; float metalness, alpha, coat, coatAlpha, coatF0, eta, tw, specWeight; };
__device__ __forceinline__ OpbrParams opbr_params(const MaterialRec* m, const ShState& st)
{
  OpbrParams o; const float* p = m->p;
  o.albedo = v3(p[MP_ALBEDO], p[MP_ALBEDO + 1], p[MP_ALBEDO + 2]);
  o.metalTint = v3(p[MP_F0], p[MP_F0 + 1], p[MP_F0 + 2]);
  o.specColor = v3(p[7], p[8], p[9]);
  o.specWeight = p[18]; o.metalness = p[10];
  o.alpha = p[MP_ALPHA]; o.coat = p[MP_COAT]; o.coatAlpha = p[MP_COAT_ALPHA]; o.coatF0 = p[MP_COAT_F0]; o.eta = p[MP_ETA];
  o.coatTint = v3(1.0f, 1.0f, 1.0f) * (1.0f - o.coat) + v3(p[19], p[20], p[21]) * o.coat;
  o.tw = p[23];
  o.transTint = (p[28] > 0.0f) ? v3(1.0f, 1.0f, 1.0f) : v3(p[24], p[25], p[26]);
  if (st.texMask & (1u << TEX_BASE_COLOR)) o.albedo = st.texBaseColor * p[17];
  if (st.texMask & (1u << TEX_ROUGHNESS)) o.alpha = fmax2(st.texRoughness * st.texRoughness, 0.001f);
  if (st.texMask & (1u << TEX_METALLIC)) o.metalness = st.texMetallic;
  return o;
}

__device__ inline void opbr_sample(const MaterialRec* m, const ShState& st, V3 k1, float x0, float x1, float x2, BsdfSample& out)
{
  OpbrParams o = opbr_params(m, st);
  V3 l1 = to_local(st, k1);
  float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
  float z = x2;
  float Fc = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(nk1));
  if (z < Fc) {
    GgxOut g = ggx_sample(l1, o.coatAlpha, x0, x1);
    V3 k2 = to_world(st, g.l2);
    if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
    float Fh = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(g.kh));
    float w = (Fh / Fc) * g.g2OverG1;
    out.k2 = k2; out.pdf = Fc * g.pdf; out.overPdf = v3(w, w, w); out.event = EV_GLOSSY | EV_REFLECTION;
    return;
  }
  z = (z - Fc) / (1.0f - Fc);
  if (z < o.metalness) {
    GgxOut g = ggx_sample(l1, o.alpha, x0, x1);
    V3 k2 = to_world(st, g.l2);
    if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
    V3 F = schlick_f82(o.albedo, o.metalTint, g.kh) * o.specWeight;
    out.k2 = k2; out.pdf = (1.0f - Fc) * o.metalness * g.pdf; out.overPdf = (F * o.coatTint) * g.g2OverG1; out.event = EV_GLOSSY | EV_REFLECTION;
    return;
  }
  z = (z - o.metalness) / (1.0f - o.metalness);
  float eta = relative_eta(st, o.eta);
  float Fd = fresnel_dielectric(nk1, eta);
  if (z < Fd) {
    GgxOut g = ggx_sample(l1, o.alpha, x0, x1);
    V3 k2 = to_world(st, g.l2);
    if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
    float Fh = fresnel_dielectric(g.kh, eta);
    out.k2 = k2; out.pdf = (1.0f - Fc) * (1.0f - o.metalness) * Fd * g.pdf;
    out.overPdf = (o.specColor * o.coatTint) * ((Fh / Fd) * g.g2OverG1); out.event = EV_GLOSSY | EV_REFLECTION;
    return;
  }
  z = (z - Fd) / (1.0f - Fd);
  if (z < o.tw) {
    GgxOut g = ggx_sample(l1, o.alpha, x0, x1);
    V3 h = normalize(l1 + g.l2);
    float kh = dot(l1, h);
    if (!g.valid || !(kh > 0.0f)) return;
    float Fh = fresnel_dielectric(kh, eta);
    float sin2t = (1.0f - kh * kh) / (eta * eta);
    if (!(sin2t < 1.0f)) return;
    float ct = sqrtf(1.0f - sin2t);
    V3 lt = h * (kh / eta - ct) - l1 * (1.0f / eta);
    V3 k2 = to_world(st, lt);
    if (!(lt.z < 0.0f) || !(dot(k2, st.geomNormal) < 0.0f)) return;
    float a2 = o.alpha * o.alpha, nk2 = -lt.z;
    float L1 = ggx_lambda_term(a2, nk1), L2 = ggx_lambda_term(a2, nk2);
    float G1 = 2.0f * nk1 / (nk1 + L1), G2 = 2.0f * nk1 * nk2 / (nk2 * L1 + nk1 * L2);
    float w = ((1.0f - Fh) / (1.0f - Fd)) * (G2 / G1);
    out.k2 = normalize(k2); out.pdf = (1.0f - Fc) * (1.0f - o.metalness) * (1.0f - Fd) * o.tw * g.pdf;
    out.overPdf = (o.transTint * o.coatTint) * w; out.event = EV_GLOSSY | EV_TRANSMISSION;
    return;
  }
  V3 l = gi_sample_hemisphere(x0, x1);
  V3 k2 = to_world(st, l);
  if (!(l.z > 0.0f) || !(dot(k2, st.geomNormal) > 0.0f)) return;
  out.k2 = k2; out.pdf = (1.0f - Fc) * (1.0f - o.metalness) * (1.0f - Fd) * (1.0f - o.tw) * (l.z / GI_PI);
  out.overPdf = o.albedo * o.coatTint; out.event = EV_DIFFUSE | EV_REFLECTION;
}

__device__ inline void opbr_evaluate(const MaterialRec* m, const ShState& st, V3 k1, V3 k2, BsdfEval& out)
{
  OpbrParams o = opbr_params(m, st);
  V3 l1 = to_local(st, k1), l2 = to_local(st, k2);
  float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
  float eta = relative_eta(st, o.eta);
  float Fc = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(nk1));
  float Fd = fresnel_dielectric(nk1, eta);
  float fc, pc, khc; ggx_eval(l1, l2, o.coatAlpha, fc, pc, khc);
  float fs, ps, khs; ggx_eval(l1, l2, o.alpha, fs, ps, khs);
  float Fch = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(khc));
  V3 Fm = schlick_f82(o.albedo, o.metalTint, khs) * o.specWeight;
  float Fdh = fresnel_dielectric(khs, eta);
  float cd = l2.z / GI_PI;
  float base = 1.0f - Fc, diel = 1.0f - o.metalness;
  V3 gl = v3(Fch * fc, Fch * fc, Fch * fc);
  gl = gl + ((Fm * o.coatTint) * fs) * (base * o.metalness);
  gl = gl + ((o.specColor * o.coatTint) * (Fdh * fs)) * (base * diel);
  out.glossy = gl;
  out.diffuse = (o.albedo * o.coatTint) * (cd * base * diel * (1.0f - Fd) * (1.0f - o.tw));
  out.pdf = Fc * pc + base * (o.metalness * ps + diel * (Fd * ps + (1.0f - Fd) * (1.0f - o.tw) * cd));
}

constexpr uint32_t KLASS_DYNAMIC = 0xffffffffu; // read the class from the material record (debug / AOV paths)
template <uint32_t KLASS>
__device__ inline void bsdf_sample(const MaterialRec* m, const ShState& st, V3 k1, float x0, float x1, float x2, BsdfSample& out)
{
  out.event = EV_ABSORB; out.pdf = 0.0f; out.overPdf = v3(0.0f, 0.0f, 0.0f); out.k2 = v3(0.0f, 0.0f, 0.0f);
  const uint32_t klass = (KLASS == KLASS_DYNAMIC) ? m->klass : KLASS;
  if (klass == 0u) {
    V3 l = gi_sample_hemisphere(x0, x1);
    V3 k2 = to_world(st, l);
    if (!(l.z > 0.0f) || !(dot(k2, st.geomNormal) > 0.0f)) return;
    out.k2 = k2; out.pdf = l.z / GI_PI; out.overPdf = v3(m->p[0], m->p[1], m->p[2]); out.event = EV_DIFFUSE | EV_REFLECTION;
    return;
  }
  if (klass == 1u) {
    UpsParams u = ups_params(m, st);
    V3 l1 = to_local(st, k1);
    float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
    float z = x2;
    float Fc = u.coat * (0.04f + 0.96f * schlick_w(nk1));
    if (z < Fc) {
      GgxOut g = ggx_sample(l1, u.coatAlpha, x0, x1);
      V3 k2 = to_world(st, g.l2);
      if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
      float Fh = u.coat * (0.04f + 0.96f * schlick_w(g.kh));
      float w = (Fh / Fc) * g.g2OverG1;
      out.k2 = k2; out.pdf = Fc * g.pdf; out.overPdf = v3(w, w, w); out.event = EV_GLOSSY | EV_REFLECTION;
      return;
    }
    z = (z - Fc) / (1.0f - Fc);
    V3 Fs = schlick3(u.F0, nk1);
    float ps = fmax2(Fs.x, fmax2(Fs.y, Fs.z));
    if (z < ps) {
      GgxOut g = ggx_sample(l1, u.alpha, x0, x1);
      V3 k2 = to_world(st, g.l2);
      if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
      V3 Fh = schlick3(u.F0, g.kh);
      out.k2 = k2; out.pdf = (1.0f - Fc) * ps * g.pdf; out.overPdf = Fh * (g.g2OverG1 / ps); out.event = EV_GLOSSY | EV_REFLECTION;
      return;
    }
    V3 l = gi_sample_hemisphere(x0, x1);
    V3 k2 = to_world(st, l);
    if (!(l.z > 0.0f) || !(dot(k2, st.geomNormal) > 0.0f)) return;
    out.k2 = k2; out.pdf = (1.0f - Fc) * (1.0f - ps) * (l.z / GI_PI);
    out.overPdf = (u.albedo * (v3(1.0f, 1.0f, 1.0f) - Fs)) * (1.0f / (1.0f - ps));
    out.event = EV_DIFFUSE | EV_REFLECTION;
    return;
  }
  if (klass == 2u) { opbr_sample(m, st, k1, x0, x1, x2, out); return; }
}

template <uint32_t KLASS>
__device__ inline void bsdf_evaluate(const MaterialRec* m, const ShState& st, V3 k1, V3 k2, BsdfEval& out)
{
  out.diffuse = v3(0.0f, 0.0f, 0.0f); out.glossy = v3(0.0f, 0.0f, 0.0f); out.pdf = 0.0f;
  float nk2 = dot(st.normal, k2);
  if (!(nk2 > 0.0f)) return;
  const uint32_t klass = (KLASS == KLASS_DYNAMIC) ? m->klass : KLASS;
  if (klass == 0u) {
    float c = nk2 / GI_PI;
    out.diffuse = v3(m->p[0], m->p[1], m->p[2]) * c; out.pdf = c;
    return;
  }
  if (klass == 1u) {
    UpsParams u = ups_params(m, st);
    V3 l1 = to_local(st, k1), l2 = to_local(st, k2);
    float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
    float Fc = u.coat * (0.04f + 0.96f * schlick_w(nk1));
    V3 Fs = schlick3(u.F0, nk1);
    float ps = fmax2(Fs.x, fmax2(Fs.y, Fs.z));
    float fc, pc, khc; ggx_eval(l1, l2, u.coatAlpha, fc, pc, khc);
    float fs, pss, khs; ggx_eval(l1, l2, u.alpha, fs, pss, khs);
    float Fch = u.coat * (0.04f + 0.96f * schlick_w(khc));
    V3 Fsh = schlick3(u.F0, khs);
    float cd = l2.z / GI_PI;
    out.glossy = v3(Fch * fc, Fch * fc, Fch * fc) + (Fsh * fs) * (1.0f - Fc);
    out.diffuse = (u.albedo * (v3(1.0f, 1.0f, 1.0f) - Fs)) * (cd * (1.0f - Fc));
    out.pdf = Fc * pc + (1.0f - Fc) * (ps * pss + (1.0f - ps) * cd);
    return;
  }
  if (klass == 2u) { opbr_evaluate(m, st, k1, k2, out); return; }
}

// ------------------------------------------------------------------------------------------------
// Light sampling (rp_main.chit:30-129)
// ------------------------------------------------------------------------------------------------
__device__ inline void sample_light(const SceneView& sc, const FrameUniforms& U, float k0, float k1, float k2, float k3, V3 surfacePos,
                                    V3& dirToLight, float& dist, V3& power, float& invPdf, uint32_t& dsPacked)
{
  const float sel = k0 * (float)U.totalLightCount;
  if (sel <= (float)U.sphereCount) {
    uint32_t idx = (uint32_t)(k1 * (float)U.sphereCount);
    const uint32_t last = U.sphereCount - 1u; if (idx > last) idx = last;
    V3 pos = v3(0.0f, 0.0f, 0.0f), em = pos, radius = pos; float area = 0.0f; dsPacked = 0u;
    if (idx < U.sphereCount) { const SphereLightRec l = sc.sphereLights[idx]; pos = v3(l.pos); em = v3(l.em); radius = v3(l.radius); area = l.area; dsPacked = l.ds; }
    V3 samplePos = pos + gi_sample_sphere(k2, k3, radius);
    V3 dir = samplePos - surfacePos;
    dist = length(dir);
    dirToLight = gi_safe_div(dir, dist);
    V3 ln = normalize(samplePos - pos);
    float cosTheta = fmax2(0.0f, dot(-dirToLight, ln));
    invPdf = gi_safe_div((area > 0.0f) ? (area * cosTheta) : 1.0f, dist * dist);
    power = em * U.lightIntensityMultiplier;
  } else if (sel <= (float)(U.sphereCount + U.distantCount)) {
    uint32_t idx = (uint32_t)(k1 * (float)U.distantCount);
    const uint32_t last = U.distantCount - 1u; if (idx > last) idx = last;
    const DistantLightRec l = sc.distantLights[idx];
    dist = 100000.0f; dirToLight = -v3(l.dir);
    power = v3(l.em) * U.lightIntensityMultiplier; invPdf = l.invPdf; dsPacked = l.ds;
    if (l.angle > 0.0f) {
      V3 t1, t2; gi_orthonormal_basis(dirToLight, t1, t2);
      float phi = (k2 * 2.0f * GI_PI) - GI_PI;
      float theta = k3 * l.angle;
      float sp, cp, stt, ct; gi_sincosr(phi, &sp, &cp); gi_sincosr(theta, &stt, &ct);
      dirToLight = normalize((t1 * cp + t2 * sp) * stt + dirToLight * ct);
    }
  } else if (sel <= (float)(U.sphereCount + U.distantCount + U.rectCount)) {
    uint32_t idx = (uint32_t)(k1 * (float)U.rectCount);
    const uint32_t last = U.rectCount - 1u; if (idx > last) idx = last;
    const RectLightRec l = sc.rectLights[idx];
    float sx = (k2 - 0.5f) * l.width, sy = (k3 - 0.5f) * l.height;
    V3 t0 = gi_decode_direction(l.t0), t1 = gi_decode_direction(l.t1);
    V3 samplePos = (v3(l.origin) + t0 * sx) + t1 * sy;
    V3 dir = samplePos - surfacePos;
    dist = length(dir); dirToLight = gi_safe_div(dir, dist);
    V3 ln = cross(t1, t0);
    float cosTheta = fmax2(0.0f, dot(-dirToLight, ln));
    float area = l.width * l.height;
    invPdf = gi_safe_div((area > 0.0f) ? (area * cosTheta) : 1.0f, dist * dist);
    power = v3(l.em) * U.lightIntensityMultiplier; dsPacked = l.ds;
  } else {
    uint32_t idx = (uint32_t)(k1 * (float)U.diskCount);
    const uint32_t last = U.diskCount - 1u; if (idx > last) idx = last;
    const DiskLightRec l = sc.diskLights[idx];
    float sx, sy; gi_sample_disk(k2, k3, l.rx, l.ry, sx, sy);
    V3 t0 = gi_decode_direction(l.t0), t1 = gi_decode_direction(l.t1);
    V3 samplePos = (v3(l.origin) + t0 * sx) + t1 * sy;
    V3 dir = samplePos - surfacePos;
    dist = length(dir); dirToLight = gi_safe_div(dir, dist);
    V3 ln = cross(t1, t0);
    float cosTheta = fmax2(0.0f, dot(-dirToLight, ln));
    float area = l.rx * l.ry * GI_PI;
    invPdf = gi_safe_div((area > 0.0f) ? (area * cosTheta) : 1.0f, dist * dist);
    power = v3(l.em) * U.lightIntensityMultiplier; dsPacked = l.ds;
  }
  power = power * U.exposureScale;
  invPdf = invPdf * (float)U.totalLightCount;
}

// ------------------------------------------------------------------------------------------------
// k_shade: closest-hit shading + the post-trace part of the bounce loop, over the HIT queue only
// (rp_main.chit:132-493, rp_main.rgen:397-480).  Misses never get here (k_trace routes them to k_raygen).
// ------------------------------------------------------------------------------------------------
template <uint32_t KLASS, bool TEXTURED, bool VOLUME>
__global__ __launch_bounds__(BLOCK) void k_shade(FrameUniforms U, SceneView sc, PathState st, QueueSet qs, Counters* cnt, uint32_t par)
{
  __shared__ AppendScratch<3> sh;
  const uint32_t qNext = Q_TRACE_A + (par ^ 1u), qRegen = Q_REGEN_A + (par ^ 1u), qHit = Q_HIT + KLASS;
  QueueReader rdr; reader_init(rdr, cnt, qHit, qs.cap);
  const uint32_t n = rdr.pre[NSHARD];
  const uint32_t stride = gridDim.x * BLOCK;
  uint32_t trip = 0;
  for (uint32_t base = blockIdx.x * BLOCK; base < n; base += stride, trip++) {
    const uint32_t i = base + threadIdx.x;
    bool cont = false, ended = false, shadow = false; uint32_t slot = 0;
    V3 no = v3(0.0f, 0.0f, 0.0f), k2 = no, sdir = no, nee = no; float ld = 0.0f, tMaxNext = GI_FLT_MAX;
    if (i < n) {
      const uint32_t r = reader_index(rdr, i);
      slot = qs.slot[qHit][r];
      const F4 h = ld4(&qs.a[qHit][r]);
      const F4 rd = ld4(&qs.b[qHit][r]);
      Slot* S = &st.slots[slot];
      const F4 tb = ld4(&S->thr);
      const F4 rr = ld4(&S->rad);
      V3 throughput = v3(tb.x, tb.y, tb.z), radiance = v3(rr.x, rr.y, rr.z);
      uint32_t bitfield = f2u(tb.w), rng = f2u(rr.w);
      const uint32_t bounce = bitfield & 0x00000fffu;

      const V3 rayDir = v3(rd.x, rd.y, rd.z);
      const uint32_t stackSize = VOLUME ? (U.mediumStackSize < MAX_MEDIUM_STACK ? U.mediumStackSize : MAX_MEDIUM_STACK) : 0u;
      float* M = VOLUME ? st.media + (size_t)slot * st.mediaStride : nullptr; // this path's medium stack + walkSegmentPdf
      uint32_t mediumIdx = payload_medium_idx(bitfield, stackSize);
      if (VOLUME && f2u(h.w) == VOLUME_MISS) {
        // the segment ended inside a medium before reaching a surface: scattering event (stepVolume, rp_main.miss:16-34).
        // Record: h = (tMax, origin.x, origin.y, -), rd = (dir, origin.z)
        const float* m = M + (mediumIdx - 1u) * MEDIUM_FLOATS;
        const float* wp = M + stackSize * MEDIUM_FLOATS;
        const float distance = h.x * U.metersPerSceneUnit;
        const V3 sigS = v3(m[2], m[3], m[4]), sigT = v3(m[5], m[6], m[7]);
        const V3 tr = v3(gi_expf(sigT.x * -distance), gi_expf(sigT.y * -distance), gi_expf(sigT.z * -distance));
        const V3 density = sigT * tr;
        const float pdf = dot(v3(wp[0], wp[1], wp[2]), density);
        throughput = throughput * ((sigS * tr) / pdf);
        no = v3(h.y, h.z, rd.w) + rayDir * distance;
        k2 = rayDir;
        bitfield |= 0x40000000u; // SHADE_RAY_PAYLOAD_VOLUME_WALK_MISS_FLAG
        payload_increment_walk(bitfield);
      } else {
      ShState ss;
      setup_shading_state(sc, f2u(h.w), h.y, h.z, rayDir, ss);
      const MaterialRec* mat = &sc.materials[ss.material];
      if (TEXTURED && (mat->flags & MAT_FLAG_TEXTURED)) resolve_material_textures(sc, mat, rayDir, ss); // else ss.texMask stays 0 and folds away
      const bool isDoubleSided = (ss.meshFlags & 2u) != 0u;
      // volume attenuation (rp_main.chit:160-186)
      float prevMediumIor = 1.0f, nextMediumIor = 1.0f;
      if (mediumIdx > 0u) {
        const float distance = h.x * U.metersPerSceneUnit;
        if (!VOLUME) { // empty medium stack: inside (1-bit toggle) -> Beer-Lambert with the HIT material's absorption coefficient (:169-173)
          if (KLASS == 2u) throughput = throughput * v3(gi_expf(-mat->p[MP_SIGMA_A] * distance), gi_expf(-mat->p[MP_SIGMA_A + 1] * distance), gi_expf(-mat->p[MP_SIGMA_A + 2] * distance));
        } else { // the medium on top of the stack (:174-184)
          const float* m = M + (mediumIdx - 1u) * MEDIUM_FLOATS;
          prevMediumIor = m[0];
          if (mediumIdx > 1u) nextMediumIor = M[(mediumIdx - 2u) * MEDIUM_FLOATS];
          throughput = throughput * v3(gi_expf(-m[5] * distance), gi_expf(-m[6] * distance), gi_expf(-m[7] * distance));
        }
      }
      if (VOLUME) { ss.ior1 = ss.frontFace ? prevMediumIor : -1.0f; ss.ior2 = ss.frontFace ? -1.0f : nextMediumIor; } // iorCurrent / iorOther (:188-189)
      // emission (rp_main.chit:293-343): uniform EDF, radiance == emission colour where cos > 0
      const V3 em = (ss.texMask & (1u << TEX_EMISSION)) ? ss.texEmission : v3(mat->p[3], mat->p[4], mat->p[5]);
      if (em.x != 0.0f || em.y != 0.0f || em.z != 0.0f) {
        if (ss.frontFace || !isDoubleSided) {
          const float c = dot(-rayDir, ss.normal);
          if (c > 0.0f) radiance = radiance + throughput * (em * U.exposureScale);
        }
      }
      // BSDF importance sampling (:361-389); xi = next4f, .w is drawn but unused by the closed forms
      const float x0 = gi_next1f(rng), x1 = gi_next1f(rng), x2 = gi_next1f(rng); (void)gi_next1f(rng);
      BsdfSample bs; bsdf_sample<KLASS>(mat, ss, -rayDir, x0, x1, x2, bs);
      throughput = throughput * bs.overPdf;
      k2 = bs.k2;
      const bool isTransmission = (bs.event & EV_TRANSMISSION) != 0u;
      // NEE (:394-444)
      if ((U.flags & FLAG_NEE) && (bs.event & (EV_DIFFUSE | EV_GLOSSY))) {
        const float k0 = gi_next1f(rng), k1 = gi_next1f(rng), kk2 = gi_next1f(rng), k3 = gi_next1f(rng);
        V3 dirToLight, lightPower; float lightDist, invPdf; uint32_t ds;
        sample_light(sc, U, k0, k1, kk2, k3, ss.position, dirToLight, lightDist, lightPower, invPdf, ds);
        if ((lightDist > 0.0f) && dot(dirToLight, ss.geomNormal) > 0.0f) {
          BsdfEval ev; bsdf_evaluate<KLASS>(mat, ss, -rayDir, dirToLight, ev);
          if (ev.pdf > 0.0f) {
            const float dmul = gi_half_to_float(ds & 0xffffu), smul = gi_half_to_float(ds >> 16);
            const V3 weight = throughput * (lightPower * invPdf);
            nee = nee + (weight * ev.diffuse) * dmul;
            nee = nee + (weight * ev.glossy) * smul;
          }
        }
        // rp_main.rgen:401-408: the shadow ray is traced only if it can contribute
        const V3 toLight = dirToLight * lightDist;
        ld = length(toLight);
        sdir = gi_safe_div(toLight, ld);
        shadow = gi_luminance(nee) > 1e-6f && ld > 1e-9f;
      }
      if (isTransmission) { // medium stack (:447-480)
        uint32_t newIdx = mediumIdx;
        if (VOLUME) {
          if (ss.frontFace) { // push the material's medium: mdl_ior, mdl_volume_{scattering,absorption}_coefficient, MEDIUM_DIRECTIONAL_BIAS
            newIdx = mediumIdx + 1u;
            if (newIdx <= stackSize) {
              float* m = M + (newIdx - 1u) * MEDIUM_FLOATS;
              if (KLASS == 2u) {
                const float depth = mat->p[28];
                const V3 sigS = (depth > 0.0f) ? v3(mat->p[29] / depth, mat->p[30] / depth, mat->p[31] / depth) : v3(0.0f, 0.0f, 0.0f);
                const V3 sigT = v3(mat->p[MP_SIGMA_A], mat->p[MP_SIGMA_A + 1], mat->p[MP_SIGMA_A + 2]) + sigS;
                m[0] = mat->p[MP_ETA]; m[1] = mat->p[47]; m[2] = sigS.x; m[3] = sigS.y; m[4] = sigS.z; m[5] = sigT.x; m[6] = sigT.y; m[7] = sigT.z;
              } else { m[0] = 1.0f; m[1] = 0.0f; m[2] = 0.0f; m[3] = 0.0f; m[4] = 0.0f; m[5] = 0.0f; m[6] = 0.0f; m[7] = 0.0f; }
            }
          } else if (mediumIdx > 0u) newIdx = mediumIdx - 1u; // pop
        } else newIdx = 1u - mediumIdx; // MEDIUM_STACK_SIZE == 0: toggle between inside and outside
        bitfield &= ~0x00fff000u; // medium changed -> reset walk
        bitfield = (bitfield & ~0x0f000000u) | ((newIdx << 24) & 0x0f000000u);
      }
      if (bs.event == EV_ABSORB) bitfield |= 0x80000000u; // :483-486
      const V3 gn = ss.geomNormal * (isTransmission ? -1.0f : 1.0f);
      no = gi_offset_ray_origin(ss.position, gn); // :488-489
      }
      // rp_main.rgen:441-480
      if (length(throughput) < 1e-9f) bitfield |= 0x80000000u;
      if (bounce > U.rrBounceOffset) {
        const float k = gi_next1f(rng);
        const float mt = fmax2(throughput.x, fmax2(throughput.y, throughput.z));
        const float p = fmin2(mt, U.rrInvMinTermProb);
        if (k > p) bitfield |= 0x80000000u; else throughput = throughput / p;
      }
      if (VOLUME && (bitfield & 0x40000000u)) { // :462-477: the random walk continues in a Henyey-Greenstein direction
        const float x0 = gi_next1f(rng), x1 = gi_next1f(rng);
        const float g = M[(mediumIdx - 1u) * MEDIUM_FLOATS + 1u];
        float cosTheta;
        if (fabsf(g) < 1e-3f) cosTheta = 1.0f - 2.0f * x0;
        else { const float sq = (1.0f - g * g) / ((1.0f - g) + (2.0f * g) * x0); cosTheta = ((1.0f + g * g) - sq * sq) / (2.0f * g); }
        const float sinTheta = sqrtf(fmax2(0.0f, 1.0f - cosTheta * cosTheta));
        float sp, cp; gi_sincos2pi(x1, &sp, &cp);
        V3 t, b; gi_orthonormal_basis(k2, t, b);
        k2 = ((t * sinTheta) * cp + (b * sinTheta) * sp) + k2 * cosTheta;
        bitfield &= ~0x40000000u;
      }
      bitfield++;
      cont = ((bitfield & 0x00000fffu) < U.maxBounces) && !(bitfield & 0x80000000u); // loop test :298-304
      ended = !cont;
      if (VOLUME && cont) { // top of the next loop iteration (:317-346): distance to the next collision inside a scattering medium
        const uint32_t idx2 = payload_medium_idx(bitfield, stackSize);
        if (idx2 > 0u) {
          const float* m = M + (idx2 - 1u) * MEDIUM_FLOATS;
          float* wp = M + stackSize * MEDIUM_FLOATS;
          V3 wpdf = v3(1.0f, 1.0f, 1.0f);
          const uint32_t walkLength = (bitfield & 0x00fff000u) >> 12;
          const V3 sigS = v3(m[2], m[3], m[4]), sigT = v3(m[5], m[6], m[7]);
          if ((sigS.x > 0.0f || sigS.y > 0.0f || sigS.z > 0.0f) && walkLength <= U.maxVolumeWalkLength) {
            const V3 albedo = v3(gi_safe_div(sigS.x, sigT.x), gi_safe_div(sigS.y, sigT.y), gi_safe_div(sigS.z, sigT.z));
            const float x0 = gi_next1f(rng), x1 = gi_next1f(rng);
            const V3 weights = throughput * albedo; // sampleDistance (:49-69)
            const float sum = (weights.x + weights.y) + weights.z;
            wpdf = (sum > 1e-9f) ? (weights / sum) : v3(1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 3.0f);
            float sg = (x0 < wpdf.x) ? sigT.x : ((x0 < (wpdf.x + wpdf.y)) ? sigT.y : sigT.z);
            sg = sg * U.metersPerSceneUnit;
            tMaxNext = -gi_logf(1.0f - x1) / sg;
          }
          wp[0] = wpdf.x; wp[1] = wpdf.y; wp[2] = wpdf.z;
        }
      }
      st4(&S->thr, throughput.x, throughput.y, throughput.z, u2f(bitfield));
      st4(&S->rad, radiance.x, radiance.y, radiance.z, u2f(rng));
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
      st4(&qs.b[Q_SHADOW][idx[2]], sdir.x, sdir.y, sdir.z, 0.0f);
      st4(&qs.c[Q_SHADOW][idx[2]], nee.x, nee.y, nee.z, 0.0f);
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
  float Fc = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(nk1));
  float Fd = fresnel_dielectric(nk1, eta);
  float base = 1.0f - Fc, diel = 1.0f - o.metalness;
  V3 diffuse = (o.albedo * o.coatTint) * (base * diel * (1.0f - Fd) * (1.0f - o.tw));
  V3 glossy = v3(Fc, Fc, Fc) + ((schlick_f82(o.albedo, o.metalTint, nk1) * o.specWeight) * o.coatTint) * (base * o.metalness)
              + (o.specColor * o.coatTint) * (base * diel * Fd);
  return diffuse + glossy;
}

template <uint32_t STACK, bool OVERFLOW>
__global__ __launch_bounds__(TRACE_BLOCK) void k_aov(FrameUniforms U, SceneView sc, AovTargets A, uint32_t ldsNodes, uint32_t ldsTris)
{
  extern __shared__ uint4 s_dyn[];
  uint2 (*s_stack)[TRACE_BLOCK] = reinterpret_cast<uint2 (*)[TRACE_BLOCK]>(s_dyn);
  uint4* s_nodes = s_dyn + (STACK * TRACE_BLOCK * sizeof(uint2)) / sizeof(uint4);
  uint4* s_tris = s_nodes + ldsNodes * 5u;
  for (uint32_t i = threadIdx.x; i < ldsNodes * 5u; i += TRACE_BLOCK) s_nodes[i] = reinterpret_cast<const uint4*>(sc.nodes)[(i / 5u) * sc.nodeStrideU4 + (i % 5u)];
  for (uint32_t i = threadIdx.x; i < ldsTris * 3u; i += TRACE_BLOCK) s_tris[i] = reinterpret_cast<const uint4*>(sc.tris)[(i / 3u) * 4u + (i % 3u)];
  __syncthreads();
  const uint32_t p = blockIdx.x * TRACE_BLOCK + threadIdx.x;
  if (p >= U.pixelCount) return;
  const uint32_t pixelIndex = U.rowBegin * U.imageWidth + p;
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
    uint32_t matUnused;
    if (!traverse<false, false, STACK, OVERFLOW, false, true>(sc, s_nodes, ldsNodes, s_tris, ldsTris, s_stack, origin, dir, tMin, tMax, t, u, v, tri, matUnused, tc, rng)) continue;
    ShState ss;
    setup_shading_state(sc, tri, u, v, dir, ss);
    const uint4* tp = reinterpret_cast<const uint4*>(sc.tris) + (size_t)tri * 4u;
    const uint32_t instIdx = tp[2].z;
    put3(A.opacity, v3(1.0f, 0.0f, 0.0f));
    put3(A.tangents, (ss.tangentU + v3(1.0f, 1.0f, 1.0f)) * 0.5f);
    put3(A.bitangents, (ss.tangentV + v3(1.0f, 1.0f, 1.0f)) * 0.5f);
    put3(A.barycentrics, v3(1.0f - u - v, u, v));
    if (A.texcoords) {
      const uint4 td = tp[3];
      const float bx = 1.0f - u - v;
      const FVertex* va = &sc.verts[td.x]; const FVertex* vb = &sc.verts[td.y]; const FVertex* vc = &sc.verts[td.z];
      put3(A.texcoords, v3((bx * va->u + u * vb->u) + v * vc->u, (bx * va->v + u * vb->v) + v * vc->v, 0.0f));
    }
    put3(A.thinWalled, v3(0.0f, 1.0f, 0.0f));
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
__global__ void k_debug_bsdf(const MaterialRec* mat, uint32_t count, const float* __restrict__ in, float* __restrict__ out)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float* p = in + 22 * (size_t)i; float* o = out + 15 * (size_t)i;
  ShState st; st.normal = v3(p); st.tangentU = v3(p + 3); st.tangentV = v3(p + 6); st.geomNormal = v3(p + 9);
  st.position = v3(0.0f, 0.0f, 0.0f); st.frontFace = (p[21] < 0.5f); st.meshFlags = 0u; st.material = 0u;
  st.u = 0.0f; st.v = 0.0f; st.texMask = 0u; st.ior1 = 0.0f; st.ior2 = 0.0f; st.mesh = 0u; st.prim = 0u; st.vi[0] = st.vi[1] = st.vi[2] = 0u; st.instanceId = 0; st.hu = st.hv = 0.0f;
  BsdfSample bs; bsdf_sample<KLASS_DYNAMIC>(mat, st, v3(p + 12), p[18], p[19], p[20], bs);
  BsdfEval ev; bsdf_evaluate<KLASS_DYNAMIC>(mat, st, v3(p + 12), v3(p + 15), ev);
  o[0] = bs.k2.x; o[1] = bs.k2.y; o[2] = bs.k2.z; o[3] = bs.overPdf.x; o[4] = bs.overPdf.y; o[5] = bs.overPdf.z; o[6] = bs.pdf; o[7] = (float)bs.event;
  o[8] = ev.diffuse.x; o[9] = ev.diffuse.y; o[10] = ev.diffuse.z; o[11] = ev.glossy.x; o[12] = ev.glossy.y; o[13] = ev.glossy.z; o[14] = ev.pdf;
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
                               uint32_t dynRefill, uint32_t routeBlocks)
{
  uint32_t ln, lt, bytes; traceLdsLayout(sc, ln, lt, bytes);
  const bool allLds = ln == sc.nodeCount && lt == sc.triCount && sc.triCount > 0u; // the whole scene is staged in LDS
  if (!allLds && dynRefill) { // big scene: persistent waves with dynamic ray fetch, results routed by a streaming pass
    // persistent waves pay the scratch set-up once, so trees deeper than 8 levels may keep 8 entries in LDS (more
    // resident waves) and spill the rest (TRACE_DYN_SPILL8), or keep 16 in LDS
    const bool spill8 = (dynRefill & TRACE_DYN_SPILL8) != 0u;
    dynRefill &= 0xffu;
    const uint32_t entries = (sc.bvhDepth <= 8u || spill8) ? 8u : 16u;
    const uint32_t stackBytes = entries * TRACE_BLOCK * (uint32_t)sizeof(uint2);
    if (sc.bvhDepth <= 8u) hipLaunchKernelGGL((k_trace_dyn<ANYHIT, COUNT, 8, false, CUTOUT>), dim3(blocks), dim3(TRACE_BLOCK), stackBytes, s, sc, st, qs, cnt, qIn, dynRefill);
    else if (spill8) hipLaunchKernelGGL((k_trace_dyn<ANYHIT, COUNT, 8, true, CUTOUT>), dim3(blocks), dim3(TRACE_BLOCK), stackBytes, s, sc, st, qs, cnt, qIn, dynRefill);
    else if (sc.bvhDepth <= 16u) hipLaunchKernelGGL((k_trace_dyn<ANYHIT, COUNT, 16, false, CUTOUT>), dim3(blocks), dim3(TRACE_BLOCK), stackBytes, s, sc, st, qs, cnt, qIn, dynRefill);
    else hipLaunchKernelGGL((k_trace_dyn<ANYHIT, COUNT, 16, true, CUTOUT>), dim3(blocks), dim3(TRACE_BLOCK), stackBytes, s, sc, st, qs, cnt, qIn, dynRefill);
    if (!ANYHIT) hipLaunchKernelGGL(k_route, dim3(routeBlocks), dim3(BLOCK), 0, s, sc, st, qs, cnt, qIn, qMiss);
    return;
  }
  const bool dome = !ANYHIT && (sc.domeTexture != 0u || sc.mediumStackSize != 0u); // misses need the slot: dome image lookup / scattering events
#define GI_LAUNCH_TRACE(STACK, OVF, LDS) do { \
    if (dome) hipLaunchKernelGGL((k_trace<ANYHIT, COUNT, STACK, OVF, LDS, CUTOUT, !ANYHIT>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, sc, st, qs, cnt, qIn, qMiss, ln, lt); \
    else hipLaunchKernelGGL((k_trace<ANYHIT, COUNT, STACK, OVF, LDS, CUTOUT, false>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, sc, st, qs, cnt, qIn, qMiss, ln, lt); } while (0)
  if (allLds && sc.bvhDepth <= 4u) GI_LAUNCH_TRACE(4, false, true);
  else if (allLds && sc.bvhDepth <= 8u) GI_LAUNCH_TRACE(8, false, true);
  else if (sc.bvhDepth <= 8u) GI_LAUNCH_TRACE(8, false, false);
  else if (sc.bvhDepth <= 16u) GI_LAUNCH_TRACE(16, false, false);
  else GI_LAUNCH_TRACE(16, true, false);
#undef GI_LAUNCH_TRACE
}
template <bool ANYHIT, bool COUNT>
static void launchTraceCutout(hipStream_t s, uint32_t blocks, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t qIn, uint32_t qMiss,
                              uint32_t dynRefill, uint32_t routeBlocks)
{
  if (sc.hasCutouts) launchTraceVariant<ANYHIT, COUNT, true>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks);
  else launchTraceVariant<ANYHIT, COUNT, false>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks);
}
void launchTrace(hipStream_t s, uint32_t blocks, bool anyHit, bool count, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt,
                 uint32_t qIn, uint32_t qMiss, uint32_t dynRefill, uint32_t routeBlocks)
{
  if ((dynRefill & 0xffu) > 64u) dynRefill = (dynRefill & ~0xffu) | 64u;
  if (!anyHit) { if (count) launchTraceCutout<false, true>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks); else launchTraceCutout<false, false>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks); }
  else { if (count) launchTraceCutout<true, true>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks); else launchTraceCutout<true, false>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks); }
}
void launchAov(hipStream_t s, const FrameUniforms& U, const SceneView& sc, const AovTargets& A)
{
  uint32_t ln, lt, bytes; traceLdsLayout(sc, ln, lt, bytes);
  bytes = (sc.bvhDepth <= 8u ? 8u : 16u) * TRACE_BLOCK * (uint32_t)sizeof(uint2) + ln * 80u + lt * 48u; // k_aov has no 4-entry variant
  const uint32_t blocks = (U.pixelCount + TRACE_BLOCK - 1u) / TRACE_BLOCK;
  if (sc.bvhDepth <= 8u) hipLaunchKernelGGL((k_aov<8, false>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, U, sc, A, ln, lt);
  else if (sc.bvhDepth <= 16u) hipLaunchKernelGGL((k_aov<16, false>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, U, sc, A, ln, lt);
  else hipLaunchKernelGGL((k_aov<16, true>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, U, sc, A, ln, lt);
}
void launchShade(hipStream_t s, uint32_t blocks, uint32_t klass, bool textured, bool volume, const FrameUniforms& U, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t par)
{
#define GI_LAUNCH_SHADE(K) do { \
    if (volume) { if (textured) hipLaunchKernelGGL((k_shade<K, true, true>), dim3(blocks), dim3(BLOCK), 0, s, U, sc, st, qs, cnt, par); \
                  else hipLaunchKernelGGL((k_shade<K, false, true>), dim3(blocks), dim3(BLOCK), 0, s, U, sc, st, qs, cnt, par); } \
    else if (textured) hipLaunchKernelGGL((k_shade<K, true, false>), dim3(blocks), dim3(BLOCK), 0, s, U, sc, st, qs, cnt, par); \
    else hipLaunchKernelGGL((k_shade<K, false, false>), dim3(blocks), dim3(BLOCK), 0, s, U, sc, st, qs, cnt, par); } while (0)
  if (klass == 0u) GI_LAUNCH_SHADE(0u); else if (klass == 1u) GI_LAUNCH_SHADE(1u); else GI_LAUNCH_SHADE(2u);
#undef GI_LAUNCH_SHADE
}

void launchResolveNee(hipStream_t s, const unsigned long long* key, F4* aov, uint32_t pixelCount, uint32_t firstPixel)
{
  hipLaunchKernelGGL(k_resolve_nee, dim3((pixelCount + 255u) / 256u), dim3(256), 0, s, key, aov, pixelCount, firstPixel);
}

void launchDebugBsdf(hipStream_t s, const MaterialRec* mat, uint32_t count, const float* in, float* out)
{
  hipLaunchKernelGGL(k_debug_bsdf, dim3((count + 63u) / 64u), dim3(64), 0, s, mat, count, in, out);
}

} // namespace gi
