// gi_trace.hip -- the traversal kernels of the wavefront path tracer (gfx950): k_trace (block-synchronous, scenes staged in LDS), k_trace_dyn / k_trace_dyn2
// (persistent waves with dynamic ray fetch for scenes that do not fit; results routed by gi_kernels.hip k_route).  They replace the two traceRayEXT calls of
// the reference's ray generation shader (/root/reference/src/gi/shaders/rp_main.rgen:381-393, 412-424: closest hit and shadow test; hardware BVH traversal
// there).  The walk itself is gi_traversal.h.  Built with -ffp-contract=off (arithmetic contract, gi_device_math.h); the box tests use explicit fmaf: they are
// conservative filters and never influence results.

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

template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool ALL_LDS, bool CUTOUT, bool DOME>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace(SceneView sc, PathState st, QueueSet qs, Counters* cnt, uint32_t qIn, uint32_t qMiss, uint32_t ldsNodes,
    uint32_t ldsTris,
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
      // the any-hit test needs the path's rng state (shadow rays carry their copy)
      rng = CUTOUT ? (ANYHIT ? f2u(rdir.w) : (fresh ? qs.fresh[qIn - Q_TRACE_A][r].rng : f2u(st.slots[slot].rad.w))) : 0u;
      // shadow ray (rp_main.rgen:397-429): origin = next ray origin, tMin 0.01, tMax = distance to the light sample
      if (!ANYHIT) trav_init(R, v3(ro.x, ro.y, ro.z), v3(rdir.x, rdir.y, rdir.z), ro.w, rdir.w);
      else trav_init(R, v3(ro.x, ro.y, ro.z), v3(rdir.x, rdir.y, rdir.z), 0.01f, ro.w);
      wave_ray_begin(W, R.tBest);
      alive = true;
    }
    while (__ballot(alive)) {
      if (wave_step<ANYHIT, COUNT, STACK, OVERFLOW, ALL_LDS,
          CUTOUT>(R, alive, W, sc, s_nodes, ldsNodes, s_tris, ldsTris, s_stack, overflow, tc, rng)) alive = false;
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
      // thin batches: one OpenPBR launch (same bits: gi_shading.h "BASE variant")
      if (klass == SHADE_CLASS_OPBR_BASE && (U.flags & FLAG_MERGE_SHADE_VARIANTS)) klass = 2u;
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
        // (tMax, origin) for the scattering event
        else { st4(&qs.a[qIn][r], rdir.w, ro.x, ro.y, u2f(MISS)); reinterpret_cast<float*>(&qs.b[qIn][r])[3] = ro.z; }
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
    if (__lane_id() == 0) { atomicAdd(ANYHIT ? &cnt->shadowNodesVisited : &cnt->nodesVisited, a);
        atomicAdd(ANYHIT ? &cnt->shadowTrisTested : &cnt->trisTested, b); }
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
// * shadow walks (ANYHIT) end at their first hit, so near-to-far order is optional
// for them: the SLOT instantiation visits children in slot order (no octant flip
//     in its node test) and the host launches whichever order the scene's shadow walks have been cheaper in (trace_dyn_body, gi_render.cpp shadowOrder).
// ------------------------------------------------------------------------------------------------
// bit in k_trace_dyn's `refill` argument (shadow launches): the launch is the slot-order instantiation (the prologue counts its rays as such)
constexpr uint32_t DYN_SLOT_ORDER = 0x200u;
// rays per cursor atomic (a device-scope atomic on one line completes ~88 times per microsecond; 64 / 256 / 512 measured: r04x)
constexpr uint32_t DYN_CLAIM = 128;
constexpr uint32_t DYN_FLUSH_AT = 8;  // the triangle ring is flushed below 64 pairs once this many finished walks wait for it (0 / 2 / 24 measured: r04c)
constexpr int DYN_WAVES = 5; // resident waves per SIMD the register allocation aims for (84 - 96 VGPRs).  6 waves (80 VGPRs, 3 - 8 dwords spilled) do not pay:
                             // C3 trace 55.3 -> 56.2 ms, C5 119 -> 123 (profiles/r05r_six_waves_variants.txt)
                             // -- the SIMD's issue rate is shared, more waves do not raise it
constexpr uint32_t DYN_THIN_WALKERS = 8; // the ring is flushed at the end of every step while this few lanes walk (16: the same, r05d)

template <bool TWO> struct DynRay { using type = RayWalk; };
template <> struct DynRay<true> { using type = RayTrav2; };
// wave-uniform by construction: keep it in an SGPR
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// HELP: the instantiation with helper lanes (thin launches, below); a full launch runs the one without -- the bookkeeping alone (record and ring addresses
// computed from a register instead of the lane number, the stack base) costs the full
// launches 3 - 6 % of their traversal time when it is compiled into their loop (r05g).
template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool CUTOUT, bool TWO, bool HELP, bool SLOT = false>
__device__ __forceinline__ void trace_dyn_body(const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t qIn, uint32_t refill,
    WaveTri& W,
                                               uint32_t shardCount, uint32_t claim)
{
  // Shadow walks end at their first hit, so near-to-far order is not needed for the
  // result -- and which order finds an occluder sooner depends on the scene (C3's soup:
  // slot order visits 7 % fewer nodes; C5's interior, where the occluders sit near the shaded surface: 28 % more, r05c).  The host tells the launch which one
  // to use (the SLOT instantiation: no octant flip in the node test, children are visited
  // in slot order) and the kernel counts the walks' node visits for it to choose by.
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
  // (End of a launch: every wave finds its shard dry and walks the other cursors, 8 atomics per wave.  Publishing "dry" bits on a line of their own and reading
  // them first saves those atomics and changed no launch time, r05k: not built in.)
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
        // the any-hit test needs the path's rng state (shadow rays carry their copy; a camera ray whose Slot is still unwritten has it beside the record)
        if (CUTOUT) {
          if (ANYHIT) prng = f2u(prd.w);
          else { const uint32_t sw = qs.slot[qIn][prec]; prng = (sw & TRACE_FRESH) ? qs.fresh[qIn - Q_TRACE_A][prec].rng : f2u(st.slots[sw].rad.w); }
        }
      }
    }
  };
  // A ray's walk can be shared (flat layout): once the wave has nothing left to claim, lanes without a ray HELP the longest walks instead of idling -- a helper
  // takes the bottom entry of a walking lane's traversal stack (the oldest deferred group, i.e. the largest subtree still to do), copies the ray and walks that
  // part.  All lanes working on a ray report to the same LDS record (`key`: the lane that owns the ray -- the atomicMin hit key is order-independent, so the
  // result does not change), pairs in the triangle ring name the owner (whose registers hold the ray until the end), and the owner's record counts its live
  // helpers: the ray is finished when the owner's own walk has drained and that count is zero.  Why: a launch ends with its slowest ray, a 100-step ray
  // outlives the average one six times over, and the thin launches of a low-spp frame (one sample
  // per pixel and call is hdGatling's default) are NOTHING BUT that tail.  Only in thin launches:
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
        // the result if nothing is hit: (tMax, origin.xy, MISS) -- k_route needs the origin for scattering events (medium stacks only); no helpers
        wt_hit_put(W, lane, f2u(ro.x), f2u(ro.y), MISS, 0u);
        alive = true; draining = false; lastEnd = ringHead; // no pair of this ray is pending
        keyReg = lane; base = 0u; helper = false;
      }
      if (COUNT) { ds[6]++; ds[7] += take; }
      chunkUsed += take;
      if (chunkUsed == chunkCount) next_chunk(); // loads complete while the wave keeps traversing
    // nothing in flight and nothing left to claim (an exhausted chunk is replaced at once, so chunkUsed == chunkCount means there is none)
    } else if (nIdle == 64u) break;
    else if (HELP && chunkCount == 0u && nIdle != 0u) {
      const unsigned long long donors = __ballot(alive && !draining && R.sp > base && base < STACK);
      if (donors) {
        // the k-th idle lane helps the k-th donor: donors leave their lane number in lane
        // k (forward permute; the other lanes aim at lane 63, which no helper reads:
        // with a non-donor in the wave there are at most 63 donors, ranks 0 .. 62)
        const bool donor = alive && !draining && R.sp > base && base < STACK; // (entries beyond STACK live in the lane's scratch: OVERFLOW variants)
        const uint32_t dRank = (uint32_t)__popcll(donors & ((1ull << lane) - 1ull)), tRank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
        const uint32_t compact = (uint32_t)__builtin_amdgcn_ds_permute((int)((donor ? dRank : 63u) << 2), (int)lane);
        const uint32_t nPairs = (uint32_t)__popcll(donors) < nIdle ? (uint32_t)__popcll(donors) : nIdle;
        const bool thief = !alive && tRank < nPairs;
        // (all lanes: wave-uniform control flow; only thieves use it)
        const int from = (int)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(tRank << 2), (int)compact);
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
      // The triangle ring is carried from step to step: a node step yields fewer pairs than a batch holds (C3: 23 per step, C4: 18), so flushing at the end of
      // every step ran the ~110-instruction batch at a third of its lanes.  A batch runs when 64 pairs are pending; the rest waits.  A ray whose walk has ended
      // while pairs of it are still pending is DRAINING: its lane keeps the ray (a pending pair fetches the ray from its owner lane at batch time) and sits out
      // the node phases until the ring has moved past its last pair (the ring is FIFO: `head` has reached `lastEnd`).  The ring is flushed below 64 pairs when
      // DYN_FLUSH_AT or more lanes are blocked like that, or when few lanes walk.  Results do not depend on any of this (the hit key under atomicMin does not
      // depend on when a pair is tested); only the culling distance a walking ray sees may lag by a step or two.
      auto batch = [&](uint32_t n) { wave_tri_batch<COUNT, false, CUTOUT, true>(W, ringHead, n, R, rng, sc, nullptr, 0u, tc); ringHead += n;
          if (COUNT) { ds[4]++; ds[5] += n; } };
      const bool walking = alive && !draining;
      if (COUNT) { ds[0]++; ds[1] += (unsigned long long)__popcll(__ballot(alive)); ds[2] += (unsigned long long)__popcll(__ballot(walking));
          ds[3] += (unsigned long long)__popcll(__ballot(alive && draining)); }
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
        // (more pairs than the ring has room for) one ballot round per triangle; a batch as soon as 64 pairs are pending (<= 63 + 64 <= the ring's 128 entries)
        } else for (;;) {
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
        // a shadow walk ends at the first hit
        else if (!draining && (wt_best_id(W, key) != 0u || trav_pop<STACK, OVERFLOW>(R, s_stack, overflow, HELP ? base : 0u))) draining = true;
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
            // ONE 16-byte store per finished ray: the batches left the finished record in LDS (material class in the top four bits of the triangle word; until
            // r04 the material word went into b.w as a second store into another line: C3 85 GB of write traffic per frame for 26 GB of results)
            uint32_t word = h.z;
            // scene-order id -> index of the hit's TriRec
            if constexpr (TWO) { if (word != MISS) word = sc.flatOfOrig[word & 0x0fffffffu] | (word & 0xf0000000u); }
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
// what every wave of a k_trace_dyn launch does first: the ray counts of the queue's shards (lane k keeps shard k's; read back with readlane where a range is
// entered), the launch's total (one writer adds it to the frame's statistics), and the rays
// per cursor atomic -- DYN_CLAIM while every wave of the launch can have a claim of its own;
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
  // single writer per launch
  if (blockIdx.x == 0
      && threadIdx.x == 0) { if (ANYHIT) { cnt->shadowRays += nRays; cnt->shadowOrderRays[(refill & DYN_SLOT_ORDER) ? 1 : 0] += nRays;
      } else cnt->segments += nRays; }
  claim = nRays >= gridDim.x * (TRACE_BLOCK / 64u) * DYN_CLAIM ? DYN_CLAIM : 64u;
  // Waves the launch has no chunk for leave at once, without touching the cursors: every wave that stays walks all NSHARD cursors before it gives up, and a
  // device-scope atomic on one line completes ~88 times per microsecond -- 8 192 waves x 8 cursors were a 0.1 ms floor under every launch that held any ray at
  // all, i.e. under each of the 13 thin launches of a spp-1 frame (r05: C4 233 rays, 0.146 ms).  ceil(n / 64) chunks + one ragged chunk per shard; the waves
  // that stay claim until every shard is dry.
  return (blockIdx.x * (TRACE_BLOCK / 64u) + (threadIdx.x >> 6)) < (nRays + 63u) / 64u + NSHARD;
}
template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool CUTOUT, bool SLOT = false>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(DYN_WAVES, 8))) void k_trace_dyn(SceneView sc, PathState st, QueueSet qs,
    Counters* cnt, uint32_t qIn, uint32_t refill)
{
  __shared__ WaveTri s_wave[TRACE_BLOCK / 64];
  uint32_t shardCount, claim;
  if (!trace_dyn_prologue<ANYHIT>(qs, cnt, qIn, refill, shardCount, claim)) return;
  // a THIN launch -- fewer rays than two chunks per wave -- lasts as long as its
  // slowest ray, not as its throughput allows: its idle lanes help (trace_dyn_body)
  if (claim == 64u) trace_dyn_body<ANYHIT, COUNT, STACK, OVERFLOW, CUTOUT, false, true,
      SLOT>(sc, st, qs, cnt, qIn, refill, s_wave[threadIdx.x >> 6], shardCount, claim);
  else trace_dyn_body<ANYHIT, COUNT, STACK, OVERFLOW, CUTOUT, false, false, SLOT>(sc, st, qs, cnt, qIn, refill, s_wave[threadIdx.x >> 6], shardCount, claim);
}
// the two-level layout (wave_step2): 16 LDS stack entries, world + object-space ray in registers
template <bool ANYHIT, bool COUNT, bool CUTOUT>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_trace_dyn2(SceneView sc, PathState st, QueueSet qs, Counters* cnt,
    uint32_t qIn, uint32_t refill)
{
  __shared__ WaveTri s_wave[TRACE_BLOCK / 64];
  uint32_t shardCount, claim;
  if (!trace_dyn_prologue<ANYHIT>(qs, cnt, qIn, refill, shardCount, claim)) return;
  trace_dyn_body<ANYHIT, COUNT, 16, false, CUTOUT, true, false>(sc, st, qs, cnt, qIn, refill, s_wave[threadIdx.x >> 6], shardCount, claim);
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------------
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
static void launchTraceVariant(hipStream_t s, uint32_t blocks, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t qIn,
    uint32_t qMiss,
                               uint32_t dynRefill, uint32_t routeBlocks, const FrameUniforms& U, F4* sampleBuf)
{
  uint32_t ln, lt, bytes; traceLdsLayout(sc, ln, lt, bytes);
  const bool allLds = ln == sc.nodeCount && lt == sc.triCount && sc.triCount > 0u; // the whole scene is staged in LDS
  if (!allLds && dynRefill) { // big scene: persistent waves with dynamic ray fetch, results routed by a streaming pass
    // persistent waves pay the scratch set-up once, so trees deeper than 8 levels may keep 8 entries in LDS (more
    // resident waves) and spill the rest (TRACE_DYN_SPILL8), or keep 16 in LDS
    const uint32_t refill = (dynRefill & 0xffu) | (ANYHIT ? (dynRefill & DYN_SLOT_ORDER) : 0u);
    if (sc.twoLevel) { // instanced scene: TLAS + shared per-mesh BLASes
      hipLaunchKernelGGL((k_trace_dyn2<ANYHIT, COUNT, CUTOUT>), dim3(blocks), dim3(TRACE_BLOCK), 16u * TRACE_BLOCK * (uint32_t)sizeof(uint2), s, sc, st, qs,
          cnt, qIn, refill);
      if (!ANYHIT) launchRoute(s, routeBlocks, sc, st, qs, cnt, qIn, qMiss, U, sampleBuf);
      return;
    }
    const bool spill8 = (dynRefill & TRACE_DYN_SPILL8) != 0u;
    // 8 entries (16 KB per block), 12 (24 KB: 5 blocks per CU still fit next to the 8 KB of WaveTri) or 16 (32 KB: 4 blocks -- one wave per SIMD fewer)
    const uint32_t entries = (sc.bvhDepth <= 8u || spill8) ? 8u : (sc.bvhDepth <= 12u ? 12u : 16u);
    const uint32_t stackBytes = entries * TRACE_BLOCK * (uint32_t)sizeof(uint2);
#define GI_LAUNCH_DYN(STACK, OVF) do { \
      if (ANYHIT && (refill & DYN_SLOT_ORDER)) \
        hipLaunchKernelGGL((k_trace_dyn<ANYHIT, COUNT, STACK, OVF, CUTOUT, ANYHIT>), dim3(blocks), dim3(TRACE_BLOCK), stackBytes, s, sc, st, qs, cnt, qIn, \
                           refill); \
      else hipLaunchKernelGGL((k_trace_dyn<ANYHIT, COUNT, STACK, OVF, CUTOUT, false>), dim3(blocks), dim3(TRACE_BLOCK), stackBytes, s, sc, st, qs, cnt, qIn, \
          refill); } while (0)
    if (sc.bvhDepth <= 8u) GI_LAUNCH_DYN(8, false);
    else if (spill8) GI_LAUNCH_DYN(8, true);
    else if (sc.bvhDepth <= 12u) GI_LAUNCH_DYN(12, false);
    else if (sc.bvhDepth <= 16u) GI_LAUNCH_DYN(16, false);
    else GI_LAUNCH_DYN(16, true);
#undef GI_LAUNCH_DYN
    if (!ANYHIT) launchRoute(s, routeBlocks, sc, st, qs, cnt, qIn, qMiss, U, sampleBuf);
    return;
  }
  const bool dome = !ANYHIT && (sc.domeTexture != 0u || sc.mediumStackSize != 0u); // misses need the slot: dome image lookup / scattering events
#define GI_LAUNCH_TRACE(STACK, OVF, LDS) do { \
    if (dome) \
      hipLaunchKernelGGL((k_trace<ANYHIT, COUNT, STACK, OVF, LDS, CUTOUT, !ANYHIT>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, sc, st, qs, cnt, qIn, qMiss, \
                         ln, lt, U, sampleBuf); \
    else hipLaunchKernelGGL((k_trace<ANYHIT, COUNT, STACK, OVF, LDS, CUTOUT, false>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, sc, st, qs, cnt, qIn, qMiss, \
        ln, lt, U, sampleBuf); } while (0)
  if (allLds && sc.bvhDepth <= 4u) GI_LAUNCH_TRACE(4, false, true);
  else if (allLds && sc.bvhDepth <= 8u) GI_LAUNCH_TRACE(8, false, true);
  else if (sc.bvhDepth <= 8u) GI_LAUNCH_TRACE(8, false, false);
  else if (sc.bvhDepth <= 16u) GI_LAUNCH_TRACE(16, false, false);
  else GI_LAUNCH_TRACE(16, true, false);
#undef GI_LAUNCH_TRACE
}
template <bool ANYHIT, bool COUNT>
static void launchTraceCutout(hipStream_t s, uint32_t blocks, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t qIn,
    uint32_t qMiss,
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
  if (!anyHit) { if (count) launchTraceCutout<false, true>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks, U, sampleBuf);
      else launchTraceCutout<false, false>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks, U, sampleBuf); }
  else { if (count) launchTraceCutout<true, true>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks, U, sampleBuf);
      else launchTraceCutout<true, false>(s, blocks, sc, st, qs, cnt, qIn, qMiss, dynRefill, routeBlocks, U, sampleBuf); }
}

} // namespace gi
