// gi_traversal.h -- software traversal of the 8-wide quantised BVH: per-lane traversal state and node test, the per-lane step
// (k_aov), the wave-cooperative triangle stage and wave_step (k_trace, k_trace_dyn).  Included by gi_kernels.hip only.
#pragma once

#include "gi_queues.h"
#include "gi_texture.h"

namespace gi {

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
// RayWalk is what a walk itself needs (k_trace_dyn keeps only this per lane: the nearest hit lives in the wave's LDS record); RayTrav adds the hit
// bookkeeping of the per-lane form.
struct RayWalk { V3 o, d; float idx, idy, idz, tMin, tBest; uint32_t octinv; uint2 G; uint32_t sp; };
struct RayTrav : RayWalk {
  uint32_t bestTri, bestOrig, bestMat; float bestU, bestV;
  bool found;
};

__device__ __forceinline__ void walk_init(RayWalk& R, V3 o, V3 d, float tMin, float tMax)
{
  R.o = o; R.d = d; R.tMin = tMin; R.tBest = tMax;
  // reciprocal direction for the slab tests only (guard against 0: boxes are padded, a huge finite value is safe)
  const float gx = (fabsf(d.x) < 1e-30f) ? (d.x < 0.0f ? -1e-30f : 1e-30f) : d.x;
  const float gy = (fabsf(d.y) < 1e-30f) ? (d.y < 0.0f ? -1e-30f : 1e-30f) : d.y;
  const float gz = (fabsf(d.z) < 1e-30f) ? (d.z < 0.0f ? -1e-30f : 1e-30f) : d.z;
  // v_rcp_f32 (1 ulp) instead of three IEEE divisions: the reciprocals only feed the box tests, whose far planes are widened by 1e-5
  R.idx = __builtin_amdgcn_rcpf(gx); R.idy = __builtin_amdgcn_rcpf(gy); R.idz = __builtin_amdgcn_rcpf(gz);
  R.octinv = ((d.x >= 0.0f ? 1u : 0u) | (d.y >= 0.0f ? 2u : 0u) | (d.z >= 0.0f ? 4u : 0u)) * 0x01010101u; // replicated into the 4 bytes (trav_node_test)
  R.G = make_uint2(0u, 0x80000000u); // virtual group holding only the root
  R.sp = 0u;
}
__device__ __forceinline__ void trav_init(RayTrav& R, V3 o, V3 d, float tMin, float tMax)
{
  walk_init(R, o, d, tMin, tMax);
  R.bestTri = 0xffffffffu; R.bestOrig = 0xffffffffu; R.bestMat = 0u; R.bestU = 0.0f; R.bestV = 0.0f; R.found = false;
}

// Node half of a traversal step, part 1: takes the next unvisited child of the current node group (pushing the rest) and returns its node index.  Internal
// children sit in bits 24-31 of G.y, flipped by the ray octant when the node was tested (ORDERED), so "highest bit first" is near-to-far along the ray.
// Caller guarantees R.G has node bits.
template <uint32_t STACK, bool OVERFLOW, bool ORDERED = true>
__device__ __forceinline__ uint32_t trav_node_pick(RayWalk& R, uint2 (*s_stack)[TRACE_BLOCK], uint2 (&overflow)[OVERFLOW ? OVF_STACK : 1])
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
  const uint32_t slot = ORDERED ? (bit - 24u) ^ (R.octinv & 7u) : bit - 24u;
  const uint32_t rel = (uint32_t)__popc((G.y & 0xffu) & ((1u << slot) - 1u));
  R.sp = sp;
  return G.x + rel;
}

// Part 2: tests the ray against the node's 8 quantised child boxes and returns the triangle group (base, mask) of its hit leaf children; R.G becomes the group
// of hit internal children.  The box test is a conservative filter (explicit fma, far planes and tBest widened by 1e-5 relative), it never decides a result.
// Written for the VALU: the two planes of an axis go through one packed fma (v_pk_fma_f32 issues at the rate of v_fma_f32, tools/valu_calib.hip), the per-child
// meta bytes (slot index, child bits, octant flip of internal children) are decoded four at a time with byte-parallel integer ops, and empty slots (meta 0)
// contribute no bits, so the hit mask is assembled without a branch.  ORDERED = false (shadow walks: the first hit ends them, near-to-far buys nothing) leaves
// the internal children in slot order and saves the flip.
typedef float gi_f2 __attribute__((ext_vector_type(2)));
template <bool SLACK = false, bool ORDERED = true>
__device__ __forceinline__ uint2 trav_node_test(RayWalk& R, const uint4& n0, const uint4& n1, const uint4& n2, const uint4& n3, const uint4& n4,
    float slack = 0.0f)
{
  const V3 o = R.o, d = R.d;
  constexpr float WIDEN = 1.00001f;
  // ray in the node's quantisation frame: t(q) = q * a + b per axis; .x = near plane, .y = far plane (widened)
  const float sx = u2f((n0.w & 0xffu) << 23), sy = u2f(((n0.w >> 8) & 0xffu) << 23), sz = u2f(((n0.w >> 16) & 0xffu) << 23);
  const float ax = sx * R.idx, ay = sy * R.idy, az = sz * R.idz;
  const float bx = (u2f(n0.x) - o.x) * R.idx, by = (u2f(n0.y) - o.y) * R.idy, bz = (u2f(n0.z) - o.z) * R.idz;
  const gi_f2 Ax = {ax, ax * WIDEN}, Ay = {ay, ay * WIDEN}, Az = {az, az * WIDEN};
  // SLACK (two-level layout): the ray was transformed into the instance's object space in fp32; every box is grown by `slack`
  // (an absolute object-space bound on that rounding) so the filter stays conservative for the WORLD-space exact test
  const float ex = SLACK ? slack * fabsf(R.idx) : 0.0f, ey = SLACK ? slack * fabsf(R.idy) : 0.0f, ez = SLACK ? slack * fabsf(R.idz) : 0.0f;
  // (without SLACK the `+ 0` must not be left to the compiler: x + 0 is not x for x = -0, so it keeps three adds per node test)
  const gi_f2 Bx = SLACK ? gi_f2{bx - ex, (bx + ex) * WIDEN} : gi_f2{bx, bx * WIDEN}, By = SLACK ? gi_f2{by - ey, (by + ey) * WIDEN} : gi_f2{by, by * WIDEN},
              Bz = SLACK ? gi_f2{bz - ez, (bz + ez) * WIDEN} : gi_f2{bz, bz * WIDEN};
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
    uint32_t idx4 = m4 & 0x1f1f1f1fu;
    if (ORDERED) {
      const uint32_t inner4 = ((m4 & (m4 << 1)) >> 4) & 0x01010101u;
      uint32_t hi4 = inner4 << 8;
      // (x << 8) - x == x * 0xff per byte: kept as shift + subtract (left alone the compiler folds it into v_mul_lo_u32, a quarter-rate instruction)
      asm volatile("" : "+v"(hi4));
      idx4 = (m4 ^ (oct4 & (hi4 - inner4))) & 0x1f1f1f1fu;
    }
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
      const uint32_t pos = (idx4 >> sh) & 0xffu;
      const uint32_t contrib = ((bits4 >> sh) & 0xffu) << pos;
      hitmask |= (tn <= tf) ? contrib : 0u;
    }
  }
  R.G = make_uint2(n1.x, (hitmask & 0xff000000u) | (n0.w >> 24));
  return make_uint2(n1.y, hitmask & 0x00ffffffu);
}

// fetch of a node's five 16-byte pieces from global memory: 32-bit byte offset on the scalar base (a scene's node array stays below 4 GiB: the flat layout ends
// at 2^26 triangles), two shift-adds instead of a 64-bit multiply-add
__device__ __forceinline__ void node_load(const SceneView& sc, uint32_t nodeIdx, uint4& n0, uint4& n1, uint4& n2, uint4& n3, uint4& n4)
{
  uint32_t off = nodeIdx << 4;
  asm volatile("" : "+v"(off)); // (keeps the compiler from folding the two shifts into a quarter-rate v_mul_lo_u32)
  off += nodeIdx << 6;
  const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sc.nodes) + off);
  n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3]; n4 = p[4];
}

// The per-lane composition (each lane fetches its own node: from LDS when staged there, else from global memory)
template <bool COUNT, uint32_t STACK, bool OVERFLOW, bool ALL_LDS, bool ORDERED = true>
__device__ __forceinline__ uint2 trav_node(RayWalk& R, const SceneView& sc, const uint4* s_nodes, uint32_t ldsNodes,
                                           uint2 (*s_stack)[TRACE_BLOCK], uint2 (&overflow)[OVERFLOW ? OVF_STACK : 1], TraceCounters& tc)
{
  const uint32_t nodeIdx = trav_node_pick<STACK, OVERFLOW, ORDERED>(R, s_stack, overflow);
  uint4 n0, n1, n2, n3, n4;
  if (ALL_LDS || nodeIdx < ldsNodes) { const uint4* p = s_nodes + nodeIdx * 5u; n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3]; n4 = p[4]; }
  else node_load(sc, nodeIdx, n0, n1, n2, n3, n4);
  if (COUNT) tc.nodes++;
  return trav_node_test<false, ORDERED>(R, n0, n1, n2, n3, n4);
}

// End of a step: when the current group has no unvisited internal child left, continue with the stack top.
// Returns true when the traversal is finished.
// `base`: the stack entries below it are no longer this walk's (k_trace_dyn: handed to helper lanes).
template <uint32_t STACK, bool OVERFLOW>
__device__ __forceinline__ bool trav_pop(RayWalk& R, uint2 (*s_stack)[TRACE_BLOCK], uint2 (&overflow)[OVERFLOW ? OVF_STACK : 1], uint32_t base = 0u)
{
  if (R.G.y & 0xff000000u) return false;
  if (R.sp == base) return true;
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
      const float opacity = cutout_opacity_at(sc, c.w, triIdx, u, v);
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
  // per ray lane: that hit as (triangle index, u bits, v bits, material word) -- k_trace_dyn: (u, v, triangle | class << 28) of
  // the finished result record (its t is the key's upper word), and in .w the number of helper lanes walking parts of this ray
  uint4 hit[64];
  uint32_t queue[128];         // ring of pending (ray lane, triangle) pairs (a 256-entry ring measured slower)
};
// WaveTri lives in LDS, but through a C++ reference the compiler only sees a generic pointer and emits FLAT loads / stores (vector-memory
// path, each volatile one followed by s_waitcnt vmcnt(0), i.e. a stall on every outstanding global load).  These accessors cast back to
// address space 3, so the exchanges are ds_read / ds_write with lgkmcnt waits.
#define GI_LDS __attribute__((address_space(3)))
typedef uint32_t gi_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wt_queue_put(WaveTri& W, uint32_t i, uint32_t v) { *(volatile GI_LDS uint32_t*)&((GI_LDS WaveTri*)&W)->queue[i] = v; }
__device__ __forceinline__ uint32_t wt_queue_get(WaveTri& W, uint32_t i) { return *(volatile GI_LDS uint32_t*)&((GI_LDS WaveTri*)&W)->queue[i]; }
__device__ __forceinline__ void wt_best_put(WaveTri& W, uint32_t i, unsigned long long v)
{ *(volatile GI_LDS unsigned long long*)&((GI_LDS WaveTri*)&W)->best[i] = v; }
__device__ __forceinline__ unsigned long long wt_best_get(WaveTri& W, uint32_t i)
{ return *(volatile GI_LDS unsigned long long*)&((GI_LDS WaveTri*)&W)->best[i]; }
// t bits of the nearest hit so far (or of tMax)
__device__ __forceinline__ uint32_t wt_best_t(WaveTri& W, uint32_t i) { return ((volatile GI_LDS uint32_t*)&((GI_LDS WaveTri*)&W)->best[i])[1]; }
// scene-order id + 1, 0 = no hit yet
__device__ __forceinline__ uint32_t wt_best_id(WaveTri& W, uint32_t i) { return ((volatile GI_LDS uint32_t*)&((GI_LDS WaveTri*)&W)->best[i])[0]; }
__device__ __forceinline__ void wt_best_min(WaveTri& W, uint32_t i, unsigned long long v)
{ __hip_atomic_fetch_min((GI_LDS unsigned long long*)&((GI_LDS WaveTri*)&W)->best[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wt_hit_put(WaveTri& W, uint32_t i, uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{ gi_u4 v = {x, y, z, w}; *(volatile GI_LDS gi_u4*)&((GI_LDS WaveTri*)&W)->hit[i] = v; }
typedef uint32_t gi_u3 __attribute__((ext_vector_type(3)));
// (leaves .w alone)
__device__ __forceinline__ void wt_hit_put3(WaveTri& W, uint32_t i, uint32_t x, uint32_t y, uint32_t z)
{ gi_u3 v = {x, y, z}; *(volatile GI_LDS gi_u3*)&((GI_LDS WaveTri*)&W)->hit[i] = v; }
__device__ __forceinline__ uint32_t wt_helpers_add(WaveTri& W, uint32_t i, uint32_t v)
{ return __hip_atomic_fetch_add(&((GI_LDS uint32_t*)&((GI_LDS WaveTri*)&W)->hit[i])[3], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint4 wt_hit_get(WaveTri& W, uint32_t i)
{ const gi_u4 v = *(volatile GI_LDS gi_u4*)&((GI_LDS WaveTri*)&W)->hit[i]; return make_uint4(v.x, v.y, v.z, v.w); }

// 64 (ray lane, triangle) pairs of the ring, one per lane.  RESULT_RECORD (k_trace_dyn): the winner leaves the ray's finished result record in WaveTri::hit.
template <bool COUNT, bool ALL_LDS, bool CUTOUT, bool RESULT_RECORD = false>
__device__ __forceinline__ void wave_tri_batch(WaveTri& W, uint32_t head, uint32_t cnt, const RayWalk& R, uint32_t rng, const SceneView& sc,
                                               const uint4* s_tris, uint32_t ldsTris, TraceCounters& tc)
{
  const uint32_t lane = __lane_id();
  const bool act = lane < cnt;
  const uint32_t e = act ? wt_queue_get(W, (head + lane) & 127u) : 0u;
  const uint32_t rl = e >> TRI_ID_BITS, triIdx = e & ((1u << TRI_ID_BITS) - 1u);
  // the owning lane's ray (executed by all lanes: wave-uniform control flow)
  const V3 o = v3(__shfl(R.o.x, (int)rl), __shfl(R.o.y, (int)rl), __shfl(R.o.z, (int)rl));
  const V3 d = v3(__shfl(R.d.x, (int)rl), __shfl(R.d.y, (int)rl), __shfl(R.d.z, (int)rl));
  const float tMin = __shfl(R.tMin, (int)rl);
  const uint32_t rrng = CUTOUT ? (uint32_t)__shfl((int)rng, (int)rl) : 0u;
  if (act) {
    uint4 a, b, c;
    // (the compiler loads c.x here and sinks the loads of c.y -- scene-order id -- and c.w -- material word -- into the accept branch; loading all 48 bytes
    // up front measured SLOWER in full launches, r03a: the vector-memory request path, not the dependent round trip, is what the batch waits for -- and no
    // faster in thin ones, r05i)
    if (ALL_LDS || triIdx < ldsTris) { const uint4* p = s_tris + triIdx * 3u; a = p[0]; b = p[1]; c = p[2]; }
    else { const uint4* p = reinterpret_cast<const uint4*>(sc.tris) + (size_t)triIdx * 4u; a = p[0]; b = p[1]; c = p[2]; }
    if (COUNT) tc.tris++;
    float t, u, v;
    bool accept = tri_test(o, d, tMin, a, b, c, t, u, v);
    if (CUTOUT && accept && (c.w & (1u << 28))) { // non-opaque material: stochastic cutout (ignoreIntersectionEXT, rp_main.ahit:57-60)
      const float opacity = cutout_opacity_at(sc, c.w, triIdx, u, v);
      accept = !(cutout_random(rrng, c.y) > opacity);
    }
    if (accept) {
      const unsigned long long key = ((unsigned long long)f2u(t) << 32) | (unsigned long long)(c.y + 1u);
      wt_best_min(W, rl, key);
      if (wt_best_get(W, rl) == key) {
        // the material class k_route sorts by rides in the top four bits
        if (RESULT_RECORD) wt_hit_put3(W, rl, f2u(u), f2u(v), triIdx | (((c.w >> 24) & 0xfu) << 28));
        else wt_hit_put(W, rl, triIdx, f2u(u), f2u(v), c.w);
      }
    }
  }
}

// wave prefix sum (inclusive) over one value per lane: six DPP adds
__device__ __forceinline__ uint32_t wave_scan_inclusive(uint32_t v)
{
  int scan = (int)v;
  scan += __builtin_amdgcn_update_dpp(0, scan, 0x111, 0xf, 0xf, false); // row_shr:1
  scan += __builtin_amdgcn_update_dpp(0, scan, 0x112, 0xf, 0xf, false); // row_shr:2
  scan += __builtin_amdgcn_update_dpp(0, scan, 0x114, 0xf, 0xf, false); // row_shr:4
  scan += __builtin_amdgcn_update_dpp(0, scan, 0x118, 0xf, 0xf, false); // row_shr:8
  scan += __builtin_amdgcn_update_dpp(0, scan, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
  scan += __builtin_amdgcn_update_dpp(0, scan, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
  return (uint32_t)scan;
}

// One step of all rays of a wave (k_trace: scenes staged in LDS): node phase per lane, then the cooperative triangle stage, then pop.  Wave-uniform
// control flow; lanes without a ray (alive == false) only help testing triangles.  Returns true when this lane's ray is finished.
template <bool ANYHIT, bool COUNT, uint32_t STACK, bool OVERFLOW, bool ALL_LDS, bool CUTOUT>
__device__ __forceinline__ bool wave_step(RayTrav& R, bool alive, WaveTri& W, const SceneView& sc, const uint4* s_nodes, uint32_t ldsNodes,
                                          const uint4* s_tris, uint32_t ldsTris, uint2 (*s_stack)[TRACE_BLOCK], uint2 (&overflow)[OVERFLOW ? OVF_STACK : 1],
                                          TraceCounters& tc, uint32_t rng)
{
  const uint32_t lane = __lane_id();
  uint2 Gt = make_uint2(0u, 0u);
  if (alive) Gt = trav_node<COUNT, STACK, OVERFLOW, ALL_LDS>(R, sc, s_nodes, ldsNodes, s_stack, overflow, tc);
  uint32_t head = 0u, tail = 0u; // wave-uniform
  { // positions from a wave prefix sum over the per-lane pair counts, then every lane writes its own pairs
    const uint32_t cntL = (uint32_t)__popc(Gt.y);
    const uint32_t scan = wave_scan_inclusive(cntL);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)scan, 63);
    if (total <= 128u) {
      uint32_t pos = scan - cntL;
      while (Gt.y) {
        const uint32_t k = (uint32_t)__ffs((int)Gt.y) - 1u;
        Gt.y &= Gt.y - 1u;
        wt_queue_put(W, pos, (lane << TRI_ID_BITS) | (Gt.x + k));
        pos++;
      }
      tail = total;
      while (tail - head >= 64u) { wave_tri_batch<COUNT, ALL_LDS, CUTOUT>(W, head, 64u, R, rng, sc, s_tris, ldsTris, tc); head += 64u; }
    }
  }
  for (;;) { // (more pairs than the ring holds) one ballot round per triangle, a batch whenever 64 pairs are pending
    const unsigned long long m = __ballot(Gt.y != 0u);
    if (!m) break;
    if (Gt.y) {
      const uint32_t k = (uint32_t)__ffs((int)Gt.y) - 1u;
      Gt.y &= Gt.y - 1u;
      wt_queue_put(W, (tail + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & 127u, (lane << TRI_ID_BITS) | (Gt.x + k));
    }
    tail += (uint32_t)__popcll(m);
    if (tail - head >= 64u) { wave_tri_batch<COUNT, ALL_LDS, CUTOUT>(W, head, 64u, R, rng, sc, s_tris, ldsTris, tc); head += 64u; }
  }
  if (tail != head) wave_tri_batch<COUNT, ALL_LDS, CUTOUT>(W, head, tail - head, R, rng, sc, s_tris, ldsTris, tc);
  bool done = false;
  if (alive) {
    const unsigned long long key = wt_best_get(W, lane);
    R.tBest = u2f((uint32_t)(key >> 32));
    R.found = (uint32_t)key != 0u;
    done = (ANYHIT && R.found) ? true : trav_pop<STACK, OVERFLOW>(R, s_stack, overflow);
  }
  return done;
}

// start of a ray in the cooperative scheme (after trav_init)
__device__ __forceinline__ void wave_ray_begin(WaveTri& W, float tMax) { wt_best_put(W, __lane_id(), (unsigned long long)f2u(tMax) << 32); }
// result of a finished ray
__device__ __forceinline__ void wave_ray_end(WaveTri& W, RayTrav& R)
{
  __atomic_signal_fence(__ATOMIC_SEQ_CST); // compiler only: the winning lane's store precedes this load in the wave's program order
  if (R.found) { const uint4 h = wt_hit_get(W, __lane_id()); R.bestTri = h.x; R.bestU = u2f(h.y); R.bestV = u2f(h.z); R.bestMat = h.w; }
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



// ------------------------------------------------------------------------------------------------
// Two-level traversal (k_trace_dyn2; SceneView::tlasNodes ...).  Instanced scenes flattened into one BVH are HBM-latency bound: 5 M
// flattened triangles + their subtrees are 0.4 GB of node / triangle records touched at random.  Here a lane walks the TLAS with
// the world-space ray; a TLAS leaf reference names an instance: the lane transforms the ray into that instance's object space
// (w2o, no renormalisation: t keeps its meaning) and walks the mesh's BLAS, which all instances share and which stays in the caches.
// The box tests are filters only.  A candidate triangle is REBUILT in world space from the object-space vertices with the instance
// transform -- the host's xformPoint arithmetic, so (v0, e1, e2) equal the flat layout's TriRec bit for bit -- and tested against the
// WORLD-space ray by the same tri_test: hits, tie-breaks and any-hit decisions are those of the flat layout.
// Stack (16 LDS entries; the host checks the bound): node groups as before, plus "instance groups" (x = TLAS_ITEM_TAG | first
// reference, y = 24-bit mask of the hit leaf references) that wait below the BLAS entries of the instance being walked.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t TLAS_ITEM_TAG = 0x80000000u, NO_INSTANCE = 0xffffffffu;
struct RayTrav2 : RayWalk { V3 wo, wd; uint32_t inst, spBase; float slack; };

__device__ __forceinline__ void trav2_set_ray(RayTrav2& R, V3 o, V3 d)
{
  R.o = o; R.d = d;
  const float gx = (fabsf(d.x) < 1e-30f) ? (d.x < 0.0f ? -1e-30f : 1e-30f) : d.x;
  const float gy = (fabsf(d.y) < 1e-30f) ? (d.y < 0.0f ? -1e-30f : 1e-30f) : d.y;
  const float gz = (fabsf(d.z) < 1e-30f) ? (d.z < 0.0f ? -1e-30f : 1e-30f) : d.z;
  R.idx = __builtin_amdgcn_rcpf(gx); R.idy = __builtin_amdgcn_rcpf(gy); R.idz = __builtin_amdgcn_rcpf(gz);
  R.octinv = ((d.x >= 0.0f ? 1u : 0u) | (d.y >= 0.0f ? 2u : 0u) | (d.z >= 0.0f ? 4u : 0u)) * 0x01010101u;
}
__device__ __forceinline__ void trav2_init(RayTrav2& R, V3 o, V3 d, float tMin, float tMax)
{
  walk_init(R, o, d, tMin, tMax);
  R.wo = o; R.wd = d; R.inst = NO_INSTANCE; R.spBase = 0u; R.slack = 0.0f;
}
__device__ __forceinline__ void trav2_enter(RayTrav2& R, const SceneView& sc, uint32_t inst)
{
  const float4* ip = reinterpret_cast<const float4*>(&sc.instTrav[inst]); // one 128-byte line: o2w rows | w2o[0..8], blasRoot, triBase, matFlags | slack
  const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2], r3 = ip[3], r4 = ip[4], r5 = ip[5], r6 = ip[6];
  const V3 rel = v3(R.wo.x - r0.w, R.wo.y - r1.w, R.wo.z - r2.w); // o2w translation = column 3
  const float w[9] = {r3.x, r3.y, r3.z, r3.w, r4.x, r4.y, r4.z, r4.w, r5.x};
  const V3 o = v3((w[0] * rel.x + w[1] * rel.y) + w[2] * rel.z, (w[3] * rel.x + w[4] * rel.y) + w[5] * rel.z, (w[6] * rel.x + w[7] * rel.y) + w[8] * rel.z);
  const V3 d = v3((w[0] * R.wd.x + w[1] * R.wd.y) + w[2] * R.wd.z, (w[3] * R.wd.x + w[4] * R.wd.y) + w[5] * R.wd.z,
      (w[6] * R.wd.x + w[7] * R.wd.y) + w[8] * R.wd.z);
  // rounding of the transform: a few ulps of the summed term magnitudes, for the origin and -- over the distance the ray covers inside
  // the mesh, <= |o'| + its extent -- for the direction; 4e-6 is > 30 ulps of that
  const float m0 = (fabsf(w[0] * rel.x) + fabsf(w[1] * rel.y)) + fabsf(w[2] * rel.z), m1 = (fabsf(w[3] * rel.x) + fabsf(w[4] * rel.y)) + fabsf(w[5] * rel.z),
              m2 = (fabsf(w[6] * rel.x) + fabsf(w[7] * rel.y)) + fabsf(w[8] * rel.z);
  R.slack = 4.0e-6f * ((m0 + m1) + (m2 + r6.x));
  trav2_set_ray(R, o, d);
  R.inst = inst; R.spBase = R.sp;
  R.G = make_uint2(f2u(r5.y), 0x80000000u); // virtual group holding only the BLAS root
}
__device__ __forceinline__ void trav2_leave(RayTrav2& R)
{
  trav2_set_ray(R, R.wo, R.wd);
  R.inst = NO_INSTANCE; R.slack = 0.0f;
}

// 64 (ray lane, mesh triangle) pairs: rebuild the world-space triangle of the owning lane's instance, then the flat layout's test
template <bool COUNT, bool CUTOUT>
__device__ __forceinline__ void wave_tri_batch2(WaveTri& W, uint32_t head, uint32_t cnt, const RayTrav2& R, uint32_t rng, const SceneView& sc,
    TraceCounters& tc)
{
  const uint32_t lane = __lane_id();
  const bool act = lane < cnt;
  const uint32_t e = act ? wt_queue_get(W, (head + lane) & 127u) : 0u;
  const uint32_t rl = e >> TRI_ID_BITS, bt = e & ((1u << TRI_ID_BITS) - 1u);
  const V3 o = v3(__shfl(R.wo.x, (int)rl), __shfl(R.wo.y, (int)rl), __shfl(R.wo.z, (int)rl));
  const V3 d = v3(__shfl(R.wd.x, (int)rl), __shfl(R.wd.y, (int)rl), __shfl(R.wd.z, (int)rl));
  const float tMin = __shfl(R.tMin, (int)rl);
  const uint32_t inst = (uint32_t)__shfl((int)R.inst, (int)rl);
  const uint32_t rrng = CUTOUT ? (uint32_t)__shfl((int)rng, (int)rl) : 0u;
  if (act) {
    const float4* tp = reinterpret_cast<const float4*>(&sc.blasTris[bt]);   // one line: p0.xyz p1.x | p1.yz p2.xy | p2.z prim
    const float4 ta = tp[0], tb = tp[1], tq = tp[2];
    const float4* ip = reinterpret_cast<const float4*>(&sc.instTrav[inst]); // one line (hot: the walk entered this instance through it)
    const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2], r5 = ip[5];
    const uint4 tv = make_uint4(f2u(r5.y), f2u(r5.z), f2u(r5.w), 0u); // (blasRoot, triBase, matFlags, -)
    const float4 pa = make_float4(ta.x, ta.y, ta.z, 0.0f), pb = make_float4(ta.w, tb.x, tb.y, 0.0f), pc = make_float4(tb.z, tb.w, tq.x, 0.0f);
    const uint4 t4 = make_uint4(0u, 0u, 0u, f2u(tq.y));
    // host xformPoint (gi_build.cpp): ((a0 p0 + a1 p1) + a2 p2) + a3
    const V3 p0 = v3(((r0.x * pa.x + r0.y * pa.y) + r0.z * pa.z) + r0.w, ((r1.x * pa.x + r1.y * pa.y) + r1.z * pa.z) + r1.w,
        ((r2.x * pa.x + r2.y * pa.y) + r2.z * pa.z) + r2.w);
    const V3 p1 = v3(((r0.x * pb.x + r0.y * pb.y) + r0.z * pb.z) + r0.w, ((r1.x * pb.x + r1.y * pb.y) + r1.z * pb.z) + r1.w,
        ((r2.x * pb.x + r2.y * pb.y) + r2.z * pb.z) + r2.w);
    const V3 p2 = v3(((r0.x * pc.x + r0.y * pc.y) + r0.z * pc.z) + r0.w, ((r1.x * pc.x + r1.y * pc.y) + r1.z * pc.z) + r1.w,
        ((r2.x * pc.x + r2.y * pc.y) + r2.z * pc.z) + r2.w);
    const V3 e1 = p1 - p0, e2 = p2 - p0;
    const uint32_t orig = tv.y + t4.w; // scene-order id in the flat numbering
    const uint4 a = make_uint4(f2u(p0.x), f2u(p0.y), f2u(p0.z), f2u(e1.x)), b = make_uint4(f2u(e1.y), f2u(e1.z), f2u(e2.x), f2u(e2.y)),
        c = make_uint4(f2u(e2.z), orig, inst, tv.z);
    if (COUNT) tc.tris++;
    float t, u, v;
    bool accept = tri_test(o, d, tMin, a, b, c, t, u, v);
    if (CUTOUT && accept && (c.w & (1u << 28))) {
      const float opacity = cutout_opacity_at(sc, c.w, sc.flatOfOrig[orig], u, v);
      accept = !(cutout_random(rrng, orig) > opacity);
    }
    if (accept) {
      const unsigned long long key = ((unsigned long long)f2u(t) << 32) | (unsigned long long)(orig + 1u);
      wt_best_min(W, rl, key);
      // the result record, with the SCENE-ORDER id: the kernel maps it to the flat index
      if (wt_best_get(W, rl) == key) wt_hit_put3(W, rl, f2u(u), f2u(v), orig | (((c.w >> 24) & 0xfu) << 28));
    }
  }
}

template <bool ANYHIT, bool COUNT, bool CUTOUT>
__device__ __forceinline__ bool wave_step2(RayTrav2& R, bool alive, WaveTri& W, const SceneView& sc, uint2 (*s_stack)[TRACE_BLOCK], TraceCounters& tc,
    uint32_t rng)
{
  const uint32_t lane = __lane_id(), tid = threadIdx.x;
  uint2 none[1];
  uint2 Gt = make_uint2(0u, 0u);
  if (alive && (R.G.y & 0xff000000u)) {
    const uint32_t nodeIdx = trav_node_pick<16u, false>(R, s_stack, none);
    const bool top = R.inst == NO_INSTANCE;
    const uint4* p = reinterpret_cast<const uint4*>(top ? sc.tlasNodes : sc.blasNodes) + (size_t)nodeIdx * 5u;
    const uint4 n0 = p[0], n1 = p[1], n2 = p[2], n3 = p[3], n4 = p[4];
    if (COUNT) tc.nodes++;
    Gt = trav_node_test<true>(R, n0, n1, n2, n3, n4, R.slack);
    if (top) { // hit leaf references name instances: they wait on the stack
      if (Gt.y) { s_stack[R.sp][tid] = make_uint2(Gt.x | TLAS_ITEM_TAG, Gt.y); R.sp++; }
      Gt.y = 0u;
    }
  }
  uint32_t head = 0u, tail = 0u; // wave-uniform
  for (;;) {
    const unsigned long long m = __ballot(Gt.y != 0u);
    if (!m) break;
    if (Gt.y) {
      const uint32_t k = (uint32_t)__ffs((int)Gt.y) - 1u;
      Gt.y &= Gt.y - 1u;
      wt_queue_put(W, (tail + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & 127u, (lane << TRI_ID_BITS) | (Gt.x + k));
    }
    tail += (uint32_t)__popcll(m);
    if (tail - head >= 64u) { wave_tri_batch2<COUNT, CUTOUT>(W, head, 64u, R, rng, sc, tc); head += 64u; }
  }
  if (tail != head) wave_tri_batch2<COUNT, CUTOUT>(W, head, tail - head, R, rng, sc, tc);
  bool done = false;
  if (alive) {
    const unsigned long long key = wt_best_get(W, lane);
    R.tBest = u2f((uint32_t)(key >> 32));
    if (ANYHIT && (uint32_t)key != 0u) done = true;
    else if (!(R.G.y & 0xff000000u)) {
      if (R.inst != NO_INSTANCE && R.sp == R.spBase) trav2_leave(R); // this instance's BLAS is exhausted
      if (R.sp == 0u) done = true;
      else {
        const uint32_t sp = --R.sp;
        uint2 E = s_stack[sp][tid];
        if (E.x & TLAS_ITEM_TAG) {
          const uint32_t k = (uint32_t)__ffs((int)E.y) - 1u;
          E.y &= E.y - 1u;
          if (E.y) { s_stack[sp][tid] = E; R.sp = sp + 1u; }
          trav2_enter(R, sc, sc.tlasItems[(E.x & ~TLAS_ITEM_TAG) + k]);
        } else R.G = E;
      }
    }
  }
  return done;
}

} // namespace gi
