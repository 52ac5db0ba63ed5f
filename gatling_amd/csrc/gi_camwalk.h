// gi_camwalk.h -- the shared walk of a wave's camera rays (k_raygen<CAM>).  Included by gi_kernels.hip only.
//
// In the pixel-major work order the 64 camera rays a wave of k_raygen generates are samples of ONE pixel (gi_queues.h work_item; Gaussian pixel filter, sigma 0.375 px):
// on C3 a per-lane walk of such rays visits 10.2 nodes per ray and the UNION of the 64 walks is 13.1 nodes, 72 % of them visited by 48 lanes or more
// (tools/bvh_quality.cpp `packet`, profiles/r06d_camera_packet_coherence.txt) -- the walks are the same walk.  k_trace_dyn cannot use that: it runs one walk per lane,
// each with its own node fetch, its own decode of the node's meta bytes, its own stack in LDS and its share of the (lane, triangle) pair ring: 331 VALU instructions
// per wave and node step, whatever the lanes have in common.  Here the WAVE walks: one node per step, fetched once through the scalar cache; which child is internal,
// which is a leaf, where its triangles are and which of the two quantised planes is the near one are scalar facts (the rays of a walk share their direction octant:
// a lane whose octant differs from the wave's takes the ordinary route through the TRACE queue); a lane's part is the slab test itself -- six conversions, three packed
// fma, min / max, one compare per child -- whose result is the child's LANE MASK, which is also all the wave-level stack needs.  Leaf triangles are tested on the
// spot by the lanes of the leaf's mask (a node's hit leaf triangles are fetched together, one per lane, and read back with readlane), nearest hit and tie-break in registers.
//
// Results are those of any other walk (DESIGN.md "Traversal contract"): the slab test is the conservative filter of trav_node_test (same arithmetic: explicit fma,
// far plane and tBest widened by 1e-5), a lane takes part in a child exactly when its own test passes, triangles go through tri_test and the contract's accept rule
// (tMin < t < tBest, ties to the lower scene-order id), cutouts through the order-independent any-hit draw.  Visiting order (near-to-far by the wave's octant) and
// culling distance (a lane's own tBest at the time of the test) only decide what is visited, never what is found.
#pragma once

#include "gi_traversal.h"

namespace gi {

constexpr uint32_t CAM_STACK = 18; // levels of the wave's stack: one group per tree level (the host enables the walk for trees of bvhDepth + 1 <= CAM_STACK levels)
// one node's hit internal children, waiting: first child node, the node's internal mask, the children still to visit, and per child slot the lanes whose ray entered it
struct CamGroup { uint32_t childBase, imask, pending, pad; unsigned long long lanes[8]; };
struct CamStack { CamGroup g[CAM_STACK]; }; // per wave, in LDS (1 440 bytes)

typedef const __attribute__((address_space(4))) gi_u4 gi_const_u4; // scene arrays are read-only while a kernel runs: uniform addresses in the constant address space become s_load
__device__ __forceinline__ uint4 load_uniform_u4(const void* base, uint32_t byteOffset)
{
  const gi_u4 v = *reinterpret_cast<gi_const_u4*>(reinterpret_cast<uintptr_t>(base) + byteOffset);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{
  return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); } // lane: wave-uniform

// All 64 lanes call this (wave-uniform control flow).  `active`: the lane carries a ray; every active lane's direction octant is `oct` (bit 0 = d.x >= 0, ...).
// Out: t, u, v and the result word (triangle index | shade class << 28, or MISS) -- the record k_trace_dyn leaves in a ray's place.
template <bool COUNT, bool CUTOUT>
__device__ __forceinline__ void cam_walk(const SceneView& sc, CamStack& S, bool active, uint32_t oct, V3 o, V3 d, float tMin, float tMax, uint32_t rng,
                                         float& outT, float& outU, float& outV, uint32_t& outWord, TraceCounters& tc)
{
  const uint32_t lane = __lane_id();
  GI_LDS CamStack* L = (GI_LDS CamStack*)&S;
  float tBest = tMax, bu = 0.0f, bv = 0.0f; uint32_t bestOrig = 0xffffffffu, bestWord = MISS;
  // the ray in slab form, as walk_init: reciprocal direction (guarded against 0; v_rcp_f32: the reciprocals only feed the filter)
  const float gx = (fabsf(d.x) < 1e-30f) ? (d.x < 0.0f ? -1e-30f : 1e-30f) : d.x;
  const float gy = (fabsf(d.y) < 1e-30f) ? (d.y < 0.0f ? -1e-30f : 1e-30f) : d.y;
  const float gz = (fabsf(d.z) < 1e-30f) ? (d.z < 0.0f ? -1e-30f : 1e-30f) : d.z;
  const float idx = __builtin_amdgcn_rcpf(gx), idy = __builtin_amdgcn_rcpf(gy), idz = __builtin_amdgcn_rcpf(gz);
  constexpr float WIDEN = 1.00001f;
  const bool negX = (oct & 1u) == 0u, negY = (oct & 2u) == 0u, negZ = (oct & 4u) == 0u; // wave-uniform: the near plane of an axis is the high one when the rays run down it
  unsigned long long lanes = __ballot(active);
  uint32_t node = 0u, sp = 0u;
  if (lanes == 0ull) { outT = tMax; outU = 0.0f; outV = 0.0f; outWord = MISS; return; }
  for (;;) {
    const bool in = ((lanes >> lane) & 1ull) != 0ull;
    if (COUNT && in) tc.nodes++;
    uint32_t off = node << 4; off += node << 6; // 80 bytes per node
    const uint4 n0 = load_uniform_u4(sc.nodes, off), n1 = load_uniform_u4(sc.nodes, off + 16u), n2 = load_uniform_u4(sc.nodes, off + 32u), n3 = load_uniform_u4(sc.nodes, off + 48u),
                n4 = load_uniform_u4(sc.nodes, off + 64u);
    // ray in the node's quantisation frame (trav_node_test): t(q) = q * a + b per axis, .x near plane, .y far plane (widened)
    const float sx = u2f((n0.w & 0xffu) << 23), sy = u2f(((n0.w >> 8) & 0xffu) << 23), sz = u2f(((n0.w >> 16) & 0xffu) << 23);
    const float ax = sx * idx, ay = sy * idy, az = sz * idz;
    const float bx = (u2f(n0.x) - o.x) * idx, by = (u2f(n0.y) - o.y) * idy, bz = (u2f(n0.z) - o.z) * idz;
    const gi_f2 Ax = {ax, ax * WIDEN}, Ay = {ay, ay * WIDEN}, Az = {az, az * WIDEN};
    const gi_f2 Bx = {bx, bx * WIDEN}, By = {by, by * WIDEN}, Bz = {bz, bz * WIDEN};
    const uint32_t imask = n0.w >> 24;
    uint32_t pending = 0u;
    unsigned long long lanesOf[8];
    // The hit leaf slots' triangles are fetched TOGETHER: lane j takes the j-th of them (<= 24 per node) and remembers which lanes entered its leaf; the tests below
    // then read a triangle out of its lane with readlane.  (The first version fetched them one after the other through the scalar cache: every triangle a dependent
    // round trip, a dozen per walk on top of the node fetches -- 1.8 x the per-lane walk's time on C3, profiles/r06e_camwalk_first_version.log.)
    uint32_t nT = 0u; // triangles gathered (wave-uniform)
    uint32_t myTri = 0u; unsigned long long myMask = 0ull;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      lanesOf[k] = 0ull;
      const uint32_t meta = ((k < 4 ? n1.z : n1.w) >> (8u * (uint32_t)(k & 3))) & 0xffu; // scalar
      if (meta == 0u) continue;                                                            // (empty slot: a scalar branch)
      const uint32_t sh = 8u * (uint32_t)(k & 3);
      const uint32_t qlox = ((k < 4 ? n2.x : n2.y) >> sh) & 0xffu, qloy = ((k < 4 ? n2.z : n2.w) >> sh) & 0xffu, qloz = ((k < 4 ? n3.x : n3.y) >> sh) & 0xffu;
      const uint32_t qhix = ((k < 4 ? n3.z : n3.w) >> sh) & 0xffu, qhiy = ((k < 4 ? n4.x : n4.y) >> sh) & 0xffu, qhiz = ((k < 4 ? n4.z : n4.w) >> sh) & 0xffu;
      const gi_f2 qx = {(float)(negX ? qhix : qlox), (float)(negX ? qlox : qhix)};
      const gi_f2 qy = {(float)(negY ? qhiy : qloy), (float)(negY ? qloy : qhiy)};
      const gi_f2 qz = {(float)(negZ ? qhiz : qloz), (float)(negZ ? qloz : qhiz)};
      const gi_f2 tx = __builtin_elementwise_fma(qx, Ax, Bx), ty = __builtin_elementwise_fma(qy, Ay, By), tz = __builtin_elementwise_fma(qz, Az, Bz);
      const float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, tMin));
      const float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, tBest * WIDEN));
      const unsigned long long m = __ballot(in && (tn <= tf));
      if (m == 0ull) continue;
      if ((imask >> k) & 1u) { lanesOf[k] = m; pending |= 1u << k; continue; }
      // a leaf slot: its 1 .. 3 triangles join the node's batch
      const uint32_t unary = meta >> 5, cntT = unary == 1u ? 1u : (unary == 3u ? 2u : 3u), triFirst = n1.y + (meta & 31u);
      if (lane >= nT && lane < nT + cntT) { myTri = triFirst + (lane - nT); myMask = m; }
      nT += cntT;
    }
    if (nT != 0u) {
      uint4 ta = make_uint4(0u, 0u, 0u, 0u), tb = ta, tc4 = ta;
      if (lane < nT) { const uint4* p = reinterpret_cast<const uint4*>(sc.tris) + (size_t)myTri * 4u; ta = p[0]; tb = p[1]; tc4 = p[2]; }
      for (uint32_t j = 0; j < nT; j++) { // wave-uniform loop; triangle j lives in lane j
        const uint4 a = make_uint4(rdl(ta.x, j), rdl(ta.y, j), rdl(ta.z, j), rdl(ta.w, j)), b = make_uint4(rdl(tb.x, j), rdl(tb.y, j), rdl(tb.z, j), rdl(tb.w, j)),
                    c = make_uint4(rdl(tc4.x, j), rdl(tc4.y, j), rdl(tc4.z, j), rdl(tc4.w, j));
        const uint32_t triIdx = rdl(myTri, j);
        const unsigned long long m = ((unsigned long long)rdl((uint32_t)(myMask >> 32), j) << 32) | rdl((uint32_t)myMask, j);
        if ((m >> lane) & 1ull) {
          if (COUNT) tc.tris++;
          float t, u, v;
          const bool inside = tri_test(o, d, tMin, a, b, c, t, u, v);
          const bool better = (t < tBest) | ((t == tBest) & (bestOrig != 0xffffffffu) & (c.y < bestOrig));
          bool accept = inside & better;
          if (CUTOUT && accept && (c.w & (1u << 28))) { // non-opaque material: stochastic cutout (ignoreIntersectionEXT, rp_main.ahit:57-60)
            const float opacity = cutout_opacity_at(sc, c.w, triIdx, u, v);
            accept = !(cutout_random(rng, c.y) > opacity);
          }
          if (accept) { tBest = t; bu = u; bv = v; bestOrig = c.y; bestWord = triIdx | (((c.w >> 24) & 0xfu) << 28); }
        }
      }
    }
    // the node's hit internal children wait on the wave's stack; the walk goes on with the nearest child of the top group
    if (pending != 0u) {
      if (lane == 0u) { L->g[sp].childBase = n1.x; L->g[sp].imask = imask; L->g[sp].pending = pending; }
#pragma unroll
      for (int k = 0; k < 8; k++) if (lane == (uint32_t)k + 1u) L->g[sp].lanes[k] = lanesOf[k];
      sp++;
    }
    if (sp == 0u) break;
    __builtin_amdgcn_wave_barrier();
    const uint32_t top = sp - 1u;
    const uint32_t gPending = uni32(*(volatile GI_LDS uint32_t*)&L->g[top].pending), gMask = uni32(*(volatile GI_LDS uint32_t*)&L->g[top].imask),
                   gBase = uni32(*(volatile GI_LDS uint32_t*)&L->g[top].childBase);
    // near-to-far along the wave's octant: the slot with the highest (slot ^ oct) first -- trav_node_pick's rule
    uint32_t flipped = 0u;
#pragma unroll
    for (uint32_t s = 0; s < 8u; s++) flipped |= ((gPending >> s) & 1u) << (s ^ oct);
    const uint32_t slot = (31u - (uint32_t)__clz((int)flipped)) ^ oct;
    lanes = uni64(*(volatile GI_LDS unsigned long long*)&L->g[top].lanes[slot]);
    node = gBase + (uint32_t)__popc(gMask & ((1u << slot) - 1u));
    const uint32_t left = gPending & ~(1u << slot);
    if (left == 0u) sp = top;
    else if (lane == 0u) *(volatile GI_LDS uint32_t*)&L->g[top].pending = left;
    __builtin_amdgcn_wave_barrier();
  }
  outT = tBest; outU = bu; outV = bv; outWord = bestWord;
}

} // namespace gi
