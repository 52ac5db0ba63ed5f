// gi_kernels.hip -- the wavefront path tracer's stage kernels for gfx950 (CDNA4, wave64).
//
// Replaces the Vulkan ray-tracing megakernel of the reference:
//   rp_main.rgen  (/root/reference/src/gi/shaders/rp_main.rgen:185-521)  -> k_raygen + the host bounce loop
//   traceRayEXT   (rp_main.rgen:381-393, 412-424; HW BVH traversal)        -> k_trace<closest>, k_trace<any>
//   rp_main.chit  (rp_main.chit:132-493) + rp_main.miss (:55-86)           -> k_shade
//   rp_main_shadow.miss + the NEE add (rp_main.rgen:426-429)               -> k_trace<any> epilogue
// One slot per pixel of the tile walks its samples in order, so the per-pixel float accumulation order of
// rp_main.rgen:498 is preserved exactly while different slots sit in different stages/queues.
//
// Built with -ffp-contract=off (arithmetic contract, see gi_device_math.h).  Box tests inside the traversal use
// explicit fmaf: they are conservative filters and never influence results.

#include <hip/hip_runtime.h>

#include "gi_device_math.h"
#include "gi_kernels.h"
#include "gi_types.h"

namespace gi {

// ------------------------------------------------------------------------------------------------
// wave64 stream compaction: ballot + prefix popcount, one atomic per wave and queue
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_append(bool pred, uint32_t value, uint32_t* __restrict__ queue, uint32_t* counter)
{
  unsigned long long m = __ballot(pred);
  if (m == 0ull) return;
  uint32_t lane = __lane_id();
  int leader = __ffsll((long long)m) - 1;
  uint32_t base = 0;
  if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = __shfl(base, leader);
  if (pred) queue[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}

__device__ __forceinline__ F4 ld4(const F4* p) { float4 v = *reinterpret_cast<const float4*>(p); return F4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void st4(F4* p, float x, float y, float z, float w) { *reinterpret_cast<float4*>(p) = make_float4(x, y, z, w); }

// ------------------------------------------------------------------------------------------------
// k_init: every slot starts in the regen queue with "no sample in flight"
// ------------------------------------------------------------------------------------------------
__global__ void k_init(PathState st, uint32_t* qRegen, Counters* cnt, uint32_t n)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    cnt->count[Q_TRACE_A] = 0; cnt->count[Q_TRACE_B] = 0; cnt->count[Q_REGEN] = n; cnt->count[Q_SHADOW] = 0;
    cnt->segments = 0; cnt->shadowRays = 0; cnt->nodesVisited = 0; cnt->trisTested = 0; cnt->shadowNodesVisited = 0; cnt->shadowTrisTested = 0;
  }
  for (; i < n; i += gridDim.x * blockDim.x) {
    st4(&st.acc[i], 0.0f, 0.0f, 0.0f, u2f(0xffffffffu));
    qRegen[i] = i;
  }
}

__global__ void k_reset(Counters* cnt, uint32_t a, uint32_t b, uint32_t c)
{
  if (threadIdx.x == 0) { cnt->count[a] = 0; cnt->count[b] = 0; cnt->count[c] = 0; }
}

// ------------------------------------------------------------------------------------------------
// k_raygen: persistent-thread ray generation + per-sample finish (rp_main.rgen:213-283, 483-515)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_raygen(FrameUniforms U, PathState st, const uint32_t* __restrict__ qRegen,
                                                uint32_t* __restrict__ qTrace, Counters* cnt, uint32_t traceIdx, F4* __restrict__ colorOut)
{
  const uint32_t n = cnt->count[Q_REGEN];
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t slot = qRegen[i];
    F4 acc = ld4(&st.acc[slot]);
    uint32_t s = f2u(acc.w);
    V3 pixelColor = v3(acc.x, acc.y, acc.z);
    if (s != 0xffffffffu) { // finish the sample that just terminated (:489-498)
      F4 r = ld4(&st.rad[slot]);
      V3 rad = v3(r.x, r.y, r.z);
      float mv = fmax2(rad.x, fmax2(rad.y, rad.z));
      if (mv > U.maxSampleValue) rad = rad * (U.maxSampleValue / mv);
      V3 sc = v3(fmax2(0.0f, rad.x), fmax2(0.0f, rad.y), fmax2(0.0f, rad.z));
      pixelColor = pixelColor + sc * U.invSpp;
    }
    s = s + 1u; // 0xffffffff + 1 == 0
    const uint32_t pixelIndex = U.rowBegin * U.imageWidth + slot; // :195 (global index: RNG is tile independent)
    bool more = s < U.spp;
    if (more) {
      const uint32_t px = pixelIndex % U.imageWidth, py = pixelIndex / U.imageWidth;
      uint32_t rng = gi_hash_init(pixelIndex * ((U.sampleOffset + s) + 1u)); // :223, common.glsl:121-124
      float r0 = gi_next1f(rng), r1 = gi_next1f(rng);                       // :224 (always drawn)
      float sox = 0.5f, soy = 0.5f;
      if (U.flags & FLAG_JITTER) {
        if (U.flags & FLAG_FIS) { float gx, gy; gi_fis_gauss(r0, r1, gx, gy); sox = 0.5f + gx; soy = 0.5f + gy; }
        else { sox = r0; soy = r1; }
      }
      V3 camRight = v3(U.camRight), camUp = v3(U.camUp), camPos = v3(U.camPos);
      V3 P = (v3(U.L) + (camRight * ((float)px + sox)) * U.WX) + (camUp * ((float)py + soy)) * U.HY; // :239-242
      V3 origin = camPos;
      V3 dir = normalize(P - origin);
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
      float tMin = 0.0f, tMax = GI_FLT_MAX;
      if (U.flags & FLAG_CLIP) { // :287-288, 308-314 (bounce 0 only)
        float cosCone = fmax2(1e-5f, dot(dir, v3(U.camFwd)));
        tMin = U.clipNear / cosCone; tMax = U.clipFar / cosCone;
      }
      st4(&st.rayO[slot], origin.x, origin.y, origin.z, tMin);
      st4(&st.rayD[slot], dir.x, dir.y, dir.z, tMax);
      st4(&st.thr[slot], 1.0f, 1.0f, 1.0f, u2f(0u)); // :274-276
      st4(&st.rad[slot], 0.0f, 0.0f, 0.0f, u2f(rng));
      st4(&st.acc[slot], pixelColor.x, pixelColor.y, pixelColor.z, u2f(s));
    } else { // :506-515
      V3 prev = pixelColor;
      if ((U.flags & FLAG_PROGRESSIVE) && U.sampleOffset > 0u) { F4 p = ld4(&colorOut[pixelIndex]); prev = v3(p.x, p.y, p.z); }
      V3 c = (prev * U.sampleOffsetF + pixelColor * U.sppF) * U.invTotalSampleCount;
      st4(&colorOut[pixelIndex], c.x, c.y, c.z, 1.0f);
    }
    wave_append(more, slot, qTrace, &cnt->count[traceIdx]);
  }
}

// ------------------------------------------------------------------------------------------------
// k_trace: software traversal of the 8-wide quantised BVH, one ray per lane.
//   * persistent blocks stage the top of the tree (and, for small scenes, all triangles) into LDS once
//   * per-lane traversal stack: 8 entries in LDS + scratch overflow
//   * octant-ordered child visits (Ylitie et al. 2017), two-sided Moeller-Trumbore on 48-byte records
// Traversal contract: accept tMin < t < tBest; ties go to the lower scene-order triangle id.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t LDS_NODES = 384;  // 30 KiB
constexpr uint32_t LDS_TRIS = 128;   // 6 KiB
constexpr uint32_t LDS_STACK = 8;    // x 8 B x 256 lanes = 16 KiB
constexpr uint32_t OVF_STACK = 40;
constexpr uint32_t TRACE_BLOCK = 256;

struct TraceCounters { uint32_t nodes, tris; };

template <bool ANYHIT, bool COUNT>
__device__ __forceinline__ bool traverse(const SceneView& sc, const uint4* s_nodes, uint32_t ldsNodes, const uint4* s_tris, uint32_t ldsTris,
                                         uint2 (*s_stack)[TRACE_BLOCK], V3 o, V3 d, float tMin, float tMax,
                                         float& outT, float& outU, float& outV, uint32_t& outTri, TraceCounters& tc)
{
  const uint32_t tid = threadIdx.x;
  // reciprocal direction for the slab tests only (guard against 0: boxes are padded, a huge finite value is safe)
  const float gx = (fabsf(d.x) < 1e-30f) ? (d.x < 0.0f ? -1e-30f : 1e-30f) : d.x;
  const float gy = (fabsf(d.y) < 1e-30f) ? (d.y < 0.0f ? -1e-30f : 1e-30f) : d.y;
  const float gz = (fabsf(d.z) < 1e-30f) ? (d.z < 0.0f ? -1e-30f : 1e-30f) : d.z;
  const float idx = 1.0f / gx, idy = 1.0f / gy, idz = 1.0f / gz;
  const uint32_t octinv = (d.x >= 0.0f ? 1u : 0u) | (d.y >= 0.0f ? 2u : 0u) | (d.z >= 0.0f ? 4u : 0u);

  float tBest = tMax; uint32_t bestTri = 0xffffffffu, bestOrig = 0xffffffffu; float bestU = 0.0f, bestV = 0.0f;
  uint2 overflow[OVF_STACK];
  uint32_t sp = 0;
  uint2 G = make_uint2(0u, 0x80000000u); // virtual group holding only the root
  bool found = false;

  for (;;) {
    uint2 Gt;
    if (G.y & 0xff000000u) {
      const uint32_t bit = 31u - (uint32_t)__clz((int)(G.y & 0xff000000u));
      G.y &= ~(1u << bit);
      if (G.y & 0xff000000u) { if (sp < LDS_STACK) s_stack[sp][tid] = G; else overflow[sp - LDS_STACK] = G; sp++; }
      const uint32_t slot = (bit - 24u) ^ octinv;
      const uint32_t rel = (uint32_t)__popc((G.y & 0xffu) & ((1u << slot) - 1u));
      const uint32_t nodeIdx = G.x + rel;
      uint4 n0, n1, n2, n3, n4;
      if (nodeIdx < ldsNodes) { const uint4* p = s_nodes + nodeIdx * 5u; n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3]; n4 = p[4]; }
      else { const uint4* p = reinterpret_cast<const uint4*>(sc.nodes) + (size_t)nodeIdx * 5u; n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3]; n4 = p[4]; }
      if (COUNT) tc.nodes++;
      // ray in the node's quantisation frame
      const float sx = u2f((n0.w & 0xffu) << 23), sy = u2f(((n0.w >> 8) & 0xffu) << 23), sz = u2f(((n0.w >> 16) & 0xffu) << 23);
      const float ax = sx * idx, ay = sy * idy, az = sz * idz;
      const float bx = (u2f(n0.x) - o.x) * idx, by = (u2f(n0.y) - o.y) * idy, bz = (u2f(n0.z) - o.z) * idz;
      // near/far plane bytes per axis, chosen by direction sign
      const bool nxn = d.x < 0.0f, nyn = d.y < 0.0f, nzn = d.z < 0.0f;
      const uint32_t qlox[2] = {n2.x, n2.y}, qloy[2] = {n2.z, n2.w}, qloz[2] = {n3.x, n3.y};
      const uint32_t qhix[2] = {n3.z, n3.w}, qhiy[2] = {n4.x, n4.y}, qhiz[2] = {n4.z, n4.w};
      const uint32_t metaw[2] = {n1.z, n1.w};
      uint32_t hitmask = 0u;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t nearx = nxn ? qhix[h] : qlox[h], farx = nxn ? qlox[h] : qhix[h];
        const uint32_t neary = nyn ? qhiy[h] : qloy[h], fary = nyn ? qloy[h] : qhiy[h];
        const uint32_t nearz = nzn ? qhiz[h] : qloz[h], farz = nzn ? qloz[h] : qhiz[h];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t sh = 8u * (uint32_t)k;
          const float t0x = fmaf((float)((nearx >> sh) & 0xffu), ax, bx), t1x = fmaf((float)((farx >> sh) & 0xffu), ax, bx);
          const float t0y = fmaf((float)((neary >> sh) & 0xffu), ay, by), t1y = fmaf((float)((fary >> sh) & 0xffu), ay, by);
          const float t0z = fmaf((float)((nearz >> sh) & 0xffu), az, bz), t1z = fmaf((float)((farz >> sh) & 0xffu), az, bz);
          const float tn = fmaxf(fmaxf(t0x, t0y), fmaxf(t0z, tMin));
          const float tf = fminf(fminf(t1x, t1y), fminf(t1z, tBest));
          const uint32_t meta = (metaw[h] >> sh) & 0xffu;
          if (tn <= tf * 1.00001f + 1e-30f && meta != 0u) {
            const uint32_t inner = ((meta & 0x18u) == 0x18u) ? octinv : 0u;
            hitmask |= (meta >> 5) << ((meta & 31u) ^ inner);
          }
        }
      }
      G = make_uint2(n1.x, (hitmask & 0xff000000u) | (n0.w >> 24));
      Gt = make_uint2(n1.y, hitmask & 0x00ffffffu);
    } else {
      Gt = G; G = make_uint2(0u, 0u);
    }
    // triangles of this node
    while (Gt.y) {
      const uint32_t k = (uint32_t)__ffs((int)Gt.y) - 1u;
      Gt.y &= Gt.y - 1u;
      const uint32_t triIdx = Gt.x + k;
      uint4 a, b, c;
      if (triIdx < ldsTris) { const uint4* p = s_tris + triIdx * 3u; a = p[0]; b = p[1]; c = p[2]; }
      else { const uint4* p = reinterpret_cast<const uint4*>(sc.tris) + (size_t)triIdx * 3u; a = p[0]; b = p[1]; c = p[2]; }
      if (COUNT) tc.tris++;
      const V3 v0 = v3(u2f(a.x), u2f(a.y), u2f(a.z)), e1 = v3(u2f(a.w), u2f(b.x), u2f(b.y)), e2 = v3(u2f(b.z), u2f(b.w), u2f(c.x));
      const uint32_t orig = c.w;
      // two-sided Moeller-Trumbore, operation order == oracle tri_test
      const V3 pv = cross(d, e2);
      const float det = dot(e1, pv);
      if (det == 0.0f) continue;
      const float inv = 1.0f / det;
      const V3 tv = o - v0;
      const float u = dot(tv, pv) * inv;
      if (!(u >= 0.0f)) continue;
      const V3 qv = cross(tv, e1);
      const float v = dot(d, qv) * inv;
      if (!(v >= 0.0f) || !(u + v <= 1.0f)) continue;
      const float t = dot(e2, qv) * inv;
      if (!(t > tMin)) continue;
      if (t < tBest || (t == tBest && bestOrig != 0xffffffffu && orig < bestOrig)) {
        tBest = t; bestU = u; bestV = v; bestTri = triIdx; bestOrig = orig; found = true;
        if (ANYHIT) { G.y = 0u; sp = 0u; break; }
      }
    }
    if (!(G.y & 0xff000000u)) {
      if (sp == 0u) break;
      sp--;
      G = (sp < LDS_STACK) ? s_stack[sp][tid] : overflow[sp - LDS_STACK];
    }
  }
  outT = tBest; outU = bestU; outV = bestV; outTri = bestTri;
  return found;
}

template <bool ANYHIT, bool COUNT>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace(SceneView sc, PathState st, const uint32_t* __restrict__ queue, Counters* cnt, uint32_t queueIdx)
{
  __shared__ uint4 s_nodes[LDS_NODES * 5];
  __shared__ uint4 s_tris[LDS_TRIS * 3];
  __shared__ uint2 s_stack[LDS_STACK][TRACE_BLOCK];
  const uint32_t n = cnt->count[queueIdx];
  if (blockIdx.x * TRACE_BLOCK >= n) return; // whole block idle (uniform)
  const uint32_t ldsNodes = sc.nodeCount < LDS_NODES ? sc.nodeCount : LDS_NODES;
  const uint32_t ldsTris = sc.triCount <= LDS_TRIS ? sc.triCount : 0u;
  for (uint32_t i = threadIdx.x; i < ldsNodes * 5u; i += TRACE_BLOCK) s_nodes[i] = reinterpret_cast<const uint4*>(sc.nodes)[i];
  for (uint32_t i = threadIdx.x; i < ldsTris * 3u; i += TRACE_BLOCK) s_tris[i] = reinterpret_cast<const uint4*>(sc.tris)[i];
  __syncthreads();

  TraceCounters tc{0u, 0u};
  uint32_t rays = 0;
  const uint32_t stride = gridDim.x * TRACE_BLOCK;
  for (uint32_t i = blockIdx.x * TRACE_BLOCK + threadIdx.x; i < n; i += stride) {
    const uint32_t slot = queue[i];
    rays++;
    if (!ANYHIT) {
      const F4 ro = ld4(&st.rayO[slot]), rd = ld4(&st.rayD[slot]);
      float t, u, v; uint32_t tri;
      traverse<false, COUNT>(sc, s_nodes, ldsNodes, s_tris, ldsTris, s_stack, v3(ro.x, ro.y, ro.z), v3(rd.x, rd.y, rd.z), ro.w, rd.w, t, u, v, tri, tc);
      st4(&st.hit[slot], t, u, v, u2f(tri));
    } else {
      // shadow ray (rp_main.rgen:397-429): origin = next ray origin, tMin 0.01, tMax = distance to the light sample
      const F4 ro = ld4(&st.rayO[slot]), sd = ld4(&st.neeD[slot]), nc = ld4(&st.neeC[slot]);
      float t, u, v; uint32_t tri;
      const bool occluded = traverse<true, COUNT>(sc, s_nodes, ldsNodes, s_tris, ldsTris, s_stack, v3(ro.x, ro.y, ro.z), v3(sd.x, sd.y, sd.z), 0.01f, nc.w, t, u, v, tri, tc);
      if (!occluded) {
        F4 r = ld4(&st.rad[slot]);
        st4(&st.rad[slot], r.x + nc.x, r.y + nc.y, r.z + nc.z, r.w);
      }
    }
  }
  // statistics: one atomic per wave
  {
    unsigned long long r = rays;
    for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off);
    if (__lane_id() == 0 && r) atomicAdd(ANYHIT ? &cnt->shadowRays : &cnt->segments, r);
    if (COUNT) {
      unsigned long long a = tc.nodes, b = tc.tris;
      for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
      if (__lane_id() == 0) { atomicAdd(ANYHIT ? &cnt->shadowNodesVisited : &cnt->nodesVisited, a); atomicAdd(ANYHIT ? &cnt->shadowTrisTested : &cnt->trisTested, b); }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Shading state (mdl_shading_state.glsl:4-98) from flat scene buffers
// ------------------------------------------------------------------------------------------------
struct ShState { V3 normal, geomNormal, position, tangentU, tangentV; bool frontFace; uint32_t meshFlags, material; };

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
  const uint4 tail = reinterpret_cast<const uint4*>(sc.tris)[(size_t)triIdx * 3u + 2u]; // e2.z, instance, prim, origId
  const uint32_t instIdx = tail.y, prim = tail.z;
  const InstanceRec* inst = &sc.instances[instIdx];
  float o2w[12], w2o[9];
  {
    const float4* ip = reinterpret_cast<const float4*>(inst);
    float4 r0 = ip[0], r1 = ip[1], r2 = ip[2], r3 = ip[3], r4 = ip[4], r5 = ip[5];
    o2w[0] = r0.x; o2w[1] = r0.y; o2w[2] = r0.z; o2w[3] = r0.w; o2w[4] = r1.x; o2w[5] = r1.y; o2w[6] = r1.z; o2w[7] = r1.w;
    o2w[8] = r2.x; o2w[9] = r2.y; o2w[10] = r2.z; o2w[11] = r2.w;
    w2o[0] = r3.x; w2o[1] = r3.y; w2o[2] = r3.z; w2o[3] = r3.w; w2o[4] = r4.x; w2o[5] = r4.y; w2o[6] = r4.z; w2o[7] = r4.w; w2o[8] = r5.x;
    const uint4 m = *reinterpret_cast<const uint4*>(&sc.meshes[f2u(r5.y)]);
    s.material = m.z; s.meshFlags = m.w;
    const uint32_t* f = sc.faces + ((size_t)m.x + prim) * 3u;
    const uint32_t i0 = f[0], i1 = f[1], i2 = f[2];
    const float4* vb = reinterpret_cast<const float4*>(sc.verts + m.y);
    const float4 a1 = vb[2 * (size_t)i0], a2 = vb[2 * (size_t)i0 + 1];
    const float4 b1 = vb[2 * (size_t)i1], b2 = vb[2 * (size_t)i1 + 1];
    const float4 c1 = vb[2 * (size_t)i2], c2 = vb[2 * (size_t)i2 + 1];
    const float bx = 1.0f - hu - hv, by = hu, bz = hv;
    const V3 pa = v3(a1.x, a1.y, a1.z), pb = v3(b1.x, b1.y, b1.z), pc = v3(c1.x, c1.y, c1.z);
    const V3 localPos = (pa * bx + pb * by) + pc * bz;
    s.position = xform_point(o2w, localPos, 1.0f);
    V3 gn = normalize(cross(pb - pa, pc - pa));
    gn = normalize(xform_normal(w2o, gn));
    const V3 n0 = gi_decode_direction(f2u(a2.x)), n1 = gi_decode_direction(f2u(b2.x)), n2 = gi_decode_direction(f2u(c2.x));
    const V3 ln = normalize((n0 * bx + n1 * by) + n2 * bz);
    V3 nrm = normalize(xform_normal(w2o, ln));
    s.frontFace = dot(gn, -rayDir) >= 0.0f;
    if (!s.frontFace) { gn = -gn; nrm = -nrm; }
    const V3 t0 = gi_decode_direction(f2u(a2.y)), t1 = gi_decode_direction(f2u(b2.y)), t2 = gi_decode_direction(f2u(c2.y));
    const V3 lt = normalize((t0 * bx + t1 * by) + t2 * bz);
    V3 tg = normalize(xform_point(o2w, lt, 0.0f));
    tg = normalize(tg - nrm * dot(tg, nrm));
    const float bs = (bx * a1.w + by * b1.w) + bz * c1.w;
    s.tangentU = tg; s.tangentV = cross(nrm, tg) * bs;
    s.normal = nrm; s.geomNormal = gn;
  }
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

struct UpsParams { V3 albedo, F0; float alpha, coat, coatAlpha; };
__device__ __forceinline__ UpsParams ups_params(const MaterialRec* m)
{
  UpsParams u;
  V3 dc = v3(m->p[0], m->p[1], m->p[2]);
  float r = m->p[11], cr = m->p[13];
  u.alpha = fmax2(r * r, 0.001f);
  u.coatAlpha = fmax2(cr * cr, 0.001f);
  u.coat = m->p[12];
  if (m->p[6] != 0.0f) { u.F0 = v3(m->p[7], m->p[8], m->p[9]); u.albedo = dc; }
  else {
    float ior = m->p[16], metal = m->p[10];
    float q = (1.0f - ior) / (1.0f + ior); float f0 = q * q;
    u.F0 = v3(f0, f0, f0) * (1.0f - metal) + dc * metal;
    u.albedo = dc * (1.0f - metal);
  }
  return u;
}

struct BsdfSample { V3 k2, overPdf; float pdf; uint32_t event; };
struct BsdfEval { V3 diffuse, glossy; float pdf; };

__device__ inline void bsdf_sample(const MaterialRec* m, const ShState& st, V3 k1, float x0, float x1, float x2, BsdfSample& out)
{
  out.event = EV_ABSORB; out.pdf = 0.0f; out.overPdf = v3(0.0f, 0.0f, 0.0f); out.k2 = v3(0.0f, 0.0f, 0.0f);
  const uint32_t klass = m->klass;
  if (klass == 0u) {
    V3 l = gi_sample_hemisphere(x0, x1);
    V3 k2 = to_world(st, l);
    if (!(l.z > 0.0f) || !(dot(k2, st.geomNormal) > 0.0f)) return;
    out.k2 = k2; out.pdf = l.z / GI_PI; out.overPdf = v3(m->p[0], m->p[1], m->p[2]); out.event = EV_DIFFUSE | EV_REFLECTION;
    return;
  }
  if (klass == 1u) {
    UpsParams u = ups_params(m);
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
}

__device__ inline void bsdf_evaluate(const MaterialRec* m, const ShState& st, V3 k1, V3 k2, BsdfEval& out)
{
  out.diffuse = v3(0.0f, 0.0f, 0.0f); out.glossy = v3(0.0f, 0.0f, 0.0f); out.pdf = 0.0f;
  float nk2 = dot(st.normal, k2);
  if (!(nk2 > 0.0f)) return;
  const uint32_t klass = m->klass;
  if (klass == 0u) {
    float c = nk2 / GI_PI;
    out.diffuse = v3(m->p[0], m->p[1], m->p[2]) * c; out.pdf = c;
    return;
  }
  if (klass == 1u) {
    UpsParams u = ups_params(m);
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
// k_shade: closest-hit / miss shading + the post-trace part of the bounce loop
// (rp_main.chit:132-493, rp_main.miss:55-86, rp_main.rgen:397-480)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_shade(FrameUniforms U, SceneView sc, PathState st, const uint32_t* __restrict__ qCur,
                                               uint32_t* __restrict__ qNext, uint32_t* __restrict__ qRegen, uint32_t* __restrict__ qShadow,
                                               Counters* cnt, uint32_t curIdx, uint32_t nextIdx)
{
  const uint32_t n = cnt->count[curIdx];
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t slot = qCur[i];
    const F4 h = ld4(&st.hit[slot]);
    const F4 tb = ld4(&st.thr[slot]);
    const F4 rr = ld4(&st.rad[slot]);
    V3 throughput = v3(tb.x, tb.y, tb.z), radiance = v3(rr.x, rr.y, rr.z);
    uint32_t bitfield = f2u(tb.w), rng = f2u(rr.w);
    const uint32_t bounce = bitfield & 0x00000fffu;
    const uint32_t tri = f2u(h.w);
    bool shadow = false;

    if (tri == 0xffffffffu) {
      // miss: uniform fallback dome == colour-AOV clear value (rp_main.miss:68-86, Gi.cpp:2184-2199, 2232-2238)
      bitfield |= 0x80000000u;
      radiance = radiance + throughput * v3(U.background);
    } else {
      const F4 rd = ld4(&st.rayD[slot]);
      const V3 rayDir = v3(rd.x, rd.y, rd.z);
      ShState ss;
      setup_shading_state(sc, tri, h.y, h.z, rayDir, ss);
      const MaterialRec* mat = &sc.materials[ss.material];
      const bool isDoubleSided = (ss.meshFlags & 2u) != 0u;
      // emission (rp_main.chit:293-343): uniform EDF, radiance == emission colour where cos > 0
      const V3 em = v3(mat->p[3], mat->p[4], mat->p[5]);
      if (em.x != 0.0f || em.y != 0.0f || em.z != 0.0f) {
        if (ss.frontFace || !isDoubleSided) {
          const float c = dot(-rayDir, ss.normal);
          if (c > 0.0f) radiance = radiance + throughput * (em * U.exposureScale);
        }
      }
      // BSDF importance sampling (:361-389); xi = next4f, .w is drawn but unused by the closed forms
      const float x0 = gi_next1f(rng), x1 = gi_next1f(rng), x2 = gi_next1f(rng); (void)gi_next1f(rng);
      BsdfSample bs; bsdf_sample(mat, ss, -rayDir, x0, x1, x2, bs);
      throughput = throughput * bs.overPdf;
      const bool isTransmission = (bs.event & EV_TRANSMISSION) != 0u;
      // NEE (:394-444)
      if ((U.flags & FLAG_NEE) && (bs.event & (EV_DIFFUSE | EV_GLOSSY))) {
        const float k0 = gi_next1f(rng), k1 = gi_next1f(rng), k2 = gi_next1f(rng), k3 = gi_next1f(rng);
        V3 dirToLight, lightPower; float lightDist, invPdf; uint32_t ds;
        sample_light(sc, U, k0, k1, k2, k3, ss.position, dirToLight, lightDist, lightPower, invPdf, ds);
        V3 nee = v3(0.0f, 0.0f, 0.0f);
        if ((lightDist > 0.0f) && dot(dirToLight, ss.geomNormal) > 0.0f) {
          BsdfEval ev; bsdf_evaluate(mat, ss, -rayDir, dirToLight, ev);
          if (ev.pdf > 0.0f) {
            const float dmul = gi_half_to_float(ds & 0xffffu), smul = gi_half_to_float(ds >> 16);
            const V3 weight = throughput * (lightPower * invPdf);
            nee = nee + (weight * ev.diffuse) * dmul;
            nee = nee + (weight * ev.glossy) * smul;
          }
        }
        // rp_main.rgen:401-408: the shadow ray is traced only if it can contribute
        const V3 toLight = dirToLight * lightDist;
        const float ld = length(toLight);
        const V3 sdir = gi_safe_div(toLight, ld);
        shadow = gi_luminance(nee) > 1e-6f && ld > 1e-9f;
        if (shadow) { st4(&st.neeC[slot], nee.x, nee.y, nee.z, ld); st4(&st.neeD[slot], sdir.x, sdir.y, sdir.z, 0.0f); }
      }
      if (bs.event == EV_ABSORB) bitfield |= 0x80000000u; // :483-486
      const V3 gn = ss.geomNormal * (isTransmission ? -1.0f : 1.0f);
      const V3 no = gi_offset_ray_origin(ss.position, gn); // :488-489
      st4(&st.rayO[slot], no.x, no.y, no.z, 0.0f);
      st4(&st.rayD[slot], bs.k2.x, bs.k2.y, bs.k2.z, GI_FLT_MAX);
    }
    // rp_main.rgen:441-480
    if (length(throughput) < 1e-9f) bitfield |= 0x80000000u;
    if (bounce > U.rrBounceOffset) {
      const float k = gi_next1f(rng);
      const float mt = fmax2(throughput.x, fmax2(throughput.y, throughput.z));
      const float p = fmin2(mt, U.rrInvMinTermProb);
      if (k > p) bitfield |= 0x80000000u; else throughput = throughput / p;
    }
    bitfield++;
    const bool cont = ((bitfield & 0x00000fffu) < U.maxBounces) && !(bitfield & 0x80000000u); // loop test :298-304
    st4(&st.thr[slot], throughput.x, throughput.y, throughput.z, u2f(bitfield));
    st4(&st.rad[slot], radiance.x, radiance.y, radiance.z, u2f(rng));
    wave_append(cont, slot, qNext, &cnt->count[nextIdx]);
    wave_append(!cont, slot, qRegen, &cnt->count[Q_REGEN]);
    wave_append(shadow, slot, qShadow, &cnt->count[Q_SHADOW]);
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
  st.position = v3(0.0f, 0.0f, 0.0f); st.frontFace = true; st.meshFlags = 0u; st.material = 0u;
  BsdfSample bs; bsdf_sample(mat, st, v3(p + 12), p[18], p[19], p[20], bs);
  BsdfEval ev; bsdf_evaluate(mat, st, v3(p + 12), v3(p + 15), ev);
  o[0] = bs.k2.x; o[1] = bs.k2.y; o[2] = bs.k2.z; o[3] = bs.overPdf.x; o[4] = bs.overPdf.y; o[5] = bs.overPdf.z; o[6] = bs.pdf; o[7] = (float)bs.event;
  o[8] = ev.diffuse.x; o[9] = ev.diffuse.y; o[10] = ev.diffuse.z; o[11] = ev.glossy.x; o[12] = ev.glossy.y; o[13] = ev.glossy.z; o[14] = ev.pdf;
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------------
void launchInit(hipStream_t s, const PathState& st, uint32_t* qRegen, Counters* cnt, uint32_t n)
{
  uint32_t blocks = (n + 255u) / 256u; if (blocks > 4096u) blocks = 4096u; if (blocks == 0u) blocks = 1u;
  hipLaunchKernelGGL(k_init, dim3(blocks), dim3(256), 0, s, st, qRegen, cnt, n);
}
void launchReset(hipStream_t s, Counters* cnt, uint32_t a, uint32_t b, uint32_t c)
{
  hipLaunchKernelGGL(k_reset, dim3(1), dim3(64), 0, s, cnt, a, b, c);
}
void launchRaygen(hipStream_t s, uint32_t blocks, const FrameUniforms& U, const PathState& st, const uint32_t* qRegen, uint32_t* qTrace,
                  Counters* cnt, uint32_t traceIdx, F4* colorOut)
{
  hipLaunchKernelGGL(k_raygen, dim3(blocks), dim3(256), 0, s, U, st, qRegen, qTrace, cnt, traceIdx, colorOut);
}
void launchTrace(hipStream_t s, uint32_t blocks, bool anyHit, bool count, const SceneView& sc, const PathState& st, const uint32_t* queue,
                 Counters* cnt, uint32_t queueIdx)
{
  if (!anyHit) {
    if (count) hipLaunchKernelGGL((k_trace<false, true>), dim3(blocks), dim3(TRACE_BLOCK), 0, s, sc, st, queue, cnt, queueIdx);
    else hipLaunchKernelGGL((k_trace<false, false>), dim3(blocks), dim3(TRACE_BLOCK), 0, s, sc, st, queue, cnt, queueIdx);
  } else {
    if (count) hipLaunchKernelGGL((k_trace<true, true>), dim3(blocks), dim3(TRACE_BLOCK), 0, s, sc, st, queue, cnt, queueIdx);
    else hipLaunchKernelGGL((k_trace<true, false>), dim3(blocks), dim3(TRACE_BLOCK), 0, s, sc, st, queue, cnt, queueIdx);
  }
}
void launchShade(hipStream_t s, uint32_t blocks, const FrameUniforms& U, const SceneView& sc, const PathState& st, const uint32_t* qCur,
                 uint32_t* qNext, uint32_t* qRegen, uint32_t* qShadow, Counters* cnt, uint32_t curIdx, uint32_t nextIdx)
{
  hipLaunchKernelGGL(k_shade, dim3(blocks), dim3(256), 0, s, U, sc, st, qCur, qNext, qRegen, qShadow, cnt, curIdx, nextIdx);
}

void launchDebugBsdf(hipStream_t s, const MaterialRec* mat, uint32_t count, const float* in, float* out)
{
  hipLaunchKernelGGL(k_debug_bsdf, dim3((count + 63u) / 64u), dim3(64), 0, s, mat, count, in, out);
}

} // namespace gi
