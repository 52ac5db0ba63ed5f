// gi_debug.cpp -- giCTraceRays and the debug / self-check entry points
// (one of the translation units gi_c.cpp was split into in round 6; shared declarations: gi_host.h)
#include "gi_host.h"

// ---------------------------------------------------------------------------------------------------------------
// giCTraceRays: closest hits through the device traversal kernel (parity tests of the BVH8 path)
// ---------------------------------------------------------------------------------------------------------------
static int giCTraceRaysImpl(GiCScene* s, uint32_t count, const float* origins, const float* dirs, float tMin, float tMax, float* outTUV, int32_t* outInstPrim);
extern "C" int giCTraceRays(GiCScene* s, uint32_t count, const float* origins, const float* dirs, float tMin, float tMax, float* outTUV, int32_t* outInstPrim)
{
  try { return giCTraceRaysImpl(s, count, origins, dirs, tMin, tMax, outTUV, outInstPrim); }
  catch (const std::exception& e) { setError(std::string("giCTraceRays: ") + e.what()); return -1; }
}
static int giCTraceRaysImpl(GiCScene* s, uint32_t count, const float* origins, const float* dirs, float tMin, float tMax, float* outTUV, int32_t* outInstPrim)
{
  if (!g_ctx.initialized || !s || (count && (!origins || !dirs || !outTUV || !outInstPrim))) { setError("giCTraceRays: bad arguments"); return -1; }
  if (count == 0) return 0;
  std::lock_guard<std::mutex> guard(s->mutex);
  hipStream_t st = g_ctx.stream;
  if (syncSceneGeometry(s) != GI_C_OK) return -1;
  // the render loop's grids: k_trace_dyn (scenes beyond LDS) is persistent per wave and wants every resident wave slot filled (8 blocks per CU offered)
  const bool inLds = s->nodeCount <= 384u && s->triCount <= 128u;
  const uint32_t blocks = std::min<uint32_t>((count + 255u) / 256u, (uint32_t)g_ctx.cuCount * (inLds ? 3u : 8u));
  if (ensurePathState(s, count, blocks, blocks) != GI_C_OK) return -1;
  // ray records go straight into the TRACE_A queue (segment k holds rays [k*per, (k+1)*per))
  const size_t qn = (size_t)s->queueCap * NSHARD;
  std::vector<uint32_t> qslot(qn, 0u); std::vector<F4> qa(qn), qb(qn);
  Counters c{};
  const uint32_t per = (count + NSHARD - 1u) / NSHARD;
  for (uint32_t k = 0; k < NSHARD; k++) { uint32_t lo = k * per; c.count[Q_TRACE_A][k].v = lo < count ? std::min(per, count - lo) : 0u; }
  for (uint32_t i = 0; i < count; i++) {
    size_t r = (size_t)(i / per) * s->queueCap + (i % per);
    qslot[r] = i;
    qa[r] = F4{origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], tMin};
    qb[r] = F4{dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], tMax};
  }
  if (hipMemcpyAsync(s->qSlot[Q_TRACE_A].ptr, qslot.data(), qn * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(s->qA[Q_TRACE_A].ptr, qa.data(), qn * sizeof(F4), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(s->qB[Q_TRACE_A].ptr, qb.data(), qn * sizeof(F4), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(s->dCounters.ptr, &c, sizeof(c), hipMemcpyHostToDevice, st) != hipSuccess) { setError("giCTraceRays: upload failed"); return -1; }
  // Cut-out materials: the any-hit test of a closest-hit walk draws from the path's random state, which it reads from the ray's Slot -- here slot i of a pool no
  // path has written (or that an earlier render left behind).  The rays of this entry point carry the state 0, like the oracle's (found by tests/fuzz_parity.py:
  // a quarter of the random scenes with cut-outs answered giCTraceRays with other triangles than the oracle, and not the same ones twice).
  if (hipMemsetAsync(s->slots.ptr, 0, (size_t)count * sizeof(Slot), st) != hipSuccess) { setError("giCTraceRays: clearing the slots failed"); return -1; }
  PathState ps{s->slots.ptr, nullptr, 0u, nullptr, 0u, nullptr};
  // (no TRACE_FRESH entries: the uniforms are not read)
  launchTrace(st, blocks, false, false, makeView(s), ps, makeQueueSet(s), s->dCounters.ptr, Q_TRACE_A, Q_REGEN_B, traceDynRefill(s), blocks, FrameUniforms{},
      nullptr);
  std::vector<TriRec> tris(s->triCount);
  if (hipMemcpyAsync(&c, s->dCounters.ptr, sizeof(c), hipMemcpyDeviceToHost, st) != hipSuccess ||
      (s->triCount && hipMemcpyAsync(tris.data(), s->dTris.ptr, s->triCount * sizeof(TriRec), hipMemcpyDeviceToHost, st) != hipSuccess) ||
      hipStreamSynchronize(st) != hipSuccess) { setError("giCTraceRays: readback failed"); return -1; }
  std::vector<F4> hit(count, F4{tMax, 0.0f, 0.0f, 0.0f});
  for (uint32_t i = 0; i < count; i++) { uint32_t m = 0xffffffffu; memcpy(&hit[i].w, &m, 4); }
  // results stay in the ray records (a = t, u, v, triangle | class << 28); the class queues hold their indices
  std::vector<uint32_t> hitIdx(qn);
  if (hipMemcpy(qa.data(), s->qA[Q_TRACE_A].ptr, qn * sizeof(F4), hipMemcpyDeviceToHost) != hipSuccess) {
    setError("giCTraceRays: readback failed");
    return -1;
  }
  for (uint32_t klass = 0; klass < MAT_CLASS_COUNT; klass++) {
    if (hipMemcpy(hitIdx.data(), s->qSlot[Q_HIT + klass].ptr, qn * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      setError("giCTraceRays: readback failed");
      return -1;
    }
    for (uint32_t k = 0; k < NSHARD; k++)
      for (uint32_t j = 0; j < c.count[Q_HIT + klass][k].v; j++) {
        const uint32_t ri = hitIdx[(size_t)k * s->queueCap + j] & 0x3fffffffu;
        if (ri < qn && qslot[ri] < count) { F4 h = qa[ri]; uint32_t w; memcpy(&w, &h.w, 4); w &= 0x0fffffffu; memcpy(&h.w, &w, 4); hit[qslot[ri]] = h; }
      }
  }
  int hits = 0;
  for (uint32_t i = 0; i < count; i++) {
    uint32_t tri; memcpy(&tri, &hit[i].w, 4);
    outTUV[3 * i] = hit[i].x; outTUV[3 * i + 1] = hit[i].y; outTUV[3 * i + 2] = hit[i].z;
    if (tri == 0xffffffffu || tri >= s->triCount) { outInstPrim[2 * i] = -1; outInstPrim[2 * i + 1] = -1; }
    else { outInstPrim[2 * i] = (int32_t)tris[tri].instance; outInstPrim[2 * i + 1] = (int32_t)tris[tri].prim; hits++; }
  }
  return hits;
}

// ---------------------------------------------------------------------------------------------------------------
// giCDebugValidateBvh: host-only check of the builder's conservativeness contract
// ---------------------------------------------------------------------------------------------------------------
static int validateTree(const std::vector<Node8>& nodes, const std::vector<TriRec>& trisArr, uint32_t triCount)
{
  int violations = 0;
  std::vector<uint8_t> seen(triCount, 0);
  struct Item { uint32_t node; float lo[3], hi[3]; };
  std::vector<Item> stack;
  Item root; root.node = 0; for (int a = 0; a < 3; a++) { root.lo[a] = -3.0e38f; root.hi[a] = 3.0e38f; }
  stack.push_back(root);
  while (!stack.empty()) {
    Item it = stack.back(); stack.pop_back();
    if (it.node >= nodes.size()) { violations++; continue; }
    const Node8& n = nodes[it.node];
    uint32_t rel = 0;
    for (int s = 0; s < 8; s++) {
      uint8_t meta = n.meta[s];
      if (meta == 0) { if (n.imask & (1u << s)) violations++; continue; }
      float lo[3], hi[3];
      for (int a = 0; a < 3; a++) {
        uint32_t eb = (uint32_t)n.e[a] << 23; float scale; memcpy(&scale, &eb, 4);
        lo[a] = n.p[a] + (float)n.qlo[a][s] * scale; hi[a] = n.p[a] + (float)n.qhi[a][s] * scale;
      }
      bool inner = (n.imask >> s) & 1u;
      if (inner) {
        if ((meta >> 5) != 1u || (meta & 31u) != 24u + (uint32_t)s) violations++;
        Item c; c.node = n.childBase + rel; rel++;
        for (int a = 0; a < 3; a++) { c.lo[a] = lo[a]; c.hi[a] = hi[a]; }
        // every triangle below must also be inside all ancestors: intersect the constraint boxes
        for (int a = 0; a < 3; a++) { c.lo[a] = std::max(c.lo[a], it.lo[a]); c.hi[a] = std::min(c.hi[a], it.hi[a]); }
        stack.push_back(c);
      } else {
        uint32_t unary = meta >> 5, off = meta & 31u, cnt = unary == 1u ? 1u : unary == 3u ? 2u : unary == 7u ? 3u : 0u;
        if (cnt == 0u || off + cnt > 24u) { violations++; continue; }
        for (uint32_t k = 0; k < cnt; k++) {
          uint32_t ti = n.triBase + off + k;
          if (ti >= trisArr.size()) { violations++; continue; }
          const TriRec& t = trisArr[ti];
          if (t.origId >= triCount || seen[t.origId]) { violations++; continue; }
          seen[t.origId] = 1;
          for (int v = 0; v < 3; v++)
            for (int a = 0; a < 3; a++) {
              float x = t.v0[a] + (v == 1 ? t.e1[a] : v == 2 ? t.e2[a] : 0.0f);
              float blo = std::max(lo[a], it.lo[a]), bhi = std::min(hi[a], it.hi[a]);
              if (x < blo || x > bhi) violations++;
            }
        }
      }
    }
  }
  // every active triangle sits in exactly one leaf; an inactive one (bvh8.h: a vertex that is not finite or beyond 1e18) in none
  std::vector<uint8_t> inactive(triCount, 0);
  for (const TriRec& t : trisArr) {
    if (t.origId >= triCount) { violations++; continue; }
    for (int a = 0; a < 3; a++) {
      const float x0 = t.v0[a], x1 = t.v0[a] + t.e1[a], x2 = t.v0[a] + t.e2[a];
      if (!(std::fabs(x0) <= 1.0e18f) || !(std::fabs(x1) <= 1.0e18f) || !(std::fabs(x2) <= 1.0e18f)) inactive[t.origId] = 1;
    }
  }
  for (uint32_t i = 0; i < triCount; i++) if ((seen[i] != 0) == (inactive[i] != 0)) violations++;
  return violations;
}

extern "C" int giCDebugValidateBvh(const float* triVerts, uint32_t triCount, uint32_t* outNodeCount, uint32_t* outMaxDepth)
{
  if (triCount && !triVerts) return -1;
  std::vector<TriRec> tris(triCount);
  for (uint32_t i = 0; i < triCount; i++) {
    const float* p = triVerts + 9 * (size_t)i;
    for (int a = 0; a < 3; a++) { tris[i].v0[a] = p[a]; tris[i].e1[a] = p[3 + a] - p[a]; tris[i].e2[a] = p[6 + a] - p[a]; }
    tris[i].instance = 0; tris[i].prim = i; tris[i].origId = i;
  }
  Bvh8 bvh; buildBvh8(tris, bvh);
  if (outNodeCount) *outNodeCount = (uint32_t)bvh.nodes.size();
  if (outMaxDepth) *outMaxDepth = bvh.maxDepth;
  return validateTree(bvh.nodes, bvh.tris, triCount);
}

// The same check for the PARTITIONED layout of incremental updates: the triangles are cut into `partCount` consecutive ranges, every range gets its own
// subtree in its own node range, and a top tree over the subtree roots (buildTopBvh8) joins them.  Returns the violations of the assembled tree.
extern "C" int giCDebugValidatePartitionedBvh(const float* triVerts, uint32_t triCount, uint32_t partCount, uint32_t* outNodeCount, uint32_t* outMaxDepth)
{
  if (!triVerts || triCount == 0 || partCount == 0 || partCount > triCount) return -1;
  const uint32_t topCap = partCount * 2u + 16u;
  std::vector<Node8> nodes(topCap, Node8{}); std::vector<TriRec> trisAll(triCount);
  std::vector<float> boxes(6 * (size_t)partCount); std::vector<Node8> roots(partCount);
  uint32_t subDepth = 0;
  for (uint32_t pi = 0; pi < partCount; pi++) {
    const uint32_t first = (uint32_t)((uint64_t)triCount * pi / partCount), end = (uint32_t)((uint64_t)triCount * (pi + 1) / partCount);
    std::vector<TriRec> tris(end - first);
    for (uint32_t i = first; i < end; i++) {
      const float* p = triVerts + 9 * (size_t)i; TriRec& t = tris[i - first];
      for (int a = 0; a < 3; a++) { t.v0[a] = p[a]; t.e1[a] = p[3 + a] - p[a]; t.e2[a] = p[6 + a] - p[a]; }
      t.instance = pi; t.prim = i - first; t.origId = i - first;
    }
    Bvh8 b; buildBvh8(tris, b);
    const uint32_t off = (uint32_t)nodes.size();
    for (Node8 n : b.nodes) { n.childBase += off; n.triBase += first; nodes.push_back(n); }
    for (uint32_t k = 0; k < end - first; k++) { TriRec t = b.tris[k]; t.origId += first; trisAll[first + k] = t; }
    roots[pi] = nodes[off]; nodeBounds(nodes[off], &boxes[6 * (size_t)pi]);
    subDepth = std::max(subDepth, b.maxDepth);
  }
  Bvh8 top; buildTopBvh8(boxes.data(), partCount, roots.data(), top);
  if (top.nodes.size() > topCap) return -2;
  std::copy(top.nodes.begin(), top.nodes.end(), nodes.begin());
  if (outNodeCount) *outNodeCount = (uint32_t)nodes.size();
  if (outMaxDepth) *outMaxDepth = top.maxDepth + subDepth - 1u;
  return validateTree(nodes, trisAll, triCount);
}

// giCDebugShadeClass: which k_shade variant an (untextured) material's hits are binned for -- host only
extern "C" int giCDebugShadeClass(const GiCMaterialDesc* desc)
{
  if (!desc) return -1;
  MaterialRec m{}; m.klass = desc->klass; m.flags = desc->flags & ~(MAT_FLAG_TEXTURED | MAT_FLAG_OPACITY_TEX); memcpy(m.p, desc->p, sizeof(m.p));
  deriveMaterialConstants(m);
  return (int)shadeClassOf(m);
}

// ---------------------------------------------------------------------------------------------------------------
// giCDebugEvalBsdf: closed-form BSDF sample/evaluate on the device for explicit shading frames
// ---------------------------------------------------------------------------------------------------------------
extern "C" int giCDebugEvalBsdf(const GiCMaterialDesc* desc, uint32_t count, const float* in, float* out)
{
  if (!g_ctx.initialized || !desc || (count && (!in || !out))) { setError("giCDebugEvalBsdf: bad arguments"); return GI_C_ERROR; }
  if (count == 0) return GI_C_OK;
  MaterialRec m{}; m.klass = desc->klass; m.flags = desc->flags & ~(MAT_FLAG_TEXTURED | MAT_FLAG_OPACITY_TEX); memcpy(m.p, desc->p, sizeof(m.p));
  deriveMaterialConstants(m);
  // the variant the render would shade this material's hits with (GATLING_OPTIONS=shade_variants=0: always the full closed form)
  const uint32_t shadeClass = shadeClassOf(m);
  MaterialRec* dm = nullptr; float* din = nullptr; float* dout = nullptr;
  hipStream_t st = g_ctx.stream;
  int rc = GI_C_ERROR;
  if (hipMalloc((void**)&dm, sizeof(m)) == hipSuccess && hipMalloc((void**)&din, (size_t)count * 22 * 4) == hipSuccess &&
      hipMalloc((void**)&dout, (size_t)count * 15 * 4) == hipSuccess &&
      hipMemcpyAsync(dm, &m, sizeof(m), hipMemcpyHostToDevice, st) == hipSuccess &&
      hipMemcpyAsync(din, in, (size_t)count * 22 * 4, hipMemcpyHostToDevice, st) == hipSuccess) {
    launchDebugBsdf(st, dm, shadeClass, count, din, dout);
    if (hipMemcpyAsync(out, dout, (size_t)count * 15 * 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) rc = GI_C_OK;
  }
  if (rc != GI_C_OK) setError("giCDebugEvalBsdf: HIP failure");
  if (dm) (void)hipFree(dm); if (din) (void)hipFree(din); if (dout) (void)hipFree(dout);
  return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// giCDebugTexRuntime: the MDL renderer runtime's remaining texture entry points (tex_texel_float4_2d, tex_resolution_2d, tex_lookup_float4_3d,
// tex_texel_float4_3d) on the device, for explicit queries
// ---------------------------------------------------------------------------------------------------------------
extern "C" int giCDebugTexRuntime(const float* rgba, uint32_t width, uint32_t height, uint32_t depth, uint32_t count, const float* queries, float* out)
{
  if (!g_ctx.initialized || !rgba || !width || !height || !depth || (count && (!queries || !out))) { setError("giCDebugTexRuntime: bad arguments");
      return GI_C_ERROR; }
  if (count == 0) return GI_C_OK;
  const size_t texFloats = (size_t)width * height * depth * 4;
  float* dt = nullptr; float* dq = nullptr; float* dout = nullptr;
  hipStream_t st = g_ctx.stream;
  int rc = GI_C_ERROR;
  if (hipMalloc((void**)&dt, texFloats * 4) == hipSuccess && hipMalloc((void**)&dq, (size_t)count * 32) == hipSuccess
      && hipMalloc((void**)&dout, (size_t)count * 16) == hipSuccess &&
      hipMemcpyAsync(dt, rgba, texFloats * 4, hipMemcpyHostToDevice, st) == hipSuccess
          && hipMemcpyAsync(dq, queries, (size_t)count * 32, hipMemcpyHostToDevice, st) == hipSuccess) {
    launchDebugTex(st, dt, width, height, depth, count, dq, dout);
    if (hipMemcpyAsync(out, dout, (size_t)count * 16, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) rc = GI_C_OK;
  }
  if (rc != GI_C_OK) setError("giCDebugTexRuntime: HIP failure");
  if (dt) (void)hipFree(dt); if (dq) (void)hipFree(dq); if (dout) (void)hipFree(dout);
  return rc;
}


// ---------------------------------------------------------------------------------------------------------------
// giCDebugCheckSqrt: the kernels' square root (gi_device_math.h gi_sqrt) against the compiler's correctly rounded sqrtf for the `count` bit patterns from `first` on
// (count = 2^32: every float).  Returns the number of arguments whose results differ, or -1 on error.
// ---------------------------------------------------------------------------------------------------------------
extern "C" int64_t giCDebugCheckSqrt(uint32_t first, uint64_t count)
{
  if (!g_ctx.initialized) { setError("giCDebugCheckSqrt before giCInitialize"); return -1; }
  unsigned long long* d = nullptr; unsigned long long h = 0ull;
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) { setError("giCDebugCheckSqrt: hipMalloc failed"); return -1; }
  bool ok = hipMemsetAsync(d, 0, sizeof(h), g_ctx.stream) == hipSuccess;
  if (ok) { launchDebugSqrt(g_ctx.stream, first, (unsigned long long)count, d); ok = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, g_ctx.stream) == hipSuccess; }
  ok = ok && hipStreamSynchronize(g_ctx.stream) == hipSuccess;
  (void)hipFree(d);
  if (!ok) { setError("giCDebugCheckSqrt: device error"); return -1; }
  return (int64_t)h;
}
