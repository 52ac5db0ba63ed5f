// gi_build.cpp -- scene build (flatten, pack, BVH8, two-level layout, upload) and incremental transform updates (Gi.cpp:784-1315)
// (one of the translation units gi_c.cpp was split into in round 6; shared declarations: gi_host.h)
#include "gi_host.h"

// ---------------------------------------------------------------------------------------------------------------
// scene build: flatten instances into world space, pack vertex data, build + upload the BVH8
// ---------------------------------------------------------------------------------------------------------------

// world = local * M_prim * M_instance with USD row vectors (Gi.cpp:641-658, 1191); returns rows of the 3x4
// column-vector affine.  Same operation order as glm's mat4 * mat4.
void composeTransform(const float* prim, const float* inst, float out[12])
{
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 4; r++) {
      float acc = prim[r * 4 + 0] * inst[0 * 4 + c];
      acc = acc + prim[r * 4 + 1] * inst[1 * 4 + c];
      acc = acc + prim[r * 4 + 2] * inst[2 * 4 + c];
      acc = acc + prim[r * 4 + 3] * inst[3 * 4 + c];
      out[c * 4 + r] = acc;
    }
}

void invert3x3(const float a[12], float inv[9])
{
  double m[3][3] = {{a[0], a[1], a[2]}, {a[4], a[5], a[6]}, {a[8], a[9], a[10]}};
  double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
  double c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
  double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  double det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
  double id = 1.0 / det;
  inv[0] = (float)(c00 * id);
  inv[1] = (float)((m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id);
  inv[2] = (float)((m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id);
  inv[3] = (float)(c01 * id);
  inv[4] = (float)((m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id);
  inv[5] = (float)((m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id);
  inv[6] = (float)(c02 * id);
  inv[7] = (float)((m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id);
  inv[8] = (float)((m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id);
}

inline void xformPoint(const float a[12], const float p[3], float out[3])
{
  out[0] = ((a[0] * p[0] + a[1] * p[1]) + a[2] * p[2]) + a[3];
  out[1] = ((a[4] * p[0] + a[5] * p[1]) + a[6] * p[2]) + a[7];
  out[2] = ((a[8] * p[0] + a[9] * p[1]) + a[10] * p[2]) + a[11];
}

// Shade class of a material (gi_types.h MAT_CLASS_COUNT): its BSDF class, or -- the reference's per-material feature #defines done the wavefront way,
// GlslShaderGen.cpp:204-274, Gi.cpp:1545-1562 -- the specialised variant its hits are binned and shaded by. OpenPBR BASE: every optional lobe absent (no coat,
// fuzz, thin film, anisotropy, transmission, subsurface, not thin-walled), no bound texture / primvar input, every parameter finite (the variant drops products
// with exact zeros, which a NaN or an infinity would not honour). GATLING_OPTIONS=shade_variants=0 keeps every material in its full kernel (tests: same bits).
uint32_t shadeClassOf(const MaterialRec& m)
{
  if (m.klass != GI_C_MAT_OPEN_PBR || optionValue("shade_variants", 1) == 0) return m.klass & 0xfu;
  if (m.flags & MAT_FLAG_TEXTURED) return m.klass;
  for (uint32_t i = 0; i < MAT_PARAM_COUNT; i++) if (!std::isfinite(m.p[i])) return m.klass;
  if ((uint32_t)m.p[MP_FEATURES] != 0u) return m.klass;
  if (m.p[MP_COAT] != 0.0f || m.p[GI_C_P_CLEARCOAT] != 0.0f || m.p[GI_C_P_TRANSMISSION_WEIGHT] != 0.0f) return m.klass;
  return SHADE_CLASS_OPBR_BASE;
}

// Hostile geometry (bvh8.h "Inactive items").  A coordinate the build works with: finite, at most 1e18 in magnitude.
inline bool usableCoordinate(float x) { return std::fabs(x) <= 1.0e18f; } // (false for NaN)
// An instance the flattening can use: every entry of its affine finite and its 3x3 invertible with an inverse that is finite in fp32 (w2o transforms normals
// and, in the two-level layout, rays). Every triangle of an instance that is not -- a NaN or singular giCSetMeshTransform / instance transform -- is inactive.
inline bool usableInstance(const InstanceRec& ir)
{
  for (int i = 0; i < 12; i++) if (!std::isfinite(ir.o2w[i])) return false;
  for (int i = 0; i < 9; i++) if (!std::isfinite(ir.w2o[i])) return false;
  return true;
}
// Shading attributes of a vertex as the scene build takes them: a normal or tangent with a non-finite component becomes +Z, a non-finite texture coordinate 0,
// a non-finite bitangent sign +1 (the position is left alone: it decides whether the triangle is active). The reference uploads what it is given
// (Gi.cpp:848-861) and a NaN attribute is a NaN pixel there; here hostile attributes cost the shading of the faces that use them, nothing else.
inline GiCVertex usableShadingAttributes(const GiCVertex& in)
{
  GiCVertex v = in;
  auto direction = [](float* d) { if (!std::isfinite(d[0]) || !std::isfinite(d[1]) || !std::isfinite(d[2])) { d[0] = 0.0f; d[1] = 0.0f; d[2] = 1.0f; } };
  direction(v.norm); direction(v.tangent);
  if (!std::isfinite(v.u)) v.u = 0.0f;
  if (!std::isfinite(v.v)) v.v = 0.0f;
  if (!std::isfinite(v.bitangentSign)) v.bitangentSign = 1.0f;
  return v;
}
// one flattened triangle (Gi.cpp:1188-1202 hands the instance transform to the TLAS; here it is applied); `usable` false: marked inactive for the builder
inline void flattenTriangle(const InstanceRec& ir, bool usable, const GiCMesh* m, uint32_t f, TriRec& t)
{
  float p0[3], p1[3], p2[3];
  xformPoint(ir.o2w, m->vertices[m->faces[f].v_i[0]].pos, p0);
  xformPoint(ir.o2w, m->vertices[m->faces[f].v_i[1]].pos, p1);
  xformPoint(ir.o2w, m->vertices[m->faces[f].v_i[2]].pos, p2);
  for (int a = 0; a < 3; a++) { t.v0[a] = p0[a]; t.e1[a] = p1[a] - p0[a]; t.e2[a] = p2[a] - p0[a]; }
  if (!usable) t.v0[0] = std::numeric_limits<float>::quiet_NaN();
}

// Two-level layout (SceneView::tlasNodes ...): built next to the flat BVH for instanced scenes that do not fit LDS.  The flat
// arrays stay (k_shade reads the hit's TriRec, k_aov / giCTraceRays traverse them); the two-level ones are what k_trace_dyn2 walks,
// and they are small: one BLAS per MESH instead of one subtree per instance, so traversal stays in the caches.
template <class MB>
int buildTwoLevel(GiCScene* s, const std::vector<MB>& meshBuilds, const std::vector<InstanceRec>& instances, size_t flatTris, size_t flatNodes,
    TwoLevelHost& out)
{
  s->twoLevel = false;
  int want = s->optTwoLevel;
  want = (int)optionValue("two_level", want);
  size_t uniqueTris = 0;
  for (const MB& mb : meshBuilds) uniqueTris += mb.instCount ? mb.m->faces.size() : 0;
  const bool beyondLds = flatNodes > 384u || flatTris > 128u;
  (void)uniqueTris;
  // Opt-in only. Measured (r01k): although its working set is tiny (C4: 1.5 MB of BLAS nodes + 2.6 MB of mesh triangles instead of 41 + 335 MB) the first
  // version is SLOWER than the flat layout -- C4 trace 185 -> 207 ms, C5 834 -> 1945 ms (29 instead of 20 nodes per ray: overlapping instance boxes, each visit
  // pays a ray transform, a BLAS root and a restore; candidates cost a rebuild). ... except where the flat traversal cannot address the scene: its
  // wave-cooperative triangle ring packs (lane, flat triangle) into 32 bits, 2^26 triangles; the two-level walk queues MESH triangles there (one BLAS per
  // mesh), so heavily instanced scenes beyond that bound take it automatically (r04; the hit record's triangle word, flat index | class << 28, then bounds the
  // scene at 2^28 flattened triangles)
  if (flatTris >= ((size_t)1 << 26) && want < 0) want = 1;
  // (the flat walk addresses nodes by 32-bit byte offset, gi_traversal.h node_load: 53 M nodes -- beyond any 2^26-triangle tree)
  if (flatNodes * sizeof(Node8) >= ((size_t)1 << 32) && want < 0) want = 1;
  if (want <= 0 || instances.empty() || !beyondLds) return GI_C_OK;
  std::vector<Node8> blasNodes; std::vector<BlasTri> blasTris; std::vector<InstTrav> instTrav(instances.size());
  uint32_t blasDepth = 0;
  std::vector<float> instBoxes(instances.size() * 6);
  auto padBox = [](float* lo, float* hi) { // as bvh8.cpp pads triangle boxes: 2^-20 relative, covers the rounding of the exact test's inputs
    for (int a = 0; a < 3; a++) { const float mag = std::max(std::fabs(lo[a]), std::fabs(hi[a])) + (hi[a] - lo[a]);
        const float pad = mag * 9.5367431640625e-7f + 1.0e-30f; lo[a] -= pad; hi[a] += pad; }
  };
  for (const MB& mb : meshBuilds) {
    if (mb.instCount == 0) continue;
    const GiCMesh* m = mb.m;
    const size_t nf = m->faces.size();
    std::vector<float> boxes(nf * 6);
    float mlo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (size_t f = 0; f < nf; f++) {
      float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
      bool faceOk = true; // (an unusable face keeps an inverted box: the builder leaves it out, and it must not widen the mesh magnitude below)
      for (int k = 0; k < 3; k++) { const float* p = m->vertices[m->faces[f].v_i[k]].pos;
          for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); faceOk = faceOk && usableCoordinate(p[a]); } }
      if (faceOk) padBox(lo, hi); else for (int a = 0; a < 3; a++) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; }
      for (int a = 0; a < 3; a++) { boxes[6 * f + a] = lo[a]; boxes[6 * f + 3 + a] = hi[a]; }
      if (faceOk) for (int a = 0; a < 3; a++) { mlo[a] = std::min(mlo[a], lo[a]); mhi[a] = std::max(mhi[a], hi[a]); }
    }
    Bvh8 b; std::vector<uint32_t> order;
    buildBvh8Boxes(boxes.data(), nf, b, order);
    const uint32_t nodeBase = (uint32_t)blasNodes.size(), triBase = (uint32_t)blasTris.size();
    for (Node8 n : b.nodes) { n.childBase += nodeBase; n.triBase += triBase; blasNodes.push_back(n); }
    for (uint32_t f : order) {
      BlasTri bt{};
      memcpy(bt.p0, m->vertices[m->faces[f].v_i[0]].pos, 12); memcpy(bt.p1, m->vertices[m->faces[f].v_i[1]].pos, 12);
          memcpy(bt.p2, m->vertices[m->faces[f].v_i[2]].pos, 12);
      bt.prim = f;
      blasTris.push_back(bt);
    }
    blasDepth = std::max(blasDepth, b.maxDepth);
    // object-space magnitude the transformed ray's rounding error scales with inside this mesh (see wave_step2)
    const float extent = (std::fabs(mlo[0]) + std::fabs(mlo[1]) + std::fabs(mlo[2])) + (std::fabs(mhi[0]) + std::fabs(mhi[1]) + std::fabs(mhi[2]));
    for (uint32_t ii = 0; ii < mb.instCount; ii++) {
      const uint32_t inst = mb.instFirst + ii;
      InstTrav& tv = instTrav[inst];
      tv = InstTrav{};
      memcpy(tv.o2w, instances[inst].o2w, sizeof(tv.o2w)); memcpy(tv.w2o, instances[inst].w2o, sizeof(tv.w2o));
      tv.blasRoot = nodeBase; tv.triBase = mb.triFirst + ii * (uint32_t)nf; tv.matFlags = mb.matFlags; tv.slack = extent;
      float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
      // inactive triangles (bvh8.h): a face with an unusable OBJECT-space vertex is left out of the BLAS by the builder and out of this box; an unusable
      // instance keeps the inverted box (the builder leaves it out of the TLAS). A usable face whose WORLD-space vertex is unusable is inactive in the flat
      // tree but would be walked here: such scenes keep the flat layout
      if (usableInstance(instances[inst]))
        for (size_t f = 0; f < nf; f++) {
          bool objectOk = true, worldOk = true; float q[3][3];
          for (int k = 0; k < 3; k++) {
            const float* o = m->vertices[m->faces[f].v_i[k]].pos;
            xformPoint(instances[inst].o2w, o, q[k]);
            for (int a = 0; a < 3; a++) { objectOk = objectOk && usableCoordinate(o[a]); worldOk = worldOk && usableCoordinate(q[k][a]); }
          }
          if (!objectOk) continue;
          if (!worldOk) {
            if (want > 0) fprintf(stderr,
                "[gatling_gi] two-level layout not used: an instance carries triangles that leave the usable coordinate range in world space\n");
            return GI_C_OK;
          }
          for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], q[k][a]); hi[a] = std::max(hi[a], q[k][a]); }
        }
      padBox(lo, hi);
      for (int a = 0; a < 3; a++) { instBoxes[6 * inst + a] = lo[a]; instBoxes[6 * inst + 3 + a] = hi[a]; }
    }
  }
  Bvh8 tlas; std::vector<uint32_t> tlasItems;
  buildBvh8Boxes(instBoxes.data(), instances.size(), tlas, tlasItems);
  // per-lane stack: a TLAS level can leave a node group and an instance group behind, a BLAS level a node group
  if (blasTris.size() >= ((size_t)1 << 26)) {
    if (want > 0) fprintf(stderr, "[gatling_gi] two-level layout not used: 2^26 or more unique mesh triangles\n");
    return GI_C_OK;
  }
  if (2u * tlas.maxDepth + blasDepth + 1u > 16u) {
    if (want > 0) fprintf(stderr, "[gatling_gi] two-level layout not used: trees too deep for the 16-entry stack\n");
    return GI_C_OK;
  }
  s->twoLevel = true;
  if (getenv("GATLING_BUILD_TIMING")) fprintf(stderr,
      "[gatling_gi] two-level: TLAS %zu nodes over %zu instances, %zu BLAS nodes, %zu mesh triangles (flat: %zu nodes, %zu triangles)\n",
                                              tlas.nodes.size(), instances.size(), blasNodes.size(), blasTris.size(), flatNodes, flatTris);
  out.tlasNodes.swap(tlas.nodes); out.tlasItems.swap(tlasItems); out.blasNodes.swap(blasNodes); out.blasTris.swap(blasTris); out.instTrav.swap(instTrav);
  return GI_C_OK;
}

// ... and its upload into one device's memory (the primary's and every replica's: multi-device renders replicate the scene)
int uploadSceneTo(GiCScene* s, SceneDevice& D, const SceneHost& H)
{
  const DevCtx& ctx = g_ctx.devs[D.slot];
  HIP_TRY(hipSetDevice(ctx.device));
  hipStream_t st = ctx.stream;
  if (s->twoLevel) {
    if (D.dTlasNodes.upload(H.two.tlasNodes, st) || D.dTlasItems.upload(H.two.tlasItems, st) || D.dBlasNodes.upload(H.two.blasNodes, st)
        || D.dBlasTris.upload(H.two.blasTris, st) ||
        D.dInstTrav.upload(H.two.instTrav, st) || D.dFlatOfOrig.upload(H.flatOfOrig, st))
      return GI_C_ERROR;
  }
  if (D.dTriFaceId.upload(H.triFaceId, st) || D.dTriShade.upload(H.triShade, st) || D.dTriGeomNormal.upload(H.triGeomNormal, st)) return GI_C_ERROR;
  if (D.dMeshes.upload(H.meshRecs, st) || D.dSceneData.upload(H.sceneData, st)) return GI_C_ERROR;
  { // textures: one device array per image + the TextureRec table
    for (auto* b : D.dTexels) { b->release(); delete b; }
    D.dTexels.clear();
    std::vector<TextureRec> recs(s->textures.size());
    for (size_t i = 0; i < s->textures.size(); i++) {
      auto* b = new DeviceBuffer<float>();
      D.dTexels.push_back(b);
      if (b->upload(s->textures[i]->rgba, st)) return GI_C_ERROR;
      recs[i] = TextureRec{b->ptr, s->textures[i]->width, s->textures[i]->height};
    }
    if (D.dTextures.upload(recs, st)) return GI_C_ERROR;
    HIP_TRY(hipStreamSynchronize(st)); // `recs` goes out of scope
  }
  if (D.dNodes.upload(H.bvh.nodes, st) || D.dTris.upload(H.bvh.tris, st) || D.dInstances.upload(H.instances, st) ||
      D.dVerts.upload(H.verts, st) || D.dMaterials.upload(H.mats, st))
    return GI_C_ERROR;
  HIP_TRY(hipStreamSynchronize(st)); // host vectors may go out of scope
  return GI_C_OK;
}

// devices a render of this scene may use (replicas exist for slots 1 .. n-1 after buildScene)
uint32_t sceneDeviceCount(const GiCScene* s)
{
  uint32_t n = (uint32_t)g_ctx.devs.size();
  if (s->optDevices > 0) n = std::min<uint32_t>(n, (uint32_t)s->optDevices);
  return std::max(n, 1u);
}
SceneDevice& sceneDevice(GiCScene* s, uint32_t slot) { return slot == 0u ? static_cast<SceneDevice&>(*s) : *s->replicas[slot - 1u]; }

// The flat tree's root bounds for FLAG_BOUNDS_RETIRE: the dequantised child boxes of node 0 (which contain every triangle's padded box), padded once more by
// 1e-5 of their magnitude and extent -- k_raygen's slab test adds its own per-ray rounding allowance on top.
static void setSceneBounds(GiCScene* s, const std::vector<Node8>& nodes)
{
  s->boundsValid = false;
  if (nodes.empty()) return;
  float b[6]; nodeBounds(nodes[0], b);
  for (int a = 0; a < 3; a++) {
    if (!(b[a] <= b[3 + a]) || !std::isfinite(b[a]) || !std::isfinite(b[3 + a])) return; // empty root (no triangles) or overflowing planes: no early retire
    const float pad = (std::fabs(b[a]) + std::fabs(b[3 + a]) + (b[3 + a] - b[a])) * 1.0e-5f + 1.0e-30f;
    s->bounds[a] = b[a] - pad; s->bounds[3 + a] = b[3 + a] + pad;
  }
  s->boundsValid = true;
}

int buildScene(GiCScene* s)
{
  double t0 = nowMs();
  std::unique_ptr<SceneHost> hostPtr(new SceneHost());
  SceneHost& H = *hostPtr;
  s->host.reset(); // (a failed build leaves no stale host copy behind)
  for (GiCMesh* m : s->meshes) { m->builtInstances = 0xffffffffu; m->xformDirty = false; m->instDirty.clear(); }
  std::vector<FVertex>& verts = H.verts; std::vector<InstanceRec>& instances = H.instances; std::vector<TriRec> tris; std::vector<int32_t> faceIdOf;
  std::vector<MaterialRec>& mats = H.mats; mats.resize(s->materials.size());
  for (size_t i = 0; i < s->materials.size(); i++) {
    mats[i].klass = s->materials[i]->desc.klass; mats[i].flags = s->materials[i]->desc.flags & ~(MAT_FLAG_TEXTURED | MAT_FLAG_OPACITY_TEX);
    for (uint32_t slot = 0; slot < TEX_SLOT_COUNT; slot++) {
      const GiCTextureBinding& b = s->materials[i]->tex[slot];
      TexBindingRec& r = mats[i].tex[slot];
      r = TexBindingRec{};
      auto tit = b.texture ? std::find(s->textures.begin(), s->textures.end(), b.texture) : s->textures.end();
      if (tit == s->textures.end()) {
        if (!s->materials[i]->primvarInput[slot].empty()) {
          r.mode = TEX_MODE_PRIMVAR; mats[i].flags |= MAT_FLAG_TEXTURED;
          // Frontend.cpp:251-252: named scene data answered from the UBO
          if (s->materials[i]->primvarInput[slot] == "CAMERA_POSITION") r.mode |= TEX_MODE_CAMERA_POSITION;
          if (s->materials[i]->primvarInput[slot] == "FRAME") r.mode |= TEX_MODE_FRAME;
        }
        continue;
      }
      r.tex = (uint32_t)(tit - s->textures.begin()) + 1u;
      r.mode = (uint32_t)b.wrapS | ((uint32_t)b.wrapT << 8) | (((uint32_t)b.channel & 3u) << 16);
      memcpy(r.scale, b.scale, 16); memcpy(r.bias, b.bias, 16);
      if (s->materials[i]->hasTexXf[slot]) { r.mode |= TEX_MODE_XFORM; memcpy(r.xf, s->materials[i]->texXf[slot], sizeof(r.xf)); }
      mats[i].flags |= slot == TEX_OPACITY ? MAT_FLAG_OPACITY_TEX : MAT_FLAG_TEXTURED; // opacity is looked up by the any-hit test, not by k_shade
    }
    memcpy(mats[i].p, s->materials[i]->desc.p, sizeof(float) * MAT_PARAM_COUNT);
    deriveMaterialConstants(mats[i]);
  }
  uint32_t meshIdx = 0;
  std::vector<MeshBuild>& meshBuilds = H.meshBuilds; // visible meshes in scene order (two-level layout, incremental updates)
  std::vector<MeshRec>& meshRecs = H.meshRecs; std::vector<float>& sceneData = H.sceneData;
  s->classMask = 0; s->hasCutouts = false; s->classTextured = 0; s->shadeClassMask = 0; s->shadeClassTextured = 0;
  for (GiCMesh* m : s->meshes) {
    if (!m->visible) continue; // Gi.cpp:801-804
    if (m->faces.empty()) continue;
    auto mit = std::find(s->materials.begin(), s->materials.end(), m->material);
    if (mit == s->materials.end()) { fprintf(stderr, "[gatling_gi] invalid BLAS material for mesh %s\n", m->name.c_str()); continue; } // Gi.cpp:818-822
    const uint32_t material = (uint32_t)(mit - s->materials.begin());
    if (material > 0x00ffffffu) { setError("too many materials"); return GI_C_ERROR; }
    const bool cutoutMat = mats[material].p[MP_CUTOUT] < 1.0f || (mats[material].flags & MAT_FLAG_OPACITY_TEX) != 0u;
    if (cutoutMat) s->hasCutouts = true;
    const uint32_t shadeClass = shadeClassOf(mats[material]);
    const uint32_t matFlags = material | (shadeClass << 24) | (cutoutMat ? (1u << 28) : 0u) | (((m->flipFacing ? 1u : 0u) | (m->doubleSided ? 2u : 0u)) << 30);
    s->classMask |= 1u << (mats[material].klass & 0xfu); s->shadeClassMask |= 1u << shadeClass;
    if (mats[material].flags & MAT_FLAG_TEXTURED) { s->classTextured |= 1u << (mats[material].klass & 0xfu); s->shadeClassTextured |= 1u << shadeClass; }
    const uint32_t vertexOffset = (uint32_t)verts.size();
    { // scene data the mesh's material reads (Gi.cpp:905-1019): instancer primvars first, mesh primvars override, by name
      MeshRec mr{}; mr.vertexOffset = vertexOffset;
      for (uint32_t slot = 0; slot < TEX_SLOT_COUNT; slot++) {
        const std::string& want = (*mit)->primvarInput[slot];
        if (want.empty()) continue;
        const GiCPrimvar* pv = nullptr;
        for (const GiCPrimvar& p : m->instancerPrimvars) if (p.name == want && !p.data.empty()) { pv = &p; break; }
        for (const GiCPrimvar& p : m->primvars) if (p.name == want && !p.data.empty()) { pv = &p; break; }
        if (!pv) continue; // SCENE_DATA_INVALID
        const bool isInt = pv->type > GI_C_PRIMVAR_VEC4; // Int .. Int4 (Gi.h:76-79)
        const uint32_t stride = (uint32_t)(isInt ? pv->type - GI_C_PRIMVAR_INT : pv->type) + 1u;
        size_t entries = 1; // what a lookup can index: zero-padded so that short arrays read 0 like the oracle
        if (pv->interpolation == GI_C_INTERP_VERTEX) entries = m->vertices.size();
        else if (pv->interpolation == GI_C_INTERP_UNIFORM) entries = m->faces.size();
        else if (pv->interpolation == GI_C_INTERP_INSTANCE) { int32_t mx = (int32_t)(m->instanceTransforms.size() / 16) - 1;
            for (int32_t id : m->instanceIds) mx = std::max(mx, id); entries = (size_t)std::max(mx, 0) + 1; }
        mr.sdOffset[slot] = (uint32_t)sceneData.size();
        mr.sdInfo[slot] = 1u | ((stride - 1u) << 1) | ((uint32_t)pv->interpolation << 3) | (isInt ? SD_INFO_INT : 0u);
        // (+ 2: a three-component lookup of a one- or two-component primvar reads up to two floats past its last entry -- the reference's stride arithmetic,
        // mdl_interface.glsl:343-349, finds its neighbour in the packed buffer there; here, as in the oracle, zeros.  Found by tests/fuzz_parity.py)
        const size_t need = std::max(entries * stride, pv->data.size()) + 2u;
        sceneData.insert(sceneData.end(), pv->data.begin(), pv->data.end());
        sceneData.resize(mr.sdOffset[slot] + need, 0.0f);
      }
      meshRecs.push_back(mr);
    }
    for (const GiCVertex& vIn : m->vertices) { // Gi.cpp:848-861: quantise normal/tangent to octahedral unorm2x16, then decode once
      const GiCVertex v = usableShadingAttributes(vIn);
      FVertex fv; memcpy(fv.pos, v.pos, 12); fv.bsign = v.bitangentSign;
      decodeDirection(encodeDirection(v.norm), fv.normal); decodeDirection(encodeDirection(v.tangent), fv.tangent);
      fv.u = v.u; fv.v = v.v;
      verts.push_back(fv);
    }
    // FaceId AOV values, bug-compatible: face ids are stored with a 1/2/4-byte stride chosen from maxFaceId (Gi.cpp:878-885);
    // the shader fetches the 32-bit word prim / (4/stride), shifts it by (prim % (4/stride)) * 8 bits (sic) and masks it
    // with (stride*8 - 1) (rp_main.chit:231-240).  Evaluated once per primitive here.
    std::vector<int32_t> meshFaceIdAov(m->faces.size());
    {
      const int stride = m->maxFaceId <= 255u ? 1 : (m->maxFaceId <= 65535u ? 2 : 4), invStride = 4 / stride;
      std::vector<uint8_t> packed(((size_t)m->faces.size() * stride + 3) / 4 * 4, 0);
      for (size_t i = 0; i < m->faces.size(); i++) { int32_t fid = i < m->faceIds.size() ? m->faceIds[i] : 0; memcpy(&packed[i * stride], &fid, stride); }
      for (size_t i = 0; i < m->faces.size(); i++) {
        int32_t word; memcpy(&word, &packed[(i / (size_t)invStride) * 4], 4);
        word >>= (int)((i % (size_t)invStride) * 8);
        meshFaceIdAov[i] = word & (stride * 8 - 1);
      }
    }
    size_t instCount = m->instanceTransforms.size() / 16;
    m->builtInstances = (uint32_t)instCount;
    meshBuilds.push_back(MeshBuild{m, vertexOffset, matFlags, (uint32_t)instances.size(), (uint32_t)instCount, (uint32_t)tris.size(), meshIdx, meshFaceIdAov});
    for (size_t ii = 0; ii < instCount; ii++) { // Gi.cpp:1188-1202
      InstanceRec ir{};
      composeTransform(m->transform, &m->instanceTransforms[16 * ii], ir.o2w);
      invert3x3(ir.o2w, ir.w2o);
      ir.mesh = meshIdx; ir.instanceId = ii < m->instanceIds.size() ? m->instanceIds[ii] : (int32_t)ii;
      ir.pad = (uint32_t)m->id; // object id
      uint32_t instIdx = (uint32_t)instances.size();
      instances.push_back(ir);
      const bool usable = usableInstance(ir);
      for (uint32_t f = 0; f < (uint32_t)m->faces.size(); f++) {
        TriRec t;
        flattenTriangle(ir, usable, m, f, t);
        for (int a = 0; a < 3; a++) t.vi[a] = vertexOffset + m->faces[f].v_i[a];
        t.instance = instIdx; t.prim = f; t.origId = (uint32_t)tris.size(); t.matFlags = matFlags;
        tris.push_back(t);
        faceIdOf.push_back(meshFaceIdAov[f]);
      }
    }
    meshIdx++;
  }
  Bvh8& bvh = H.bvh;
  buildBvh8(tris, bvh);
  { std::vector<TriRec>().swap(tris); } // the BVH holds its own (leaf-ordered) copy
  s->stats.inactiveTriangleCount = (uint32_t)bvh.tris.size() - bvh.activeTris;
  if (bvh.activeTris < bvh.tris.size()) { // one line per mesh (bvh8.h "Inactive items")
    std::vector<uint32_t> perMesh(meshBuilds.size(), 0u);
    for (size_t i = bvh.activeTris; i < bvh.tris.size(); i++) perMesh[instances[bvh.tris[i].instance].mesh]++;
    for (const MeshBuild& mb : meshBuilds)
      if (perMesh[mb.meshIdx]) fprintf(stderr, "[gatling_gi] warning: mesh %s: %u of %zu instanced triangle(s) have a non-finite or out-of-range (> 1e18) "
                                                "vertex or a non-invertible transform and are inactive\n",
                                       mb.m->name.c_str(), perMesh[mb.meshIdx], mb.m->faces.size() * (size_t)mb.instCount);
  }
  if (buildTwoLevel(s, meshBuilds, instances, bvh.tris.size(), bvh.nodes.size(), H.two) != GI_C_OK) return GI_C_ERROR;
  if (s->twoLevel) {
    H.flatOfOrig.resize(bvh.tris.size());
    for (size_t i = 0; i < bvh.tris.size(); i++) H.flatOfOrig[bvh.tris[i].origId] = (uint32_t)i;
  }
  double t1 = nowMs();
  // the deepest traversal variant keeps 8 (SPILL8) or 16 stack entries in LDS and OVF_STACK = 40 in scratch; trav_node_pick does not bound-check the spill
  if (bvh.maxDepth > 1u + 8u + 40u) { setError("scene BVH is deeper than the traversal stack (49 levels): degenerate geometry (long chains of nested splits)");
      return GI_C_ERROR; }
  if (bvh.tris.size() >= (1u << 26) && !s->twoLevel) {
    setError("scene has 2^26 or more triangles after instancing and no two-level layout (it is switched off, or its unique mesh triangles exceed 2^26 too): "
             "the traversal queues pack (lane, triangle) into 32 bits");
    return GI_C_ERROR;
  }
  if (bvh.tris.size() >= (1u << 28)) {
    setError("scene has 2^28 or more triangles after instancing: the hit record packs (triangle, material class) into 32 bits");
    return GI_C_ERROR;
  }
  H.triFaceId.resize(bvh.tris.size());
  for (size_t i = 0; i < bvh.tris.size(); i++) H.triFaceId[i] = faceIdOf[bvh.tris[i].origId];
  // Scenes beyond LDS: one 160-byte shading record per mesh triangle (gi_types.h TriShade); the flattened triangles name theirs in vi[0].  LDS-resident
  // scenes keep vertex indices there: the fused kernels are VALU-bound and read the host-decoded FVertex records.
  H.shadePacked = bvh.nodes.size() > 384u || bvh.tris.size() > 128u;
  H.triShade.clear();
  if (H.shadePacked) {
    std::vector<uint32_t> shadeBaseOfMesh(meshBuilds.size(), 0u);
    for (MeshBuild& mb : meshBuilds) {
      mb.shadeBase = (uint32_t)H.triShade.size(); shadeBaseOfMesh[mb.meshIdx] = mb.shadeBase;
      const GiCMesh* m = mb.m;
      for (const GiCFace& f : m->faces) {
        TriShade q{};
        for (int k = 0; k < 3; k++) {
          const GiCVertex v = usableShadingAttributes(m->vertices[f.v_i[k]]);
          // (Gi.cpp:848-861: quantised, then decoded once)
          memcpy(q.p[k], v.pos, 12); decodeDirection(encodeDirection(v.norm), q.n[k]); decodeDirection(encodeDirection(v.tangent), q.t[k]);
          q.uv[k][0] = v.u; q.uv[k][1] = v.v; q.bsign[k] = v.bitangentSign; q.vi[k] = mb.vertexOffset + f.v_i[k];
        }
        H.triShade.push_back(q);
      }
    }
    for (TriRec& t : bvh.tris) t.vi[0] = shadeBaseOfMesh[instances[t.instance].mesh] + t.prim;
  }
  // LDS-resident scenes (the fused kernels' and k_trace's shading path reads FVertex records): the world-space geometric normal of every flattened triangle,
  // made here with setup_shading_state's operations in its order (mdl_shading_state.glsl:27-28: normalize(cross(pb - pa, pc - pa)) in object space, the normal
  // transform, normalize again) -- two normalisations, a cross product and a transform per hit that depend on nothing but the triangle.  Same bits: IEEE + - *
  // / sqrt without contraction on both sides, as for the decoded FVertex normals.
  H.triGeomNormal.clear();
  if (!H.shadePacked) {
    H.triGeomNormal.resize(bvh.tris.size());
    auto normalize3 = [](float* a) { const float inv = 1.0f / sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]); a[0] = a[0] * inv; a[1] = a[1] * inv;
        a[2] = a[2] * inv; };
    for (size_t i = 0; i < bvh.tris.size(); i++) {
      const TriRec& t = bvh.tris[i];
      const float* pa = verts[t.vi[0]].pos; const float* pb = verts[t.vi[1]].pos; const float* pc = verts[t.vi[2]].pos;
      const float e1[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]}, e2[3] = {pc[0] - pa[0], pc[1] - pa[1], pc[2] - pa[2]};
      float g[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      normalize3(g);
      const float* w = instances[t.instance].w2o;
      float n[3] = {(g[0] * w[0] + g[1] * w[3]) + g[2] * w[6], (g[0] * w[1] + g[1] * w[4]) + g[2] * w[7], (g[0] * w[2] + g[1] * w[5]) + g[2] * w[8]};
      normalize3(n);
      H.triGeomNormal[i] = F4{n[0], n[1], n[2], 0.0f};
    }
  }
  s->shadePacked = H.shadePacked;
  // a new tree: the shadow walks' order is chosen anew
  s->shadowOrder = -1; s->shadowOrderRays[0] = s->shadowOrderRays[1] = s->shadowOrderSteps[0] = s->shadowOrderSteps[1] = 0;
  // one copy of the scene per device this scene renders on
  const uint32_t nDev = sceneDeviceCount(s);
  // a new replica has no lights yet
  while (s->replicas.size() + 1u < nDev) { s->replicas.emplace_back(new SceneDevice()); s->replicas.back()->slot = (uint32_t)s->replicas.size();
      s->dirty |= DIRTY_LIGHTS; }
  for (uint32_t d = 0; d < nDev; d++)
    if (uploadSceneTo(s, sceneDevice(s, d), H) != GI_C_OK) { (void)hipSetDevice(g_ctx.device); return GI_C_ERROR; }
  HIP_TRY(hipSetDevice(g_ctx.device));
  // stack entries a walk can need: a pick at level L pushes the rest of level L-1's group (gi_traversal.h trav_node_pick), the root level pushes nothing
  s->nodeCount = (uint32_t)bvh.nodes.size(); s->triCount = (uint32_t)bvh.tris.size(); s->bvhDepth = bvh.maxDepth > 1u ? bvh.maxDepth - 1u : 1u;
  setSceneBounds(s, bvh.nodes);
  s->stats.bvhBuildMs = t1 - t0; s->stats.uploadMs = nowMs() - t1;
  s->stats.nodeCount = s->nodeCount; s->stats.triangleCount = s->triCount;
  if (getenv("GATLING_BUILD_TIMING")) fprintf(stderr, "[gatling_gi] scene: %u nodes, %u triangles, %u levels (traversal stack need %u)\n", s->nodeCount,
      s->triCount, bvh.maxDepth, s->bvhDepth);
  s->host = std::move(hostPtr);
  return GI_C_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Incremental transform updates (the reference keeps every mesh's BLAS and rebuilds only the TLAS, Gi.cpp:1180-1202).
//
// A scene is first built as ONE tree over all instanced triangles (buildScene: the best tree).  The first time only transforms change, it is re-laid out
// PARTITIONED: every flattened mesh instance gets its own subtree in its own node range and keeps its triangles in its own (scene-order) range; a top tree
// over the subtree roots (buildTopBvh8: the roots are copied in as ordinary internal children) makes it one ordinary BVH8 again -- the traversal kernels, the
// shading code and the triangle ids do not change, so images stay bit-identical to a full rebuild (traversal contract: results do not depend on the tree).
// From then on moving an instance costs: its triangles re-transformed, its subtree rebuilt (a few thousand triangles), the top tree rebuilt (one item per
// instance), and those ranges uploaded -- not a 10 M-triangle SAH build and a 0.7 GB upload.  Any other edit (geometry, materials, visibility, instance
// counts) raises DIRTY_BVH and the next render rebuilds everything as one tree again.
// ---------------------------------------------------------------------------------------------------------------
void nodeBounds(const Node8& n, float box[6])
{
  for (int a = 0; a < 3; a++) { box[a] = 3.0e38f; box[3 + a] = -3.0e38f; }
  for (int sl = 0; sl < 8; sl++) {
    if (n.meta[sl] == 0) continue;
    for (int a = 0; a < 3; a++) {
      uint32_t eb = (uint32_t)n.e[a] << 23; float scale; memcpy(&scale, &eb, 4);
      box[a] = std::min(box[a], n.p[a] + (float)n.qlo[a][sl] * scale); box[3 + a] = std::max(box[3 + a], n.p[a] + (float)n.qhi[a][sl] * scale);
    }
  }
  // the dequantised planes are evaluated in fp32 here and with an fma on the device: one more ulp-scale pad keeps the item box outside both
  for (int a = 0; a < 3; a++) { const float pad = (std::fabs(box[a]) + std::fabs(box[3 + a])) * 2.4e-7f + 1.0e-30f; box[a] -= pad; box[3 + a] += pad; }
}

// One instance's InstanceRec, world-space triangles (scene order) and subtree
struct PartBuild { InstanceRec inst; Bvh8 bvh; };
void buildPart(const MeshBuild& mb, uint32_t instInMesh, bool packed, PartBuild& out)
{
  const GiCMesh* m = mb.m;
  InstanceRec ir{};
  composeTransform(m->transform, &m->instanceTransforms[16 * (size_t)instInMesh], ir.o2w);
  invert3x3(ir.o2w, ir.w2o);
  ir.mesh = mb.meshIdx; ir.instanceId = instInMesh < m->instanceIds.size() ? m->instanceIds[instInMesh] : (int32_t)instInMesh;
  ir.pad = (uint32_t)m->id;
  out.inst = ir;
  const uint32_t nf = (uint32_t)m->faces.size(), instIdx = mb.instFirst + instInMesh;
  std::vector<TriRec> tris(nf);
  const bool usable = usableInstance(ir);
  for (uint32_t f = 0; f < nf; f++) { // as buildScene
    TriRec& t = tris[f];
    flattenTriangle(ir, usable, m, f, t);
    for (int a = 0; a < 3; a++) t.vi[a] = mb.vertexOffset + m->faces[f].v_i[a];
    t.instance = instIdx; t.prim = f; t.origId = f; t.matFlags = mb.matFlags;
    if (packed) t.vi[0] = mb.shadeBase + f;
  }
  buildBvh8(tris, out.bvh);
}

// writes a built part into the scene arrays at the part's ranges (node / triangle indices rebased to absolute)
void placePart(SceneHost& H, InstPart& P, const PartBuild& B)
{
  const MeshBuild& mb = H.meshBuilds[P.meshBuild];
  P.nodeCount = (uint32_t)B.bvh.nodes.size(); P.depth = B.bvh.maxDepth;
  for (uint32_t i = 0; i < P.nodeCount; i++) { Node8 n = B.bvh.nodes[i]; n.childBase += P.nodeOff; n.triBase += P.triFirst; H.bvh.nodes[P.nodeOff + i] = n; }
  for (uint32_t k = 0; k < P.nf; k++) {
    TriRec t = B.bvh.tris[k];
    H.triFaceId[P.triFirst + k] = mb.faceIdAov[t.prim];
    t.origId += P.triFirst; // scene-order id: the instance's triangles are numbered in face order from triFirst, as in buildScene
    H.bvh.tris[P.triFirst + k] = t;
  }
  H.instances[mb.instFirst + P.instInMesh] = B.inst;
  nodeBounds(H.bvh.nodes[P.nodeOff], P.box);
}

template <class Fn> void parallelOver(size_t n, Fn&& fn)
{
  int workers = (int)std::thread::hardware_concurrency();
  if (const char* e = getenv("GATLING_BUILD_THREADS")) workers = atoi(e);
  workers = (int)std::min<size_t>((size_t)std::min(std::max(workers, 1), 32), std::max<size_t>(n, 1));
  if (workers <= 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (int w = 0; w < workers; w++) th.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) fn(i); });
  for (auto& t : th) t.join();
}

int rebuildTop(GiCScene* s, SceneHost& H)
{
  std::vector<float> boxes(H.parts.size() * 6); std::vector<Node8> roots(H.parts.size());
  uint32_t subDepth = 0;
  for (size_t i = 0; i < H.parts.size(); i++) { memcpy(&boxes[6 * i], H.parts[i].box, 24); roots[i] = H.bvh.nodes[H.parts[i].nodeOff];
      subDepth = std::max(subDepth, H.parts[i].depth); }
  Bvh8 top;
  buildTopBvh8(boxes.data(), H.parts.size(), roots.data(), top);
  if (top.nodes.size() > H.topCap) { setError("internal: top tree larger than its reserved range"); return GI_C_ERROR; }
  std::copy(top.nodes.begin(), top.nodes.end(), H.bvh.nodes.begin());
  for (size_t i = top.nodes.size(); i < H.topCap; i++) memset(&H.bvh.nodes[i], 0, sizeof(Node8));
  H.bvh.maxDepth = top.maxDepth + (subDepth > 0u ? subDepth - 1u : 0u); // the copied roots are the subtrees' first level
  if (H.bvh.maxDepth > 1u + 8u + 40u) { setError("scene BVH is deeper than the traversal stack (49 levels)"); return GI_C_ERROR; }
  s->bvhDepth = H.bvh.maxDepth > 1u ? H.bvh.maxDepth - 1u : 1u;
  return GI_C_OK;
}

// true: handled incrementally; false: the caller must run a full buildScene (not an error)
int updateTransforms(GiCScene* s, bool& handled)
{
  handled = false;
  if (!s->host || s->twoLevel || s->triCount < 4096u) return GI_C_OK; // small scenes rebuild in no time (and must stay LDS-resident)
  if (!optionValue("incremental", 1)) return GI_C_OK;
  SceneHost& H = *s->host;
  for (const MeshBuild& mb : H.meshBuilds) if (mb.m->builtInstances != mb.instCount) return GI_C_OK; // (cannot happen: count changes raise DIRTY_BVH)
  const double t0 = nowMs();
  std::vector<uint32_t> dirtyParts;
  bool converted = false;
  if (!H.partitioned) {
    // --- one-time re-layout: every instance its own subtree + ranges (costs about one full build, in parallel over the instances)
    std::vector<InstPart> parts;
    for (uint32_t b = 0; b < (uint32_t)H.meshBuilds.size(); b++) {
      const MeshBuild& mb = H.meshBuilds[b];
      const uint32_t nf = (uint32_t)mb.m->faces.size();
      for (uint32_t ii = 0; ii < mb.instCount; ii++) { InstPart P{}; P.meshBuild = b; P.instInMesh = ii; P.triFirst = mb.triFirst + ii * nf; P.nf = nf;
          parts.push_back(P); }
    }
    if (parts.empty()) return GI_C_OK;
    std::vector<PartBuild> built(parts.size());
    parallelOver(parts.size(), [&](size_t i) { buildPart(H.meshBuilds[parts[i].meshBuild], parts[i].instInMesh, H.shadePacked, built[i]); });
    H.topCap = (uint32_t)parts.size() * 2u + 16u; // top nodes <= internal top nodes + one copied root per part
    uint32_t off = H.topCap;
    for (size_t i = 0; i < parts.size(); i++) { const uint32_t n = (uint32_t)built[i].bvh.nodes.size(); parts[i].nodeOff = off;
        parts[i].nodeCap = n + n / 4u + 8u; off += parts[i].nodeCap; }
    H.bvh.nodes.assign(off, Node8{});
    H.parts.swap(parts);
    parallelOver(H.parts.size(), [&](size_t i) { placePart(H, H.parts[i], built[i]); });
    H.partitioned = true; converted = true;
  } else {
    for (uint32_t i = 0; i < (uint32_t)H.parts.size(); i++) {
      const GiCMesh* m = H.meshBuilds[H.parts[i].meshBuild].m;
      if (m->xformDirty && (m->instDirty.empty() || m->instDirty[H.parts[i].instInMesh])) dirtyParts.push_back(i);
    }
    std::vector<PartBuild> built(dirtyParts.size());
    parallelOver(dirtyParts.size(),
        [&](size_t k) { const InstPart& P = H.parts[dirtyParts[k]]; buildPart(H.meshBuilds[P.meshBuild], P.instInMesh, H.shadePacked, built[k]); });
    for (size_t k = 0; k < dirtyParts.size(); k++)
      // a subtree outgrew its range (rare): full rebuild
      if (built[k].bvh.nodes.size() > H.parts[dirtyParts[k]].nodeCap) { H.partitioned = false; H.parts.clear(); return GI_C_OK; }
    parallelOver(dirtyParts.size(), [&](size_t k) { placePart(H, H.parts[dirtyParts[k]], built[k]); });
  }
  if (rebuildTop(s, H) != GI_C_OK) return GI_C_ERROR;
  for (GiCMesh* m : s->meshes) { m->xformDirty = false; m->instDirty.clear(); }
  const double t1 = nowMs();
  // --- upload: everything after the re-layout, else the moved parts' ranges, their InstanceRecs and the top region
  const uint32_t nDev = std::min<uint32_t>(sceneDeviceCount(s), (uint32_t)s->replicas.size() + 1u);
  s->nodeCount = (uint32_t)H.bvh.nodes.size();
  setSceneBounds(s, H.bvh.nodes);
  for (uint32_t d = 0; d < nDev; d++) {
    SceneDevice& D = sceneDevice(s, d);
    if (converted) { if (uploadSceneTo(s, D, H) != GI_C_OK) { (void)hipSetDevice(g_ctx.device); return GI_C_ERROR; } continue; }
    const DevCtx& ctx = g_ctx.devs[d];
    HIP_TRY(hipSetDevice(ctx.device));
    hipStream_t st = ctx.stream;
    HIP_TRY(hipMemcpyAsync(D.dNodes.ptr, H.bvh.nodes.data(), (size_t)H.topCap * sizeof(Node8), hipMemcpyHostToDevice, st));
    for (uint32_t i : dirtyParts) {
      const InstPart& P = H.parts[i];
      const uint32_t instIdx = H.meshBuilds[P.meshBuild].instFirst + P.instInMesh;
      HIP_TRY(hipMemcpyAsync(D.dNodes.ptr + P.nodeOff, &H.bvh.nodes[P.nodeOff], (size_t)P.nodeCount * sizeof(Node8), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(D.dTris.ptr + P.triFirst, &H.bvh.tris[P.triFirst], (size_t)P.nf * sizeof(TriRec), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(D.dTriFaceId.ptr + P.triFirst, &H.triFaceId[P.triFirst], (size_t)P.nf * sizeof(int32_t), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(D.dInstances.ptr + instIdx, &H.instances[instIdx], sizeof(InstanceRec), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
  }
  HIP_TRY(hipSetDevice(g_ctx.device));
  s->stats.bvhBuildMs = t1 - t0; s->stats.uploadMs = nowMs() - t1;
  s->stats.nodeCount = s->nodeCount; s->stats.triangleCount = s->triCount;
  if (getenv("GATLING_BUILD_TIMING")) fprintf(stderr, "[gatling_gi] transform update: %s, %zu part(s) rebuilt of %zu, host %.1f ms, upload %.1f ms\n",
                                              converted ? "scene re-laid out as per-instance subtrees" : "incremental", converted
                                                  ? H.parts.size() : dirtyParts.size(), H.parts.size(), t1 - t0, nowMs() - t1);
  handled = true;
  return GI_C_OK;
}

// brings the device scene up to date with the host-side edits: incremental for transform-only edits, else a full build
int syncSceneGeometry(GiCScene* s)
{
  if ((s->dirty & DIRTY_XFORM) && !(s->dirty & (DIRTY_BVH | DIRTY_MATERIALS))) { // only transforms changed: re-transform / re-braid those instances
    bool handled = false;
    if (updateTransforms(s, handled) != GI_C_OK) return GI_C_ERROR;
    if (!handled) s->dirty |= DIRTY_BVH;
    s->dirty |= DIRTY_FRAMEBUFFER;
  }
  if (s->dirty & (DIRTY_BVH | DIRTY_MATERIALS)) {
    if (buildScene(s) != GI_C_OK) return GI_C_ERROR;
    s->dirty &= ~(DIRTY_BVH | DIRTY_MATERIALS); s->dirty |= DIRTY_FRAMEBUFFER;
  }
  s->dirty &= ~DIRTY_XFORM;
  return GI_C_OK;
}

