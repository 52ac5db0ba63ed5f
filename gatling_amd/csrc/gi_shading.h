// gi_shading.h -- everything a hit needs: path-payload bit fields, shading state (mdl_shading_state.glsl), the texture / scene-data
// runtime (mdl_interface.glsl), the dome light lookup (rp_main.miss), the closed-form BSDFs and light sampling (rp_main.chit).
// Included by gi_kernels.hip only.
#pragma once

#include "gi_queues.h"
#include "gi_texture.h"

namespace gi {

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

// NEE AOV bookkeeping (rp_main.rgen:431-435): the AOV shows the outcome of the shadow test at bounce 0 of the pixel's LAST sample
// (a shadow ray that cannot contribute is dispatched with an empty interval and counts as "not shadowed").  Samples retire out of
// order here, so every bounce-0 outcome is recorded as (sample + 1) << 1 | shadowed under atomicMax; PathState::neeKey is only
// set when the AOV is bound AND next-event estimation is on (the reference compiles the block out otherwise).
__device__ __forceinline__ void nee_aov_record_px(const PathState& st, uint32_t pixelLocal, uint32_t sampleLocal, bool shadowed)
{
  const unsigned long long order = (unsigned long long)(st.neeSampleBase + sampleLocal) + 1ull;
  atomicMax(&st.neeKey[pixelLocal], (order << 1) | (shadowed ? 1ull : 0ull));
}
__device__ __forceinline__ void nee_aov_record(const PathState& st, uint32_t slot, bool shadowed)
{
  const F4 id = ld4(&st.slots[slot].id);
  nee_aov_record_px(st, f2u(id.x), f2u(id.y), shadowed);
}
// ------------------------------------------------------------------------------------------------
// Shading state (mdl_shading_state.glsl:4-98) from flat scene buffers
// ------------------------------------------------------------------------------------------------
struct ShState {
  V3 normal, geomNormal, position, tangentU, tangentV; bool frontFace; uint32_t meshFlags, material;
  float u, v;                    // texture coordinate 0 (mdl_shading_state.glsl:62-65)
  uint32_t mesh, prim, vi[3]; int32_t instanceId; float hu, hv; // renderer state for scene-data lookups (mdl_interface.glsl:281-301)
  float ior1, ior2;              // Bsdf_sample_data.ior1/ior2 (rp_main.chit:188-189): < 0 = the material's own; 0 = empty-stack default
  bool thinWalled;               // mdl_thin_walled (rp_main.chit:155-157), set by shade_segment
  bool sssVolume;                // the render keeps a medium stack: OpenPBR's volumetric subsurface lobe is live (set by shade_segment)
  // OpenPBR geometry_coat_normal (open_pbr_surface.mtlx:87, 560): the coat lobe's own frame (resolve_material_textures)
  bool hasCoatFrame; V3 coatNormal, coatTangentU, coatTangentV;
  uint32_t texMask;              // bit per TEX_* slot whose value below replaces the material constant at this hit
  V3 texBaseColor, texEmission, texTransColor; float texRoughness, texMetallic, texTransWeight;
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

// PACKED: the three corners come from the mesh triangle's TriShade record (normals
// / tangents decoded on the host, as in FVertex) instead of three FVertex records.
template <bool PACKED = false>
__device__ __forceinline__ void setup_shading_state(const SceneView& sc, uint32_t triIdx, float hu, float hv, V3 rayDir, ShState& s)
{
  // one dependent step: the triangle record's tail names the instance, the material and the three vertices (or the shading record)
  const uint4* tp = reinterpret_cast<const uint4*>(sc.tris) + (size_t)triIdx * 4u;
  const uint4 tc = tp[2], td = tp[3]; // (e2.z, origId, instance, matFlags), (i0 | shading record, i1, i2, prim)
  s.material = tc.w & 0x00ffffffu; s.meshFlags = tc.w >> 30;
  const float4* ip = reinterpret_cast<const float4*>(&sc.instances[tc.z]);
  const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2], r3 = ip[3], r4 = ip[4], r5 = ip[5];
  V3 pa, pb, pc, n0, n1, n2, t0, t1, t2; float sa, sb, scg, ua, ub, uc, va_, vb_, vc_;
  if (PACKED) {
    // 10 x 16 bytes (two lines): positions, decoded normals and tangents, uv, signs, indices
    const uint4* q = reinterpret_cast<const uint4*>(&sc.triShade[td.x]);
    const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6], q7 = q[7], q8 = q[8], q9 = q[9];
    pa = v3(u2f(q0.x), u2f(q0.y), u2f(q0.z)); pb = v3(u2f(q0.w), u2f(q1.x), u2f(q1.y)); pc = v3(u2f(q1.z), u2f(q1.w), u2f(q2.x));
    // decoded on the host (gi_build.cpp)
    n0 = v3(u2f(q2.y), u2f(q2.z), u2f(q2.w)); n1 = v3(u2f(q3.x), u2f(q3.y), u2f(q3.z)); n2 = v3(u2f(q3.w), u2f(q4.x), u2f(q4.y));
    t0 = v3(u2f(q4.z), u2f(q4.w), u2f(q5.x)); t1 = v3(u2f(q5.y), u2f(q5.z), u2f(q5.w)); t2 = v3(u2f(q6.x), u2f(q6.y), u2f(q6.z));
    ua = u2f(q6.w); va_ = u2f(q7.x); ub = u2f(q7.y); vb_ = u2f(q7.z); uc = u2f(q7.w); vc_ = u2f(q8.x);
    sa = u2f(q8.y); sb = u2f(q8.z); scg = u2f(q8.w);
    s.vi[0] = q9.x; s.vi[1] = q9.y; s.vi[2] = q9.z;
  } else {
    const float4* va = reinterpret_cast<const float4*>(&sc.verts[td.x]);
    const float4* vb = reinterpret_cast<const float4*>(&sc.verts[td.y]);
    const float4* vc = reinterpret_cast<const float4*>(&sc.verts[td.z]);
    const float4 a1 = va[0], a2 = va[1], a3 = va[2];
    const float4 b1 = vb[0], b2 = vb[1], b3 = vb[2];
    const float4 c1 = vc[0], c2 = vc[1], c3 = vc[2];
    pa = v3(a1.x, a1.y, a1.z); pb = v3(b1.x, b1.y, b1.z); pc = v3(c1.x, c1.y, c1.z);
    n0 = v3(a2.x, a2.y, a2.z); n1 = v3(b2.x, b2.y, b2.z); n2 = v3(c2.x, c2.y, c2.z); // decoded on the host (:31-33)
    t0 = v3(a3.x, a3.y, a3.z); t1 = v3(b3.x, b3.y, b3.z); t2 = v3(c3.x, c3.y, c3.z); // decoded on the host (:48-50)
    sa = a1.w; sb = b1.w; scg = c1.w; ua = a2.w; ub = b2.w; uc = c2.w; va_ = a3.w; vb_ = b3.w; vc_ = c3.w;
    s.vi[0] = td.x; s.vi[1] = td.y; s.vi[2] = td.z;
  }
  const float o2w[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
  const float w2o[9] = {r3.x, r3.y, r3.z, r3.w, r4.x, r4.y, r4.z, r4.w, r5.x};
  const float bx = 1.0f - hu - hv, by = hu, bz = hv;                                  // :17
  const V3 localPos = (pa * bx + pb * by) + pc * bz;                                  // :24
  s.position = xform_point(o2w, localPos, 1.0f);                                      // :25
  V3 gn;
  if (PACKED) {
    gn = normalize(cross(pb - pa, pc - pa));                                          // :27
    gn = normalize(xform_normal(w2o, gn));                                            // :28
  // the same two lines, evaluated once per flattened triangle on the host (gi_build.cpp)
  } else { const F4 g = ld4(&sc.triGeomNormal[triIdx]); gn = v3(g.x, g.y, g.z); }
  const V3 ln = normalize((n0 * bx + n1 * by) + n2 * bz);                             // :35
  V3 nrm = normalize(xform_normal(w2o, ln));                                          // :36
  s.frontFace = dot(gn, -rayDir) >= 0.0f;                                             // :39
  if (!s.frontFace) { gn = -gn; nrm = -nrm; }                                         // :41-45
  const V3 lt = normalize((t0 * bx + t1 * by) + t2 * bz);                             // :52
  V3 tg = normalize(xform_point(o2w, lt, 0.0f));                                      // :53
  tg = normalize(tg - nrm * dot(tg, nrm));                                            // :56
  const float bs = (bx * sa + by * sb) + bz * scg;                                    // :58
  s.tangentU = tg; s.tangentV = cross(nrm, tg) * bs;                                  // :59
  s.u = (bx * ua + by * ub) + bz * uc; s.v = (bx * va_ + by * vb_) + bz * vc_;        // :62-65
  s.normal = nrm; s.geomNormal = gn;
  s.mesh = f2u(r5.y); s.instanceId = (int32_t)f2u(r5.z); s.prim = td.w; s.hu = hu; s.hv = hv;
  s.ior1 = 0.0f; s.ior2 = 0.0f;
  s.thinWalled = false; s.sssVolume = false; s.hasCoatFrame = false; s.texMask = 0u; s.texBaseColor = v3(0.0f, 0.0f, 0.0f); s.texEmission = s.texBaseColor;
      s.texRoughness = 0.0f; s.texMetallic = 0.0f; s.texTransColor = s.texBaseColor; s.texTransWeight = 0.0f;
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
  { // geometry_coat_normal: a tangent-space map in the surface's own (unmapped) frame -> the coat lobe's frame; same treatment as the base normal map below
    const TexBindingRec& b = m->tex[TEX_COAT_NORMAL];
    if (m->klass == 2u && b.tex != 0u) {
      float tu = st.u, tv = st.v; tex_transform_st(b, tu, tv);
      const F4 t = tex_lookup_float4_2d(sc.textures[b.tex - 1u], tu, tv, b.mode & 0xffu, (b.mode >> 8) & 0xffu);
      const float val[3] = {t.x * b.scale[0] + b.bias[0], t.y * b.scale[1] + b.bias[1], t.z * b.scale[2] + b.bias[2]};
      V3 n = normalize((st.tangentU * val[0] + st.tangentV * val[1]) + st.normal * val[2]);
      n = adapt_normal(rayDir, st.geomNormal, n);
      const float hs = dot(cross(st.normal, st.tangentU), st.tangentV) >= 0.0f ? 1.0f : -1.0f;
      const V3 tg = normalize(st.tangentU - n * dot(st.tangentU, n));
      st.hasCoatFrame = true; st.coatNormal = n; st.coatTangentU = tg; st.coatTangentV = cross(n, tg) * hs;
    }
  }
#pragma unroll
  for (uint32_t k = 0; k < SHADE_SLOT_COUNT; k++) { // TEX_OPACITY belongs to the any-hit test (cutout_opacity_at), TEX_COAT_NORMAL was resolved above
    const uint32_t slot = shade_slot(k);
    const TexBindingRec& b = m->tex[slot];
    if (b.tex == 0u) {
      if (!(b.mode & TEX_MODE_PRIMVAR) || slot == TEX_NORMAL) continue;
      const bool vec = slot == TEX_BASE_COLOR || slot == TEX_EMISSION || slot == TEX_TRANSMISSION_COLOR;
      // the two magic scene-data names (mdl_interface.glsl:329-334 float3 only, :390-395 float only)
      if (vec && (b.mode & TEX_MODE_CAMERA_POSITION)) {
        st.texMask |= 1u << slot;
        if (slot == TEX_BASE_COLOR) st.texBaseColor = v3(sc.cameraPosition); else if (slot == TEX_EMISSION) st.texEmission = v3(sc.cameraPosition);
            else st.texTransColor = v3(sc.cameraPosition);
        continue;
      }
      if (!vec && (b.mode & TEX_MODE_FRAME)) {
        st.texMask |= 1u << slot;
        if (slot == TEX_ROUGHNESS) st.texRoughness = sc.frame; else if (slot == TEX_METALLIC) st.texMetallic = sc.frame; else st.texTransWeight = sc.frame;
        continue;
      }
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
      float o[3] = {0.0f, 0.0f, 0.0f};
      if (info & SD_INFO_INT) { // scene_data_lookup_int / _int3 (:426-476): the value of the nearest vertex, per component, converted to float
        const uint32_t pick = bx > by ? (bx > bz ? i0 : i2) : (by > bz ? i1 : i2);
#pragma unroll
        for (uint32_t c = 0; c < 3u; c++) if (c == 0u || vec) o[c] = (float)(int32_t)f2u(d[pick * stride + (c < stride ? c : stride - 1u)]);
      } else {
#pragma unroll
      for (uint32_t c = 0; c < 3u; c++) if (c == 0u || vec) o[c] = (d[i0 * stride + c] * bx + d[i1 * stride + c] * by) + d[i2 * stride + c] * bz;
      }
      st.texMask |= 1u << slot;
      if (slot == TEX_BASE_COLOR) st.texBaseColor = v3(o[0], o[1], o[2]);
      else if (slot == TEX_EMISSION) st.texEmission = v3(o[0], o[1], o[2]);
      else if (slot == TEX_TRANSMISSION_COLOR) st.texTransColor = v3(o[0], o[1], o[2]);
      else if (slot == TEX_ROUGHNESS) st.texRoughness = o[0];
      else if (slot == TEX_METALLIC) st.texMetallic = o[0];
      else st.texTransWeight = o[0];
      continue;
    }
    float tu = st.u, tv = st.v; tex_transform_st(b, tu, tv);
    const F4 t = tex_lookup_float4_2d(sc.textures[b.tex - 1u], tu, tv, b.mode & 0xffu, (b.mode >> 8) & 0xffu);
    const float val[4] = {t.x * b.scale[0] + b.bias[0], t.y * b.scale[1] + b.bias[1], t.z * b.scale[2] + b.bias[2], t.w * b.scale[3] + b.bias[3]};
    const uint32_t ch = (b.mode >> 16) & 3u;
    const float sel = ch == 0u ? val[0] : (ch == 1u ? val[1] : (ch == 2u ? val[2] : val[3]));
    st.texMask |= 1u << slot;
    if (slot == TEX_BASE_COLOR) st.texBaseColor = v3(val[0], val[1], val[2]);
    else if (slot == TEX_EMISSION) st.texEmission = v3(val[0], val[1], val[2]);
    else if (slot == TEX_TRANSMISSION_COLOR) st.texTransColor = v3(val[0], val[1], val[2]);
    else if (slot == TEX_ROUGHNESS) st.texRoughness = sel;
    else if (slot == TEX_METALLIC) st.texMetallic = sel;
    else if (slot == TEX_TRANSMISSION_WEIGHT) st.texTransWeight = sel;
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
__device__ inline void dome_miss(const SceneView& sc, const PathState& st, uint32_t slot, V3 rayDir)
{
  Slot* S = &st.slots[slot];
  const F4 tb = ld4(&S->thr);
  const F4 rr = ld4(&S->rad);
  const bool isPrimaryRay = (f2u(tb.w) & 0x00000fffu) == 0u;
  if (st.neeKey && isPrimaryRay) nee_aov_record(st, slot, false); // a primary miss samples no light: "not shadowed" (rp_main.rgen:431-435)
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
// [ours] beside EV_DIFFUSE | EV_TRANSMISSION: entered through the volumetric subsurface lobe (the medium pushed is MaterialRec::sss)
enum : uint32_t { EV_SUBSURFACE = 64 };

__device__ __forceinline__ V3 to_world(const ShState& s, V3 l) { return (s.tangentU * l.x + s.tangentV * l.y) + s.normal * l.z; }
__device__ __forceinline__ V3 to_local(const ShState& s, V3 w) { return v3(dot(w, s.tangentU), dot(w, s.tangentV), dot(w, s.normal)); }
__device__ __forceinline__ V3 to_world_coat(const ShState& s, V3 l)
{ return s.hasCoatFrame ? (s.coatTangentU * l.x + s.coatTangentV * l.y) + s.coatNormal * l.z : to_world(s, l); }
__device__ __forceinline__ V3 to_local_coat(const ShState& s, V3 w)
{ return s.hasCoatFrame ? v3(dot(w, s.coatTangentU), dot(w, s.coatTangentV), dot(w, s.coatNormal)) : to_local(s, w); }
__device__ __forceinline__ float schlick_w(float c) { float m = 1.0f - c; m = fmin2(fmax2(m, 0.0f), 1.0f); float m2 = m * m; return m2 * m2 * m; }
__device__ __forceinline__ float ggx_lambda_term(float a2, float c) { return gi_sqrt(a2 + (1.0f - a2) * c * c); }
__device__ __forceinline__ V3 schlick3(V3 F0, float c) { float w = schlick_w(c); return F0 + (v3(1.0f, 1.0f, 1.0f) - F0) * w; }

struct GgxOut { V3 l2; float pdf, g2OverG1, kh; bool valid; };
__device__ inline GgxOut ggx_sample(V3 l1, float alpha, float x0, float x1)
{
  GgxOut o; o.valid = false; o.pdf = 0.0f; o.g2OverG1 = 0.0f; o.kh = 0.0f; o.l2 = v3(0.0f, 0.0f, 0.0f);
  V3 vh = normalize(v3(alpha * l1.x, alpha * l1.y, l1.z));
  float lensq = vh.x * vh.x + vh.y * vh.y;
  V3 T1 = lensq > 0.0f ? v3(-vh.y, vh.x, 0.0f) * (1.0f / gi_sqrt(lensq)) : v3(1.0f, 0.0f, 0.0f);
  V3 T2 = cross(vh, T1);
  float r = gi_sqrt(x0);
  float s, c; gi_sincos2pi(x1, &s, &c);
  float t1 = r * c, t2 = r * s;
  float sm = 0.5f * (1.0f + vh.z);
  t2 = (1.0f - sm) * gi_sqrt(fmax2(0.0f, 1.0f - t1 * t1)) + sm * t2;
  V3 nh = (T1 * t1 + T2 * t2) + vh * gi_sqrt(fmax2(0.0f, (1.0f - t1 * t1) - t2 * t2));
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

// Anisotropic form, operation for operation the oracle's (oracle/gi_oracle.cpp "Anisotropic form"): used only when ax != ay, so isotropic materials keep the
// arithmetic above bit for bit.
__device__ __forceinline__ float ggx_lambda_xy(float ax, float ay, V3 v) { return gi_sqrt(((ax * v.x) * (ax * v.x) + (ay * v.y) * (ay * v.y)) + v.z * v.z); }
__device__ __forceinline__ float ggx_d_xy(float ax, float ay, V3 h)
{
  const float hx = h.x / ax, hy = h.y / ay;
  const float dd = (hx * hx + hy * hy) + h.z * h.z;
  return 1.0f / (((GI_PI * ax) * ay) * (dd * dd));
}
__device__ inline GgxOut ggx_sample_xy(V3 l1, float ax, float ay, float x0, float x1)
{
  GgxOut o; o.valid = false; o.pdf = 0.0f; o.g2OverG1 = 0.0f; o.kh = 0.0f; o.l2 = v3(0.0f, 0.0f, 0.0f);
  V3 vh = normalize(v3(ax * l1.x, ay * l1.y, l1.z));
  float lensq = vh.x * vh.x + vh.y * vh.y;
  V3 T1 = lensq > 0.0f ? v3(-vh.y, vh.x, 0.0f) * (1.0f / gi_sqrt(lensq)) : v3(1.0f, 0.0f, 0.0f);
  V3 T2 = cross(vh, T1);
  float r = gi_sqrt(x0);
  float s, c; gi_sincos2pi(x1, &s, &c);
  float t1 = r * c, t2 = r * s;
  float sm = 0.5f * (1.0f + vh.z);
  t2 = (1.0f - sm) * gi_sqrt(fmax2(0.0f, 1.0f - t1 * t1)) + sm * t2;
  V3 nh = (T1 * t1 + T2 * t2) + vh * gi_sqrt(fmax2(0.0f, (1.0f - t1 * t1) - t2 * t2));
  V3 h = normalize(v3(ax * nh.x, ay * nh.y, fmax2(0.0f, nh.z)));
  float kh = dot(l1, h);
  V3 l2 = h * (2.0f * kh) - l1;
  if (!(l2.z > 0.0f) || !(kh > 0.0f)) return o;
  float nk1 = l1.z, nk2 = l2.z;
  float D = ggx_d_xy(ax, ay, h);
  float L1 = ggx_lambda_xy(ax, ay, l1), L2 = ggx_lambda_xy(ax, ay, l2);
  float G1 = 2.0f * nk1 / (nk1 + L1);
  float G2 = 2.0f * nk1 * nk2 / (nk2 * L1 + nk1 * L2);
  o.l2 = l2; o.kh = kh; o.pdf = G1 * D / (4.0f * nk1); o.g2OverG1 = G2 / G1; o.valid = true;
  return o;
}
__device__ inline void ggx_eval_xy(V3 l1, V3 l2, float ax, float ay, float& fcos, float& pdf, float& kh)
{
  fcos = 0.0f; pdf = 0.0f; kh = 0.0f;
  if (!(l1.z > 0.0f) || !(l2.z > 0.0f)) return;
  V3 h = normalize(l1 + l2);
  kh = dot(l1, h);
  float nk1 = l1.z, nk2 = l2.z;
  float D = ggx_d_xy(ax, ay, h);
  float L1 = ggx_lambda_xy(ax, ay, l1), L2 = ggx_lambda_xy(ax, ay, l2);
  float G1 = 2.0f * nk1 / (nk1 + L1);
  float G2 = 2.0f * nk1 * nk2 / (nk2 * L1 + nk1 * L2);
  fcos = D * G2 / (4.0f * nk1);
  pdf = G1 * D / (4.0f * nk1);
}
// open_pbr_anisotropy (open_pbr_surface.mtlx:133-136, 552-555): alpha_t = r^2 sqrt(2 / (1 + (1 - a)^2)), alpha_b = (1 - a) alpha_t
__device__ __forceinline__ void opbr_anisotropy(float alpha, float a, float& ax, float& ay)
{
  ax = alpha; ay = alpha;
  if (!(a > 0.0f)) return;
  const float inv = 1.0f - fmin2(a, 1.0f);
  ax = fmax2(alpha * gi_sqrt(2.0f / (1.0f + inv * inv)), 0.001f); ay = fmax2(inv * ax, 0.001f);
}
__device__ __forceinline__ GgxOut ggx_sample2(V3 l1, float ax, float ay, float x0, float x1)
{ return ax == ay ? ggx_sample(l1, ax, x0, x1) : ggx_sample_xy(l1, ax, ay, x0, x1); }
__device__ __forceinline__ void ggx_eval2(V3 l1, V3 l2, float ax, float ay, float& fcos, float& pdf, float& kh)
{ if (ax == ay) ggx_eval(l1, l2, ax, fcos, pdf, kh); else ggx_eval_xy(l1, l2, ax, ay, fcos, pdf, kh); }

// ior2 / ior1 of the interface (== oracle relative_eta): eta entering, 1/eta leaving when the medium stack is empty
__device__ __forceinline__ float relative_eta(const ShState& st, float materialEta)
{
  const bool outside = st.frontFace || st.thinWalled; // rp_main.chit:188-189
  float e1 = st.ior1 == 0.0f ? (outside ? 1.0f : -1.0f) : st.ior1;
  float e2 = st.ior2 == 0.0f ? (outside ? -1.0f : 1.0f) : st.ior2;
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
  float ct = gi_sqrt(1.0f - sin2t);
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
// The OpenPBR pieces below restate oracle/gi_oracle.cpp operation for operation (same names): coat roughening
// (open_pbr_surface.mtlx:101-131), base darkening under the coat (:470-541), emission through the coat (:590-619), energy-preserving
// Oren-Nayar for base_diffuse_roughness > 0 (:200-206; Portsmouth, Kutz, Hill 2024).
__device__ __forceinline__ float opbr_effective_roughness(float r, float cr, float coat)
{
  float c4 = (cr * cr) * (cr * cr), r4 = (r * r) * (r * r);
  float ra = gi_sqrt(gi_sqrt(fmin2(1.0f, 2.0f * c4 + r4)));
  return ra * coat + r * (1.0f - coat);
}
__device__ __forceinline__ V3 opbr_base_darkening(V3 baseColor, float sw, float metalness, float coat, float coatF0, float cior, float coatDarkening)
{
  const float w = coat * coatDarkening;
  if (w == 0.0f) return v3(1.0f, 1.0f, 1.0f);
  const float K = 1.0f - (1.0f - coatF0) / (cior * cior);
  const V3 Eb = (baseColor * sw) * metalness + baseColor * (1.0f - metalness);
  const float n = 1.0f - K;
  const V3 bd = v3(n / (1.0f - Eb.x * K), n / (1.0f - Eb.y * K), n / (1.0f - Eb.z * K));
  return bd * w + v3(1.0f, 1.0f, 1.0f) * (1.0f - w);
}
__device__ __forceinline__ V3 opbr_emission_factor(float coat, V3 coatColor, float coatF0, float c)
{
  if (coat == 0.0f) return v3(1.0f, 1.0f, 1.0f);
  const float f = (1.0f - coatF0) * (1.0f - schlick_w(c));
  return (coatColor * f) * coat + v3(1.0f, 1.0f, 1.0f) * (1.0f - coat);
}
__device__ __forceinline__ float eon_albedo_fit(float mu, float r)
{
  const float mc = 1.0f - mu;
  const float G = mc * (0.0571085289f + mc * (0.491881867f + mc * (-0.332181442f + mc * 0.0714429953f)));
  return (1.0f + r * G) / (1.0f + 0.28779343f * r);
}
__device__ inline V3 eon_pi_f(V3 rho, float r, V3 l1, V3 l2)
{
  const float mi = l1.z, mo = l2.z;
  const float s = l1.x * l2.x + l1.y * l2.y;
  const float sOverT = s > 0.0f ? s / fmax2(mi, mo) : s;
  const float AF = 1.0f / (1.0f + 0.28779343f * r);
  const float ss = AF * (1.0f + r * sOverT);
  const float EFo = eon_albedo_fit(mo, r), EFi = eon_albedo_fit(mi, r);
  const float avgEF = AF * (1.0f + 0.07248828f * r);
  const float ms = (fmax2(1e-7f, 1.0f - EFo) * fmax2(1e-7f, 1.0f - EFi)) / fmax2(1e-7f, 1.0f - avgEF);
  const V3 rr = rho * rho;
  const V3 rhoMs = v3((rr.x * avgEF) / (1.0f - rho.x * (1.0f - avgEF)), (rr.y * avgEF) / (1.0f - rho.y * (1.0f - avgEF)),
      (rr.z * avgEF) / (1.0f - rho.z * (1.0f - avgEF)));
  return rho * ss + rhoMs * ms;
}
struct OpbrParams { V3 albedo, metalTint, specColor, transTint, coatTint, baseColor, ssColor, fuzzColor;
    float metalness, alpha, alphaY, coat, coatAlpha, coatAlphaY, coatF0, eta, tw, specWeight, baseWeight, diffRough, ssWeight, ssAniso, fuzzWeight, fuzzAlpha,
    filmWeight, filmNm, filmIor, coatRotC, coatRotS; bool thinWalled, ssVolume, coatRot; };
// geometry_coat_tangent (open_pbr_surface.mtlx:91, 561): the coat lobe's tangent is the frame's tangent turned by coat rotation turns towards its bitangent --
// what a document binds to the input (rotate3d of Tworld about the normal; Standard Surface's coat_rotation).  The turn is applied to the local x / y of a
// direction in the coat's frame (the surface's, or geometry_coat_normal's), with the cosine / sine the host derived (MaterialRec::sss[6..7]); == oracle.
__device__ __forceinline__ V3 coat_turn_local(const OpbrParams& o, V3 l)
{ return v3(l.x * o.coatRotC + l.y * o.coatRotS, l.y * o.coatRotC - l.x * o.coatRotS, l.z); }
__device__ __forceinline__ V3 coat_turn_world(const OpbrParams& o, V3 l)
{ return v3(l.x * o.coatRotC - l.y * o.coatRotS, l.x * o.coatRotS + l.y * o.coatRotC, l.z); }
// geometry_tangent (open_pbr_surface.mtlx:89; 385, 402, 410, 449, 457: the tangent of the dielectric and conductor lobes) in the form documents feed it -- the
// geometry tangent turned about the normal (Standard Surface's specular_rotation, glTF's anisotropy_rotation): the shading frame's tangents are turned IN PLACE
// once per hit, before the BSDF runs (no lobe beneath the coat can tell a turned frame from a turned tangent: the diffuse lobes depend on the normal only).
// Called by the shade stage and the BSDF debug kernel for OpenPBR materials with MATF_SPEC_ROTATION; == oracle opbr_enter.
__device__ __forceinline__ void spec_turn_frame(const MaterialRec* m, ShState& st)
{
  const float c = m->sss[8], s = m->sss[9];
  const V3 tu = st.tangentU * c + st.tangentV * s, tv = st.tangentV * c - st.tangentU * s;
  st.tangentU = tu; st.tangentV = tv;
}
__device__ __forceinline__ OpbrParams opbr_params(const MaterialRec* m, const ShState& st)
{
  OpbrParams o; const float* p = m->p;
  o.albedo = v3(p[MP_ALBEDO], p[MP_ALBEDO + 1], p[MP_ALBEDO + 2]);
  o.baseColor = v3(p[0], p[1], p[2]); o.baseWeight = p[17]; o.diffRough = p[27]; const uint32_t feat = (uint32_t)p[MP_FEATURES];
      o.thinWalled = (feat & MATF_THIN_WALLED) != 0u;
  o.metalTint = v3(p[MP_F0], p[MP_F0 + 1], p[MP_F0 + 2]);
  o.specColor = v3(p[7], p[8], p[9]);
  o.specWeight = p[18]; o.metalness = p[10];
  o.alpha = p[MP_ALPHA]; o.coat = p[MP_COAT]; o.coatAlpha = p[MP_COAT_ALPHA]; o.coatF0 = p[MP_COAT_F0]; o.eta = p[MP_ETA];
  o.coatTint = v3(1.0f, 1.0f, 1.0f) * (1.0f - o.coat) + v3(p[19], p[20], p[21]) * o.coat;
  o.tw = p[23];
  o.transTint = (p[28] > 0.0f) ? v3(1.0f, 1.0f, 1.0f) : v3(p[24], p[25], p[26]);
  if (st.texMask & (1u << TEX_BASE_COLOR)) { o.albedo = st.texBaseColor * p[17]; o.baseColor = st.texBaseColor; }
  if (st.texMask & (1u << TEX_ROUGHNESS)) { const float r = opbr_effective_roughness(st.texRoughness, p[13], o.coat); o.alpha = fmax2(r * r, 0.001f); }
  if (st.texMask & (1u << TEX_METALLIC)) o.metalness = st.texMetallic;
  if (st.texMask & (1u << TEX_TRANSMISSION_WEIGHT)) o.tw = st.texTransWeight;
  // (with a depth the colour is the medium's: the material's constant)
  if ((st.texMask & (1u << TEX_TRANSMISSION_COLOR)) && !(p[28] > 0.0f)) o.transTint = st.texTransColor;
  o.alphaY = o.alpha; o.coatAlphaY = o.coatAlpha;
  // specular_roughness_anisotropy / coat_roughness_anisotropy
  if (feat & MATF_ANISOTROPY) { opbr_anisotropy(o.alpha, p[60], o.alpha, o.alphaY); opbr_anisotropy(o.coatAlpha, p[61], o.coatAlpha, o.coatAlphaY); }
  o.coatRot = (feat & MATF_COAT_ROTATION) != 0u; o.coatRotC = 1.0f; o.coatRotS = 0.0f;
  if (o.coatRot) { // geometry_coat_tangent; over a turned base frame (spec_turn_frame) a coat without a frame of its own takes its turn relative to that frame
    const uint32_t at = ((feat & MATF_SPEC_ROTATION) && !st.hasCoatFrame) ? 10u : 6u;
    o.coatRotC = m->sss[at]; o.coatRotS = m->sss[at + 1u];
  }
  // coat_substrate_attenuated = base_substrate * modulated_base_darkening * coat_attenuation (open_pbr_surface.mtlx:538-552)
  o.coatTint = o.coatTint * opbr_base_darkening(o.baseColor, o.specWeight, o.metalness, o.coat, o.coatF0, p[22], p[48]);
  // thin-walled subsurface (open_pbr_surface.mtlx:140-196, 207-218); the volumetric form of non-thin-walled materials is not modelled
  o.ssWeight = o.thinWalled ? p[55] : 0.0f;
  o.ssColor = v3(p[56], p[57], p[58]); o.ssAniso = p[59];
  // volumetric subsurface (open_pbr_surface.mtlx:182-192; oracle opbr_params): live when the render keeps a medium stack; the coefficients are MaterialRec::sss
  o.ssVolume = (feat & MATF_SSS_VOLUME) != 0u && st.sssVolume;
  if (o.ssVolume) o.ssWeight = fmin2(p[55], 1.0f);
  // fuzz layer (open_pbr_surface.mtlx:569-581): sheen_bsdf(fuzz_weight, fuzz_color, fuzz_roughness) on top of the coat
  o.fuzzWeight = 0.0f; o.fuzzColor = v3(1.0f, 1.0f, 1.0f); o.fuzzAlpha = 0.5f;
  if (feat & MATF_FUZZ) { o.fuzzWeight = fmin2(fmax2(p[49], 0.0f), 1.0f); o.fuzzColor = v3(p[50], p[51], p[52]); o.fuzzAlpha = fmin2(fmax2(p[53], 0.07f), 1.0f);
      }
  // thin film (open_pbr_surface.mtlx:300-304, 404-431, 450-464): weight, thickness in micrometres, ior (slot 6: OpenPBR records do not use useSpecularWorkflow)
  o.filmWeight = 0.0f; o.filmNm = 0.0f; o.filmIor = 1.0f;
  if (feat & MATF_THIN_FILM) { o.filmWeight = fmin2(fmax2(p[62], 0.0f), 1.0f); o.filmNm = fmax2(p[63], 0.0f) * 1000.0f; o.filmIor = fmax2(p[6], 1.0f); }
  return o;
}
// ---- thin film, operation for operation the oracle's (oracle/gi_oracle.cpp "thin film": Airy summation at three wavelengths; where it enters the lobes)
__device__ inline float film_reflectance(float c, float nf, float n3, float d, float lam)
{
  const float s2 = 1.0f - c * c;
  const float s2f = s2 / (nf * nf), s23 = s2 / (n3 * n3);
  if (!(s2f < 1.0f) || !(s23 < 1.0f)) return 1.0f; // total internal reflection
  const float cf = gi_sqrt(1.0f - s2f), c3 = gi_sqrt(1.0f - s23);
  const float rs12 = (c - nf * cf) / (c + nf * cf), rp12 = (nf * c - cf) / (nf * c + cf);
  const float rs23 = (nf * cf - n3 * c3) / (nf * cf + n3 * c3), rp23 = (n3 * cf - nf * c3) / (n3 * cf + nf * c3);
  const float ph = ((2.0f * nf) * d * cf) / lam; // phase difference / (2 pi)
  float sn, cs; gi_sincos2pi(ph - floorf(ph), &sn, &cs); (void)sn;
  const float ps = rs12 * rs23, pp = rp12 * rp23;
  const float Rs = ((rs12 * rs12 + rs23 * rs23) + (2.0f * ps) * cs) / ((1.0f + ps * ps) + (2.0f * ps) * cs);
  const float Rp = ((rp12 * rp12 + rp23 * rp23) + (2.0f * pp) * cs) / ((1.0f + pp * pp) + (2.0f * pp) * cs);
  return fmin2(fmax2(0.5f * (Rs + Rp), 0.0f), 1.0f);
}
__device__ inline V3 film_fresnel(float c, float nf, V3 n3, float d)
{ return v3(film_reflectance(c, nf, n3.x, d, 611.4f), film_reflectance(c, nf, n3.y, d, 548.4f), film_reflectance(c, nf, n3.z, d, 464.3f)); }
__device__ inline V3 opbr_film_dielectric(const OpbrParams& o, float c, float eta, float Fplain)
{
  const float nf = (eta < 1.0f) ? o.filmIor * eta : o.filmIor;
  return v3(Fplain, Fplain, Fplain) * (1.0f - o.filmWeight) + film_fresnel(c, nf, v3(eta, eta, eta), o.filmNm) * o.filmWeight;
}
__device__ inline V3 opbr_film_metal(const OpbrParams& o, float c, V3 Fplain)
{
  const V3 f0 = v3(fmin2(fmax2(o.albedo.x, 0.0f), 0.98f), fmin2(fmax2(o.albedo.y, 0.0f), 0.98f), fmin2(fmax2(o.albedo.z, 0.0f), 0.98f));
  const V3 r = v3(gi_sqrt(f0.x), gi_sqrt(f0.y), gi_sqrt(f0.z));
  const V3 n3 = v3((1.0f + r.x) / (1.0f - r.x), (1.0f + r.y) / (1.0f - r.y), (1.0f + r.z) / (1.0f - r.z));
  return Fplain * (1.0f - o.filmWeight) + film_fresnel(c, o.filmIor, n3, o.filmNm) * o.filmWeight;
}
// ---- fuzz (sheen) lobe, operation for operation the oracle's (oracle/gi_oracle.cpp "fuzz (sheen) lobe": the model, its sources and the layering rule are
// described there): "Charlie" distribution x Ashikhmin / Neubelt visibility, directional albedo from the table of tools/gen_fuzz_albedo.py + 0.01.
__device__ const float FUZZ_ALBEDO[16][17] = {
// rows: alpha = 1/16 .. 1, columns: mu = 0 .. 1 in steps of 1/16 (tools/gen_fuzz_albedo.py)
  {1.66603482e+00f, 1.05229735e+00f, 7.48142481e-01f, 5.42552471e-01f, 3.94544691e-01f, 2.85340458e-01f, 2.04062358e-01f, 1.43585801e-01f, 9.88851935e-02f,
      6.62436262e-02f, 4.28260490e-02f, 2.64271554e-02f, 1.53114153e-02f, 8.10563751e-03f, 3.72325582e-03f, 1.30873080e-03f, 1.95315428e-04f},
  {1.22866476e+00f, 8.68625820e-01f, 6.77242339e-01f, 5.38716376e-01f, 4.31280196e-01f, 3.45276922e-01f, 2.75279015e-01f, 2.17816725e-01f, 1.70480222e-01f,
      1.31495669e-01f, 9.95000154e-02f, 7.34108016e-02f, 5.23458906e-02f, 3.55714411e-02f, 2.24667117e-02f, 1.24994880e-02f, 5.20835957e-03f},
  {1.04291248e+00f, 7.75057077e-01f, 6.29255474e-01f, 5.20992339e-01f, 4.34587359e-01f, 3.63185972e-01f, 3.03000093e-01f, 2.51652122e-01f, 2.07522362e-01f,
      1.69441670e-01f, 1.36528745e-01f, 1.08096354e-01f, 8.35938677e-02f, 6.25700131e-02f, 4.46479134e-02f, 2.95076892e-02f, 1.68739911e-02f},
  {9.36414123e-01f, 7.16623902e-01f, 5.95793307e-01f, 5.04945695e-01f, 4.31368500e-01f, 3.69544923e-01f, 3.16451848e-01f, 2.70210087e-01f, 2.29553193e-01f,
      1.93577752e-01f, 1.61611512e-01f, 1.33137539e-01f, 1.07747734e-01f, 8.51129442e-02f, 6.49628416e-02f, 4.70720194e-02f, 3.12500633e-02f},
  {8.66332769e-01f, 6.76128209e-01f, 5.71131229e-01f, 4.91654038e-01f, 4.26739037e-01f, 3.71650100e-01f, 3.23803574e-01f, 2.81601369e-01f, 2.43972436e-01f,
      2.10157394e-01f, 1.79594919e-01f, 1.51856542e-01f, 1.26606807e-01f, 1.03577562e-01f, 8.25508535e-02f, 6.33470416e-02f, 4.58163172e-02f},
  {8.16336453e-01f, 6.46200657e-01f, 5.52162349e-01f, 4.80711669e-01f, 4.22050655e-01f, 3.71954530e-01f, 3.28124613e-01f, 2.89142847e-01f, 2.54061490e-01f,
      2.22210526e-01f, 1.93096176e-01f, 1.66342735e-01f, 1.41657025e-01f, 1.18805535e-01f, 9.75991860e-02f, 7.78828189e-02f, 5.95276207e-02f},
  {7.78707266e-01f, 6.23087585e-01f, 5.37092745e-01f, 4.71618980e-01f, 4.17691469e-01f, 3.71446818e-01f, 3.30786139e-01f, 2.94416696e-01f, 2.61475623e-01f,
      2.31353760e-01f, 2.03602642e-01f, 1.77881300e-01f, 1.53923839e-01f, 1.31518483e-01f, 1.10493734e-01f, 9.07087103e-02f, 7.20463097e-02f},
  {7.49280632e-01f, 6.04652286e-01f, 5.24816334e-01f, 4.63969946e-01f, 4.13753271e-01f, 3.70571792e-01f, 3.32474768e-01f, 2.98261851e-01f, 2.67132372e-01f,
      2.38521293e-01f, 2.12012753e-01f, 1.87290415e-01f, 1.64107352e-01f, 1.42266646e-01f, 1.21608362e-01f, 1.02000684e-01f, 8.33334997e-02f},
  {7.25595057e-01f, 5.89579582e-01f, 5.14612794e-01f, 4.57457095e-01f, 4.10229623e-01f, 3.69543940e-01f, 3.33563626e-01f, 3.01159322e-01f, 2.71578044e-01f,
      2.44288355e-01f, 2.18898997e-01f, 1.95112124e-01f, 1.72694832e-01f, 1.51460946e-01f, 1.31258771e-01f, 1.11962639e-01f, 9.34669897e-02f},
  {7.06095338e-01f, 5.77011704e-01f, 5.05992115e-01f, 4.51849908e-01f, 4.07083064e-01f, 3.68470997e-01f, 3.34268302e-01f, 3.03401917e-01f, 2.75156647e-01f,
      2.49027595e-01f, 2.24642813e-01f, 2.01718882e-01f, 1.80033773e-01f, 1.59409523e-01f, 1.39700606e-01f, 1.20785803e-01f, 1.02562815e-01f},
  {6.89747393e-01f, 5.66363335e-01f, 4.98608768e-01f, 4.46974337e-01f, 4.04269099e-01f, 3.67407918e-01f, 3.34719449e-01f, 3.05176616e-01f, 2.78094798e-01f,
      2.52990693e-01f, 2.29507983e-01f, 2.07374871e-01f, 1.86378047e-01f, 1.66346103e-01f, 1.47138327e-01f, 1.28636986e-01f, 1.10742249e-01f},
  {6.75835013e-01f, 5.57220221e-01f, 4.92211729e-01f, 4.42697257e-01f, 4.01744992e-01f, 3.66382241e-01f, 3.34999412e-01f, 3.06607515e-01f, 2.80547380e-01f,
      2.56353587e-01f, 2.33682737e-01f, 2.12272644e-01f, 1.91917211e-01f, 1.72450408e-01f, 1.53735459e-01f, 1.35657445e-01f, 1.18118644e-01f},
  {6.63845837e-01f, 5.49280763e-01f, 4.86614108e-01f, 4.38915670e-01f, 3.99472445e-01f, 3.65406990e-01f, 3.35161716e-01f, 3.07779849e-01f, 2.82623708e-01f,
      2.59242892e-01f, 2.37304971e-01f, 2.16555879e-01f, 1.96795553e-01f, 1.77862525e-01f, 1.59623355e-01f, 1.41965449e-01f, 1.24793008e-01f},
  {6.53402805e-01f, 5.42319357e-01f, 4.81673747e-01f, 4.35548574e-01f, 3.97418410e-01f, 3.64487261e-01f, 3.35242122e-01f, 3.08753759e-01f, 2.84402877e-01f,
      2.61752009e-01f, 2.40478083e-01f, 2.20333964e-01f, 2.01124832e-01f, 1.82693094e-01f, 1.64908230e-01f, 1.47659644e-01f, 1.30853310e-01f},
  {6.44222379e-01f, 5.36164165e-01f, 4.77280527e-01f, 4.32531655e-01f, 3.95554543e-01f, 3.63623768e-01f, 3.35264921e-01f, 3.09572637e-01f, 2.85943538e-01f,
      2.63951302e-01f, 2.43281066e-01f, 2.23691702e-01f, 2.04992920e-01f, 1.87030569e-01f, 1.69676632e-01f, 1.52822360e-01f, 1.36375397e-01f},
  {6.36086643e-01f, 5.30681610e-01f, 4.73347783e-01f, 4.29813147e-01f, 3.93856794e-01f, 3.62814993e-01f, 3.35247070e-01f, 3.10268551e-01f, 2.87289977e-01f,
      2.65894860e-01f, 2.45775416e-01f, 2.26695850e-01f, 2.08469898e-01f, 1.90946430e-01f, 1.73999682e-01f, 1.57522544e-01f, 1.41424328e-01f},
};
__device__ __forceinline__ float fuzz_albedo(float mu, float alpha)
{
  const float x = fmin2(fmax2(mu, 0.0f), 1.0f) * 16.0f;
  int i = (int)x; if (i > 15) i = 15;
  const float fx = x - (float)i;
  const float y = alpha * 16.0f - 1.0f;
  int j = (int)y; if (j > 14) j = 14; if (j < 0) j = 0;
  const float fy = y - (float)j;
  const float a = FUZZ_ALBEDO[j][i] * (1.0f - fx) + FUZZ_ALBEDO[j][i + 1] * fx;
  const float b = FUZZ_ALBEDO[j + 1][i] * (1.0f - fx) + FUZZ_ALBEDO[j + 1][i + 1] * fx;
  return (a * (1.0f - fy) + b * fy) + 0.01f;
}
__device__ __forceinline__ float fuzz_dv(V3 l1, V3 l2, float alpha) // D * V for local directions with l1.z, l2.z > 0
{
  const V3 h = normalize(l1 + l2);
  const float s2 = 1.0f - h.z * h.z;
  if (!(s2 > 0.0f)) return 0.0f;
  const float inv = 1.0f / alpha;
  const float D = ((2.0f + inv) * gi_expf((0.5f * inv) * gi_logf(s2))) / (2.0f * GI_PI);
  return D / (4.0f * ((l2.z + l1.z) - l2.z * l1.z));
}
// the colour factors of subsurface_thin_walled's two lobes (the mix weight 1/2 is the lobe-selection probability and cancels): see the oracle's opbr_ss_*
__device__ __forceinline__ V3 opbr_ss_color(const OpbrParams& o) { return v3(fmax2(o.ssColor.x, 0.0f), fmax2(o.ssColor.y, 0.0f), fmax2(o.ssColor.z, 0.0f)); }
__device__ __forceinline__ V3 opbr_ss_reflect(const OpbrParams& o, V3 l1, V3 l2)
{
  const V3 c = opbr_ss_color(o);
  const V3 rho = (o.diffRough > 0.0f) ? eon_pi_f(c, o.diffRough, l1, l2) : c;
  return rho * (o.ssColor * (1.0f - o.ssAniso));
}
__device__ __forceinline__ V3 opbr_ss_transmit(const OpbrParams& o) { return opbr_ss_color(o) * (o.ssColor * (1.0f + o.ssAniso)); }

// The lobe is chosen first (cheap, divergent), then ONE micro-facet sample serves whichever glossy lobe a lane took -- coat, metal,
// dielectric reflection, transmission differ in the roughness they pass and in their weights, not in the sampling arithmetic -- so a wave
// whose lanes took different lobes runs ggx_sample once instead of once per lobe.  Same operations per lane as the oracle's branch per lobe.
// everything beneath the fuzz
__device__ inline void opbr_sample_base(const OpbrParams& o, const ShState& st, V3 k1, float x0, float x1, float x2, BsdfSample& out)
{
  V3 l1 = to_local(st, k1);
  float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
  float z = x2;
  // the coat lobe lives in its own frame when geometry_coat_normal is mapped; its Fresnel term -- lobe probability and what it leaves for the base -- follows
  V3 l1c = l1; float nk1c = nk1;
  if (st.hasCoatFrame) { l1c = to_local_coat(st, k1); nk1c = fmax2(l1c.z, 1e-4f); l1c.z = nk1c; }
  if (o.coatRot) l1c = coat_turn_local(o, l1c);
  float Fc = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(nk1c));
  float eta = 0.0f, Fd = 0.0f;
  uint32_t lobe = 0u; // 0 coat, 1 metal, 2 dielectric reflection, 3 transmission, 4 diffuse
  if (!(z < Fc)) {
    z = (z - Fc) / (1.0f - Fc);
    lobe = 1u;
    if (!(z < o.metalness)) {
      z = (z - o.metalness) / (1.0f - o.metalness);
      eta = relative_eta(st, o.eta);
      Fd = fresnel_dielectric(nk1, eta);
      lobe = 2u;
      if (!(z < Fd)) {
        z = (z - Fd) / (1.0f - Fd);
        lobe = (z < o.tw) ? 3u : 4u;
      }
    }
  }
  // thin film: everything beneath the dielectric interface is weighted by (1 - F_mix)
  // / (1 - F_plain) per channel (lobes 3 and 4 only: eta and Fd are set there)
  const V3 under = (o.filmWeight > 0.0f && lobe >= 3u)
      ? (v3(1.0f, 1.0f, 1.0f) - opbr_film_dielectric(o, nk1, eta, Fd)) * (1.0f / (1.0f - Fd)) : v3(1.0f, 1.0f, 1.0f);
  if (lobe == 4u) {
    const float pBase = (1.0f - Fc) * (1.0f - o.metalness) * (1.0f - Fd) * (1.0f - o.tw);
    V3 l = gi_sample_hemisphere(x0, x1); // cosine-weighted, for every lobe of the opaque base
    if (o.ssWeight > 0.0f) { // thin-walled subsurface takes subsurface_weight of the opaque base, half of it reflected, half transmitted
      z = (z - o.tw) / (1.0f - o.tw);
      if (z < o.ssWeight) {
        const bool through = !((z / o.ssWeight) < 0.5f);
        if (!(l.z > 0.0f)) return;
        if (o.ssVolume) { // volumetric form: the whole subsurface share enters (or, met from inside, leaves) by a cosine lobe on the far side, untinted
          V3 k2 = to_world(st, v3(l.x, l.y, -l.z));
          if (!(dot(k2, st.geomNormal) < 0.0f)) return;
          out.k2 = k2; out.pdf = pBase * o.ssWeight * (l.z / GI_PI);
          out.overPdf = o.coatTint; out.event = EV_DIFFUSE | EV_TRANSMISSION | EV_SUBSURFACE;
          if (o.filmWeight > 0.0f) out.overPdf = out.overPdf * under;
          return;
        }
        if (through) { // translucent_bsdf: Lambert on the far side
          V3 k2 = to_world(st, v3(l.x, l.y, -l.z));
          if (!(dot(k2, st.geomNormal) < 0.0f)) return;
          out.k2 = k2; out.pdf = pBase * o.ssWeight * 0.5f * (l.z / GI_PI);
          out.overPdf = opbr_ss_transmit(o) * o.coatTint; out.event = EV_DIFFUSE | EV_TRANSMISSION;
          if (o.filmWeight > 0.0f) out.overPdf = out.overPdf * under;
          return;
        }
        V3 k2 = to_world(st, l);
        if (!(dot(k2, st.geomNormal) > 0.0f)) return;
        out.k2 = k2; out.pdf = pBase * o.ssWeight * 0.5f * (l.z / GI_PI);
        out.overPdf = opbr_ss_reflect(o, l1, l) * o.coatTint; out.event = EV_DIFFUSE | EV_REFLECTION;
        if (o.filmWeight > 0.0f) out.overPdf = out.overPdf * under;
        return;
      }
    }
    V3 k2 = to_world(st, l);
    if (!(l.z > 0.0f) || !(dot(k2, st.geomNormal) > 0.0f)) return;
    out.k2 = k2; out.pdf = pBase * (1.0f - o.ssWeight) * (l.z / GI_PI);
    V3 rho = (o.diffRough > 0.0f) ? eon_pi_f(o.baseColor, o.diffRough, l1, l) * o.baseWeight : o.albedo;
    out.overPdf = rho * o.coatTint; out.event = EV_DIFFUSE | EV_REFLECTION;
    if (o.filmWeight > 0.0f) out.overPdf = out.overPdf * under;
    return;
  }
  const GgxOut g = ggx_sample2(lobe == 0u ? l1c : l1, lobe == 0u ? o.coatAlpha : o.alpha, lobe == 0u ? o.coatAlphaY : o.alphaY, x0, x1);
  if (lobe == 3u) {
    V3 h = normalize(l1 + g.l2);
    float kh = dot(l1, h);
    if (!g.valid || !(kh > 0.0f)) return;
    float Fh = fresnel_dielectric(kh, eta);
    float sin2t = (1.0f - kh * kh) / (eta * eta);
    if (!(sin2t < 1.0f)) return;
    float ct = gi_sqrt(1.0f - sin2t);
    V3 lt = h * (kh / eta - ct) - l1 * (1.0f / eta);
    if (o.thinWalled) lt = v3(g.l2.x, g.l2.y, -g.l2.z); // thin-walled: no refraction, the micro-facet reflection mirrored through the surface
    V3 k2 = to_world(st, lt);
    if (!(lt.z < 0.0f) || !(dot(k2, st.geomNormal) < 0.0f)) return;
    float a2 = o.alpha * o.alpha, nk2 = -lt.z;
    float L1 = ggx_lambda_term(a2, nk1), L2 = ggx_lambda_term(a2, nk2);
    if (o.alpha != o.alphaY) { L1 = ggx_lambda_xy(o.alpha, o.alphaY, l1); L2 = ggx_lambda_xy(o.alpha, o.alphaY, lt); }
    float G1 = 2.0f * nk1 / (nk1 + L1), G2 = 2.0f * nk1 * nk2 / (nk2 * L1 + nk1 * L2);
    float w = ((1.0f - Fh) / (1.0f - Fd)) * (G2 / G1);
    out.k2 = normalize(k2); out.pdf = (1.0f - Fc) * (1.0f - o.metalness) * (1.0f - Fd) * o.tw * g.pdf;
    out.overPdf = (o.transTint * o.coatTint) * w; out.event = EV_GLOSSY | EV_TRANSMISSION;
    if (o.filmWeight > 0.0f) out.overPdf = (o.transTint * o.coatTint) * ((v3(1.0f, 1.0f, 1.0f) - opbr_film_dielectric(o, kh, eta,
        Fh)) * ((G2 / G1) / (1.0f - Fd)));
    return;
  }
  V3 k2 = (lobe == 0u) ? to_world_coat(st, o.coatRot ? coat_turn_world(o, g.l2) : g.l2) : to_world(st, g.l2);
  if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
  out.k2 = k2; out.event = EV_GLOSSY | EV_REFLECTION;
  if (lobe == 0u) {
    float Fh = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(g.kh));
    float w = (Fh / Fc) * g.g2OverG1;
    out.pdf = Fc * g.pdf; out.overPdf = v3(w, w, w);
  } else if (lobe == 1u) {
    V3 F = schlick_f82(o.albedo, o.metalTint, g.kh) * o.specWeight;
    if (o.filmWeight > 0.0f) F = opbr_film_metal(o, g.kh, schlick_f82(o.albedo, o.metalTint, g.kh)) * o.specWeight;
    out.pdf = (1.0f - Fc) * o.metalness * g.pdf; out.overPdf = (F * o.coatTint) * g.g2OverG1;
  } else {
    float Fh = fresnel_dielectric(g.kh, eta);
    out.pdf = (1.0f - Fc) * (1.0f - o.metalness) * Fd * g.pdf;
    out.overPdf = (o.specColor * o.coatTint) * ((Fh / Fd) * g.g2OverG1);
    if (o.filmWeight > 0.0f) out.overPdf = (o.specColor * o.coatTint) * (opbr_film_dielectric(o, g.kh, eta, Fh) * (g.g2OverG1 / Fd));
  }
}

__device__ inline void opbr_sample(const MaterialRec* m, const ShState& st, V3 k1, float x0, float x1, float x2, BsdfSample& out)
{
  const OpbrParams o = opbr_params(m, st);
  if (!(o.fuzzWeight > 0.0f)) { opbr_sample_base(o, st, k1, x0, x1, x2, out); return; }
  V3 l1 = to_local(st, k1);
  const float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
  const float Ef = fuzz_albedo(nk1, o.fuzzAlpha), Pf = o.fuzzWeight * fmin2(Ef, 1.0f);
  if (x2 < Pf) { // the fuzz lobe, cosine-sampled
    const V3 l = gi_sample_hemisphere(x0, x1);
    const V3 k2 = to_world(st, l);
    if (!(l.z > 0.0f) || !(dot(k2, st.geomNormal) > 0.0f)) return;
    out.k2 = k2; out.pdf = Pf * (l.z / GI_PI);
    out.overPdf = o.fuzzColor * ((fuzz_dv(l1, l, o.fuzzAlpha) * GI_PI) / Ef);
    out.event = EV_GLOSSY | EV_REFLECTION;
    return;
  }
  opbr_sample_base(o, st, k1, x0, x1, (x2 - Pf) / (1.0f - Pf), out);
  out.pdf = out.pdf * (1.0f - Pf); // bsdf / pdf is unchanged: the layers beneath are weighted by the same 1 - P they are chosen with
}

__device__ inline void opbr_evaluate_base(const OpbrParams& o, const ShState& st, V3 k1, V3 k2, BsdfEval& out)
{
  V3 l1 = to_local(st, k1), l2 = to_local(st, k2);
  float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
  float eta = relative_eta(st, o.eta);
  V3 l1c = l1, l2c = l2; float nk1c = nk1; // the coat lobe's own frame (geometry_coat_normal)
  if (st.hasCoatFrame) { l1c = to_local_coat(st, k1); nk1c = fmax2(l1c.z, 1e-4f); l1c.z = nk1c; l2c = to_local_coat(st, k2); }
  if (o.coatRot) { l1c = coat_turn_local(o, l1c); l2c = coat_turn_local(o, l2c); }
  float Fc = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(nk1c));
  float Fd = fresnel_dielectric(nk1, eta);
  // Lobes whose weight is exactly zero are not evaluated.  The oracle evaluates them and multiplies by the zero -- the same bits: every skipped factor is
  // finite and non-negative (D, G, Fresnel terms of positive roughness and nk1 >= 1e-4), so its product with the zero weight is +0, and +0 added to a
  // non-negative sum changes nothing.  (C3's material has neither coat nor metal: 2 of its 3 GGX evaluations per shadow-ray set-up.)
  float fc = 0.0f, pc = 0.0f, khc = 0.0f, Fch = 0.0f;
  if (o.coat != 0.0f) { ggx_eval2(l1c, l2c, o.coatAlpha, o.coatAlphaY, fc, pc, khc); Fch = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(khc)); }
  float fs, ps, khs; ggx_eval2(l1, l2, o.alpha, o.alphaY, fs, ps, khs);
  float Fdh = fresnel_dielectric(khs, eta);
  float cd = l2.z / GI_PI;
  float base = 1.0f - Fc, diel = 1.0f - o.metalness;
  V3 gl = v3(Fch * fc, Fch * fc, Fch * fc);
  if (o.metalness != 0.0f) { const V3 Fm = schlick_f82(o.albedo, o.metalTint, khs) * o.specWeight; gl = gl + ((Fm * o.coatTint) * fs) * (base * o.metalness); }
  gl = gl + ((o.specColor * o.coatTint) * (Fdh * fs)) * (base * diel);
  V3 under = v3(1.0f, 1.0f, 1.0f);
  if (o.filmWeight > 0.0f) { // thin film: colour Fresnel factors in the two glossy lobes, (1 - F_mix) / (1 - F_plain) on what lies beneath the interface
    const V3 FmF = opbr_film_metal(o, khs, schlick_f82(o.albedo, o.metalTint, khs)) * o.specWeight;
    gl = v3(Fch * fc, Fch * fc, Fch * fc);
    gl = gl + ((FmF * o.coatTint) * fs) * (base * o.metalness);
    gl = gl + ((o.specColor * o.coatTint) * (opbr_film_dielectric(o, khs, eta, Fdh) * fs)) * (base * diel);
    // total internal reflection (Fd == 1: back face beyond the critical angle): nothing lies beneath the interface -- without the guard 0 * (1 / 0) = NaN
    under = (Fd < 1.0f) ? (v3(1.0f, 1.0f, 1.0f) - opbr_film_dielectric(o, nk1, eta, Fd)) * (1.0f / (1.0f - Fd)) : v3(0.0f, 0.0f, 0.0f);
  }
  out.glossy = gl;
  V3 rho = (o.diffRough > 0.0f && l2.z > 0.0f) ? eon_pi_f(o.baseColor, o.diffRough, l1, l2) * o.baseWeight : o.albedo;
  const float wBase = cd * base * diel * (1.0f - Fd) * (1.0f - o.tw);
  if (o.ssVolume) { // volumetric subsurface: its share of the opaque base transmits (not reached by NEE); the diffuse lobe keeps 1 - subsurface_weight
    out.diffuse = ((rho * (1.0f - o.ssWeight)) * o.coatTint) * wBase;
    if (o.filmWeight > 0.0f) out.diffuse = out.diffuse * under;
    out.pdf = Fc * pc + base * (o.metalness * ps + diel * (Fd * ps + (1.0f - Fd) * (1.0f - o.tw) * (1.0f - o.ssWeight) * cd));
    return;
  }
  if (o.ssWeight > 0.0f) { // reflection side of the thin-walled subsurface mix (the transmitted half lies below the surface: not reached by NEE)
    const V3 ss = (l2.z > 0.0f) ? opbr_ss_reflect(o, l1, l2) : v3(0.0f, 0.0f, 0.0f);
    out.diffuse = ((rho * (1.0f - o.ssWeight) + ss * (o.ssWeight * 0.5f)) * o.coatTint) * wBase;
    if (o.filmWeight > 0.0f) out.diffuse = out.diffuse * under;
    out.pdf = Fc * pc + base * (o.metalness * ps + diel * (Fd * ps + (1.0f - Fd) * (1.0f - o.tw) * ((1.0f - o.ssWeight) + o.ssWeight * 0.5f) * cd));
    return;
  }
  out.diffuse = (rho * o.coatTint) * wBase;
  if (o.filmWeight > 0.0f) out.diffuse = out.diffuse * under;
  out.pdf = Fc * pc + base * (o.metalness * ps + diel * (Fd * ps + (1.0f - Fd) * (1.0f - o.tw) * cd));
}
__device__ inline void opbr_evaluate(const MaterialRec* m, const ShState& st, V3 k1, V3 k2, BsdfEval& out)
{
  const OpbrParams o = opbr_params(m, st);
  opbr_evaluate_base(o, st, k1, k2, out);
  if (!(o.fuzzWeight > 0.0f)) return;
  V3 l1 = to_local(st, k1); const V3 l2 = to_local(st, k2);
  const float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
  const float Ef = fuzz_albedo(nk1, o.fuzzAlpha), Eb = fmin2(Ef, 1.0f), Pf = o.fuzzWeight * Eb, keep = 1.0f - Pf;
  const float cd = fmax2(l2.z, 0.0f) / GI_PI;
  const V3 sheen = (l2.z > 0.0f) ? o.fuzzColor * (((o.fuzzWeight * (Eb / Ef)) * fuzz_dv(l1, l2, o.fuzzAlpha)) * l2.z) : v3(0.0f, 0.0f, 0.0f);
  out.glossy = out.glossy * keep + sheen;
  out.diffuse = out.diffuse * keep;
  out.pdf = Pf * cd + keep * out.pdf;
}

// ------------------------------------------------------------------------------------------------
// OpenPBR, BASE variant (shade class SHADE_CLASS_OPBR_BASE; the reference compiles its hit shaders per material with feature #defines,
// GlslShaderGen.cpp:204-274 -- here the host sorts OpenPBR materials into two variants and k_route bins their hits apart).  A BASE material has no coat, no
// fuzz, no thin film, no anisotropy, no transmission, no subsurface, is not thin-walled and binds no texture (gi_build.cpp shadeClassOf): what is left is the
// metal lobe, the dielectric reflection and the (energy-preserving Oren-Nayar) diffuse base.  The functions below are opbr_sample / opbr_evaluate with those
// weights set to their constants and every operation that is an exact identity under them removed -- x * 1, x / 1, x - 0, (1 - 0), a lobe the selection can
// never reach; additions of an exact +0 stay where the sign of a zero could differ -- so a BASE material shades to the same bits through either variant
// (test_shade_variants_are_bit_identical runs every scene through both).
// ------------------------------------------------------------------------------------------------
struct OpbrBaseParams { V3 albedo, metalTint, specColor, baseColor; float metalness, alpha, eta, specWeight, baseWeight, diffRough; };
__device__ __forceinline__ OpbrBaseParams opbr_base_params(const MaterialRec* m)
{
  OpbrBaseParams o; const float* p = m->p;
  o.albedo = v3(p[MP_ALBEDO], p[MP_ALBEDO + 1], p[MP_ALBEDO + 2]);
  o.baseColor = v3(p[0], p[1], p[2]); o.baseWeight = p[17]; o.diffRough = p[27];
  o.metalTint = v3(p[MP_F0], p[MP_F0 + 1], p[MP_F0 + 2]);
  o.specColor = v3(p[7], p[8], p[9]);
  o.specWeight = p[18]; o.metalness = p[10];
  o.alpha = p[MP_ALPHA]; o.eta = p[MP_ETA];
  return o;
}
// What the sampling routine and the evaluation of ONE hit both start from: the view direction in the shading frame, the interface's relative ior and its
// Fresnel term at the view direction.  With NEE both run for (nearly) every hit: shade_segment makes the context once and hands it to both -- the same function
// of the same arguments, so the same bits -- instead of evaluating the 65-instruction Fresnel term (three divisions, a square root) twice.
struct OpbrBaseCtx { V3 l1; float nk1, eta, Fd; bool have; };
__device__ __forceinline__ OpbrBaseCtx opbr_base_ctx(const MaterialRec* m, const ShState& st, V3 k1)
{
  OpbrBaseCtx c; c.l1 = to_local(st, k1); c.nk1 = fmax2(c.l1.z, 1e-4f); c.l1.z = c.nk1;
  c.eta = relative_eta(st, m->p[MP_ETA]); c.Fd = fresnel_dielectric(c.nk1, c.eta); c.have = true;
  return c;
}
__device__ inline void opbr_base_sample(const MaterialRec* m, const ShState& st, V3 k1, float x0, float x1, float x2, BsdfSample& out,
    const OpbrBaseCtx* ctx = nullptr)
{
  const OpbrBaseParams o = opbr_base_params(m);
  V3 l1 = ctx ? ctx->l1 : to_local(st, k1);
  const float nk1 = ctx ? ctx->nk1 : fmax2(l1.z, 1e-4f); l1.z = nk1;
  float z = x2; // (Fc = 0: z = (z - 0) / (1 - 0))
  float eta = 0.0f, Fd = 0.0f;
  uint32_t lobe = 1u; // 1 metal, 2 dielectric reflection, 4 diffuse
  if (!(z < o.metalness)) {
    z = (z - o.metalness) / (1.0f - o.metalness);
    eta = ctx ? ctx->eta : relative_eta(st, o.eta);
    Fd = ctx ? ctx->Fd : fresnel_dielectric(nk1, eta);
    lobe = 2u;
    if (!(z < Fd)) lobe = 4u; // (transmission_weight = 0: z < 0 never holds)
  }
  if (lobe == 4u) {
    const float pBase = (1.0f - o.metalness) * (1.0f - Fd);
    const V3 l = gi_sample_hemisphere(x0, x1);
    const V3 k2 = to_world(st, l);
    if (!(l.z > 0.0f) || !(dot(k2, st.geomNormal) > 0.0f)) return;
    out.k2 = k2; out.pdf = pBase * (l.z / GI_PI);
    out.overPdf = (o.diffRough > 0.0f) ? eon_pi_f(o.baseColor, o.diffRough, l1, l) * o.baseWeight : o.albedo;
    out.event = EV_DIFFUSE | EV_REFLECTION;
    return;
  }
  const GgxOut g = ggx_sample(l1, o.alpha, x0, x1);
  const V3 k2 = to_world(st, g.l2);
  if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
  out.k2 = k2; out.event = EV_GLOSSY | EV_REFLECTION;
  if (lobe == 1u) {
    const V3 F = schlick_f82(o.albedo, o.metalTint, g.kh) * o.specWeight;
    out.pdf = o.metalness * g.pdf; out.overPdf = F * g.g2OverG1;
  } else {
    const float Fh = fresnel_dielectric(g.kh, eta);
    out.pdf = ((1.0f - o.metalness) * Fd) * g.pdf;
    out.overPdf = o.specColor * ((Fh / Fd) * g.g2OverG1);
  }
}
__device__ inline void opbr_base_evaluate(const MaterialRec* m, const ShState& st, V3 k1, V3 k2, BsdfEval& out, const OpbrBaseCtx* ctx = nullptr)
{
  const OpbrBaseParams o = opbr_base_params(m);
  V3 l1 = ctx ? ctx->l1 : to_local(st, k1); const V3 l2 = to_local(st, k2);
  const float nk1 = ctx ? ctx->nk1 : fmax2(l1.z, 1e-4f); l1.z = nk1;
  const float eta = ctx ? ctx->eta : relative_eta(st, o.eta);
  const float Fd = ctx ? ctx->Fd : fresnel_dielectric(nk1, eta);
  float fs, ps, khs; ggx_eval(l1, l2, o.alpha, fs, ps, khs);
  const float Fdh = fresnel_dielectric(khs, eta);
  const float cd = l2.z / GI_PI;
  const float diel = 1.0f - o.metalness;
  V3 gl = v3(0.0f, 0.0f, 0.0f); // (the coat term: Fch * fc = 0 * 0)
  if (o.metalness != 0.0f) { const V3 Fm = schlick_f82(o.albedo, o.metalTint, khs) * o.specWeight; gl = gl + (Fm * fs) * o.metalness; }
  gl = gl + (o.specColor * (Fdh * fs)) * diel;
  out.glossy = gl;
  const V3 rho = (o.diffRough > 0.0f && l2.z > 0.0f) ? eon_pi_f(o.baseColor, o.diffRough, l1, l2) * o.baseWeight : o.albedo;
  out.diffuse = rho * ((cd * diel) * (1.0f - Fd));
  out.pdf = 0.0f + (o.metalness * ps + diel * (Fd * ps + (1.0f - Fd) * cd)); // (Fc * pc = +0 first, as in opbr_evaluate_base)
}

// class 0 (diffuse only): the colour is the material's, or the hit's textured / primvar-driven base colour (found by the differential campaign of round 6,
// tests/fuzz_parity.py: the class ignored its base-colour binding while the oracle -- and every other class -- honours it)
__device__ __forceinline__ V3 diffuse_class_color(const MaterialRec* m, const ShState& st)
{
  return (st.texMask & (1u << TEX_BASE_COLOR)) ? st.texBaseColor : v3(m->p[0], m->p[1], m->p[2]);
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
    out.k2 = k2; out.pdf = l.z / GI_PI; out.overPdf = diffuse_class_color(m, st); out.event = EV_DIFFUSE | EV_REFLECTION;
    return;
  }
  if (klass == 1u) {
    UpsParams u = ups_params(m, st);
    V3 l1 = to_local(st, k1);
    float nk1 = fmax2(l1.z, 1e-4f); l1.z = nk1;
    float z = x2;
    float Fc = u.coat * (0.04f + 0.96f * schlick_w(nk1));
    V3 Fs = v3(0.0f, 0.0f, 0.0f); float ps = 0.0f;
    uint32_t lobe = 0u; // 0 coat, 1 specular, 2 diffuse (one ggx_sample for both glossy lobes, see opbr_sample)
    if (!(z < Fc)) {
      z = (z - Fc) / (1.0f - Fc);
      Fs = schlick3(u.F0, nk1);
      ps = fmax2(Fs.x, fmax2(Fs.y, Fs.z));
      lobe = (z < ps) ? 1u : 2u;
    }
    if (lobe != 2u) {
      GgxOut g = ggx_sample(l1, lobe == 0u ? u.coatAlpha : u.alpha, x0, x1);
      V3 k2 = to_world(st, g.l2);
      if (!g.valid || !(dot(k2, st.geomNormal) > 0.0f)) return;
      out.k2 = k2; out.event = EV_GLOSSY | EV_REFLECTION;
      if (lobe == 0u) {
        float Fh = u.coat * (0.04f + 0.96f * schlick_w(g.kh));
        float w = (Fh / Fc) * g.g2OverG1;
        out.pdf = Fc * g.pdf; out.overPdf = v3(w, w, w);
      } else {
        V3 Fh = schlick3(u.F0, g.kh);
        out.pdf = (1.0f - Fc) * ps * g.pdf; out.overPdf = Fh * (g.g2OverG1 / ps);
      }
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
  if (klass == SHADE_CLASS_OPBR_BASE) { opbr_base_sample(m, st, k1, x0, x1, x2, out); return; }
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
    out.diffuse = diffuse_class_color(m, st) * c; out.pdf = c;
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
  if (klass == SHADE_CLASS_OPBR_BASE) { opbr_base_evaluate(m, st, k1, k2, out); return; }
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
    if (idx < U.sphereCount) { const SphereLightRec l = sc.sphereLights[idx]; pos = v3(l.pos); em = v3(l.em); radius = v3(l.radius); area = l.area;
        dsPacked = l.ds; }
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
    // gi_decode_direction(l.t0), gi_decode_direction(l.t1) and their cross product, evaluated once on the host (gi_types.h)
    const LightFrame lf = sc.rectFrames[idx];
    float sx = (k2 - 0.5f) * l.width, sy = (k3 - 0.5f) * l.height;
    V3 t0 = v3(lf.t0), t1 = v3(lf.t1);
    V3 samplePos = (v3(l.origin) + t0 * sx) + t1 * sy;
    V3 dir = samplePos - surfacePos;
    dist = length(dir); dirToLight = gi_safe_div(dir, dist);
    V3 ln = v3(lf.n);
    float cosTheta = fmax2(0.0f, dot(-dirToLight, ln));
    float area = l.width * l.height;
    invPdf = gi_safe_div((area > 0.0f) ? (area * cosTheta) : 1.0f, dist * dist);
    power = v3(l.em) * U.lightIntensityMultiplier; dsPacked = l.ds;
  } else {
    uint32_t idx = (uint32_t)(k1 * (float)U.diskCount);
    const uint32_t last = U.diskCount - 1u; if (idx > last) idx = last;
    const DiskLightRec l = sc.diskLights[idx];
    const LightFrame lf = sc.diskFrames[idx];
    float sx, sy; gi_sample_disk(k2, k3, l.rx, l.ry, sx, sy);
    V3 t0 = v3(lf.t0), t1 = v3(lf.t1);
    V3 samplePos = (v3(l.origin) + t0 * sx) + t1 * sy;
    V3 dir = samplePos - surfacePos;
    dist = length(dir); dirToLight = gi_safe_div(dir, dist);
    V3 ln = v3(lf.n);
    float cosTheta = fmax2(0.0f, dot(-dirToLight, ln));
    float area = l.rx * l.ry * GI_PI;
    invPdf = gi_safe_div((area > 0.0f) ? (area * cosTheta) : 1.0f, dist * dist);
    power = v3(l.em) * U.lightIntensityMultiplier; dsPacked = l.ds;
  }
  power = power * U.exposureScale;
  invPdf = invPdf * (float)U.totalLightCount;
}


} // namespace gi
