// gi_aov.hip -- k_aov, the non-colour AOVs of a frame (gfx950): replays the camera rays of every sample in order and writes what the reference's ray generation
// and closest-hit shaders write for the primary hit (/root/reference/src/gi/shaders/rp_main.rgen:132-183, 517-520; rp_main.chit:192-290).

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
// k_aov: the non-colour AOVs (rp_main.rgen:132-183, 517-520; rp_main.chit:192-290).  They depend only on the primary hit
// of each sample and are overwritten sample after sample, so they are produced by a separate per-pixel pass that
// replays the camera rays of samples 0..spp-1 in order (same RNG streams) -- exact, and off the colour path's hot loop.
// ------------------------------------------------------------------------------------------------
__device__ inline V3 bsdf_albedo(const MaterialRec* m, const ShState& st, V3 k1)
{
  float nk1 = fmax2(dot(st.normal, k1), 1e-4f);
  if (m->klass == 0u) return diffuse_class_color(m, st);
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
  const float nk1c = st.hasCoatFrame ? fmax2(dot(st.coatNormal, k1), 1e-4f) : nk1; // the coat's Fresnel term in its own frame (geometry_coat_normal)
  float Fc = o.coat * (o.coatF0 + (1.0f - o.coatF0) * schlick_w(nk1c));
  float Fd = fresnel_dielectric(nk1, eta);
  float base = 1.0f - Fc, diel = 1.0f - o.metalness;
  V3 diffuse = (o.albedo * o.coatTint) * (base * diel * (1.0f - Fd) * (1.0f - o.tw));
  V3 glossy = v3(Fc, Fc, Fc) + ((schlick_f82(o.albedo, o.metalTint, nk1) * o.specWeight) * o.coatTint) * (base * o.metalness)
              + (o.specColor * o.coatTint) * (base * diel * Fd);
  if (o.filmWeight > 0.0f) { // thin film: the two Fresnel factors carry the film's reflectance, what lies beneath the interface its complement
    const V3 Fdf = opbr_film_dielectric(o, nk1, eta, Fd);
    diffuse = ((o.albedo * o.coatTint) * (v3(1.0f, 1.0f, 1.0f) - Fdf)) * (base * diel * (1.0f - o.tw));
    glossy = v3(Fc, Fc, Fc) + ((opbr_film_metal(o, nk1, schlick_f82(o.albedo, o.metalTint, nk1)) * o.specWeight) * o.coatTint) * (base * o.metalness)
             + ((o.specColor * o.coatTint) * Fdf) * (base * diel);
  }
  if (o.fuzzWeight > 0.0f) { // the fuzz layer keeps P = fuzz_weight * min(E, 1) of the light (tinted), what is beneath gets 1 - P
    const float Pf = o.fuzzWeight * fmin2(fuzz_albedo(nk1, o.fuzzAlpha), 1.0f);
    return (diffuse + glossy) * (1.0f - Pf) + o.fuzzColor * Pf;
  }
  return diffuse + glossy;
}

template <uint32_t STACK, bool OVERFLOW, bool PACKED>
__global__ __launch_bounds__(TRACE_BLOCK) void k_aov(FrameUniforms U, SceneView sc, AovTargets A, uint32_t ldsNodes, uint32_t ldsTris)
{
  extern __shared__ uint4 s_dyn[];
  uint2 (*s_stack)[TRACE_BLOCK] = reinterpret_cast<uint2 (*)[TRACE_BLOCK]>(s_dyn);
  uint4* s_nodes = s_dyn + (STACK * TRACE_BLOCK * sizeof(uint2)) / sizeof(uint4);
  uint4* s_tris = s_nodes + ldsNodes * 5u;
  for (uint32_t i = threadIdx.x; i < ldsNodes * 5u; i += TRACE_BLOCK) s_nodes[i] = reinterpret_cast<const uint4*>(sc.nodes)[i];
  for (uint32_t i = threadIdx.x; i < ldsTris * 3u; i += TRACE_BLOCK) s_tris[i] = reinterpret_cast<const uint4*>(sc.tris)[(i / 3u) * 4u + (i % 3u)];
  __syncthreads();
  const uint32_t p = blockIdx.x * TRACE_BLOCK + threadIdx.x;
  if (p >= U.pixelCount) return;
  const uint32_t pixelIndex = tile_to_image_pixel(U, p);
  auto put3 = [&](F4* buf, V3 v) { if (buf) { float* d = reinterpret_cast<float*>(&buf[pixelIndex]); d[0] = v.x; d[1] = v.y; d[2] = v.z; } };
  auto clr3 = [&](F4* buf, int id) { put3(buf, v3(A.clear[id][0], A.clear[id][1], A.clear[id][2])); };
  clr3(A.barycentrics, 3); clr3(A.texcoords, 4); clr3(A.opacity, 7); clr3(A.tangents, 8); clr3(A.bitangents, 9); clr3(A.thinWalled, 10);
  if (A.objectId) A.objectId[pixelIndex] = (int)f2u(A.clear[11][0]);
  if (A.depth) A.depth[pixelIndex] = A.clear[12][0];
  if (A.faceId) A.faceId[pixelIndex] = (int)f2u(A.clear[13][0]);
  if (A.instanceId) A.instanceId[pixelIndex] = (int)f2u(A.clear[14][0]);
  clr3(A.doubleSided, 15);
  V3 curNormal = v3(0.0f, 0.0f, 0.0f), curAlbedo = curNormal;
  if (U.sampleOffset == 0u) { clr3(A.normal, 1); clr3(A.albedo, 16); curNormal = v3(A.clear[1][0], A.clear[1][1], A.clear[1][2]);
      curAlbedo = v3(A.clear[16][0], A.clear[16][1], A.clear[16][2]); }
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
    uint32_t matWord;
    if (!traverse<false, false, STACK, OVERFLOW, false,
        true>(sc, s_nodes, ldsNodes, s_tris, ldsTris, s_stack, origin, dir, tMin, tMax, t, u, v, tri, matWord, tc, rng)) continue;
    ShState ss;
    setup_shading_state<PACKED>(sc, tri, u, v, dir, ss);
    const uint4* tp = reinterpret_cast<const uint4*>(sc.tris) + (size_t)tri * 4u;
    const uint32_t instIdx = tp[2].z;
    if (A.opacity) {
      // rp_main.chit:199-205 writes (1,0,0) for materials without cutout transparency; for the others the any-hit shader has written
      // viridis(opacity) (white for 0) of its last candidate (rp_main.ahit:45-49) -- restated as the ACCEPTED primary hit's opacity
      V3 c = v3(1.0f, 0.0f, 0.0f);
      if (matWord & (1u << 28)) { const float op = cutout_opacity_at(sc, matWord, tri, u, v); c = (op == 0.0f) ? v3(1.0f, 1.0f, 1.0f) : gi_colormap_viridis(op);
          }
      put3(A.opacity, c);
    }
    put3(A.tangents, (ss.tangentU + v3(1.0f, 1.0f, 1.0f)) * 0.5f);
    put3(A.bitangents, (ss.tangentV + v3(1.0f, 1.0f, 1.0f)) * 0.5f);
    put3(A.barycentrics, v3(1.0f - u - v, u, v));
    if (A.texcoords) {
      const uint4 td = tp[3];
      const float bx = 1.0f - u - v;
      if (PACKED) { const TriShade& q = sc.triShade[td.x];
          put3(A.texcoords, v3((bx * q.uv[0][0] + u * q.uv[1][0]) + v * q.uv[2][0], (bx * q.uv[0][1] + u * q.uv[1][1]) + v * q.uv[2][1], 0.0f)); }
      else {
      const FVertex* va = &sc.verts[td.x]; const FVertex* vb = &sc.verts[td.y]; const FVertex* vc = &sc.verts[td.z];
      put3(A.texcoords, v3((bx * va->u + u * vb->u) + v * vc->u, (bx * va->v + u * vb->v) + v * vc->v, 0.0f));
      }
    }
    // rp_main.chit:218-220
    { const MaterialRec* tm = &sc.materials[ss.material];
        put3(A.thinWalled, (tm->klass == 2u && ((uint32_t)tm->p[MP_FEATURES] & MATF_THIN_WALLED) != 0u) ? v3(1.0f, 0.0f, 0.0f) : v3(0.0f, 1.0f, 0.0f)); }
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
// host-callable launchers
// ------------------------------------------------------------------------------------------------
void launchAov(hipStream_t s, const FrameUniforms& U, const SceneView& sc, const AovTargets& A)
{
  uint32_t ln, lt, bytes; traceLdsLayout(sc, ln, lt, bytes);
  bytes = (sc.bvhDepth <= 8u ? 8u : 16u) * TRACE_BLOCK * (uint32_t)sizeof(uint2) + ln * 80u + lt * 48u; // k_aov has no 4-entry variant
  const uint32_t blocks = (U.pixelCount + TRACE_BLOCK - 1u) / TRACE_BLOCK;
#define GI_LAUNCH_AOV(P) do { \
  if (sc.bvhDepth <= 8u) hipLaunchKernelGGL((k_aov<8, false, P>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, U, sc, A, ln, lt); \
  else if (sc.bvhDepth <= 16u) hipLaunchKernelGGL((k_aov<16, false, P>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, U, sc, A, ln, lt); \
  else hipLaunchKernelGGL((k_aov<16, true, P>), dim3(blocks), dim3(TRACE_BLOCK), bytes, s, U, sc, A, ln, lt); } while (0)
  if (sc.shadePacked) GI_LAUNCH_AOV(true); else GI_LAUNCH_AOV(false);
#undef GI_LAUNCH_AOV
}

} // namespace gi
