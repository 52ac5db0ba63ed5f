// gi_shade.hip -- k_shade, the closest-hit stage of the wavefront path tracer (gfx950),
// one instantiation per shade class x {textured, volume, NEE, record layout};
// replaces rp_main.chit (/root/reference/src/gi/shaders/rp_main.chit:132-493) and the post-trace part of the bounce loop (rp_main.rgen:397-480).  The
// arithmetic is gi_shading.h / gi_stages.h (shade_segment).  Built with -ffp-contract=off (arithmetic contract, gi_device_math.h).

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
// k_shade: closest-hit shading + the post-trace part of the bounce loop, over the HIT queue only
// (rp_main.chit:132-493, rp_main.rgen:397-480).  Misses never get here (k_trace routes them to k_raygen).
// ------------------------------------------------------------------------------------------------
// minimum resident waves per SIMD asked of the register allocator for the plain OpenPBR variants: without NEE 4 (128 VGPRs, 3 spilled: the natural 3 waves
// measured slower, r04j); with NEE 1, i.e. what its 164 VGPRs allow -- 3 (squeezed to 4 waves it spills 14 and is slower, r04c)
constexpr int SHADE_OPENPBR_PLAIN_WAVES = 4, SHADE_OPENPBR_NEE_WAVES = 1;
#ifndef GI_SHADE_BASE_WAVES      // experiment knobs (tools/build_variant.py): minimum waves per SIMD asked for the OpenPBR BASE variant, without / with NEE
#define GI_SHADE_BASE_WAVES 1
#endif
#ifndef GI_SHADE_BASE_NEE_WAVES
#define GI_SHADE_BASE_NEE_WAVES 1
#endif
template <uint32_t KLASS, bool TEXTURED, bool VOLUME, bool NEE, bool PACKED>
// (forcing the plain variant to 5 waves/SIMD -- amdgpu_waves_per_eu((...) ? 5 : 1, 8): 90 VGPRs, no spills -- is SLOWER: C2 shade 178 -> 190 ms;
// the stage is bound by the memory pipeline's scattered 16-byte requests, not by latency hiding)
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((KLASS == 2u && !TEXTURED && !VOLUME)
    ? (NEE ? SHADE_OPENPBR_NEE_WAVES : SHADE_OPENPBR_PLAIN_WAVES) : (KLASS == SHADE_CLASS_OPBR_BASE
    ? (NEE ? GI_SHADE_BASE_NEE_WAVES : GI_SHADE_BASE_WAVES) : 1), 8))) void k_shade(FrameUniforms U, SceneView sc, PathState st, QueueSet qs, Counters* cnt,
    uint32_t par, uint32_t hitClass /* the HIT queue read: KLASS, or a variant's whose hits this launch shades with the full kernel */)
{
  __shared__ AppendScratch<3> sh;
  const uint32_t qNext = Q_TRACE_A + (par ^ 1u), qRegen = Q_REGEN_A + (par ^ 1u), qHit = Q_HIT + hitClass;
  QueueReader rdr; reader_init(rdr, cnt, qHit, qs.cap);
  const uint32_t n = rdr.pre[NSHARD];
  const uint32_t stride = gridDim.x * BLOCK;
  uint32_t trip = 0;
  for (uint32_t base = blockIdx.x * BLOCK; base < n; base += stride, trip++) {
    const uint32_t i = base + threadIdx.x;
    bool cont = false, ended = false, shadow = false, shadowFirst = false; uint32_t slot = 0, rngShadow = 0u;
    V3 no = v3(0.0f, 0.0f, 0.0f), k2 = no, sdir = no, nee = no; float ld = 0.0f, tMaxNext = GI_FLT_MAX;
    if (i < n) {
      const uint32_t e = qs.slot[qHit][reader_index(rdr, i)];
      // the ray's record in the queue it was traced from (untouched until the next iteration's producers)
      const uint32_t ri = e & HIT_INDEX_MASK, qT = Q_TRACE_A + par;
      const bool fresh = (e & HIT_FRESH) != 0u;
      slot = qs.slot[qT][ri] & ~TRACE_FRESH;
      F4 h = ld4(&qs.a[qT][ri]);
      const F4 rd = ld4(&qs.b[qT][ri]);
      h.w = (e & HIT_VOLUME) ? u2f(VOLUME_MISS) : u2f(f2u(h.w) & 0x0fffffffu); // strip the class bits / mark the scattering event for shade_segment
      Slot* S = &st.slots[slot];
      F4 tb = F4{1.0f, 1.0f, 1.0f, u2f(0u)}, rr = F4{0.0f, 0.0f, 0.0f, 0.0f}; // rp_main.rgen:274-276
      // first hit of a deferred path: its rng / work item are beside the ray record; the Slot is completed here (work item now, throughput / radiance below)
      if (fresh) {
        const FreshRec f = qs.fresh[par][ri];
        rr.w = u2f(f.rng);
        uint32_t pixelLocal, sLocal; work_item(U, f.work, pixelLocal, sLocal);
        st4(&S->id, u2f(pixelLocal), u2f(sLocal), u2f(1u), 0.0f);
      } else { tb = ld4(&S->thr); rr = ld4(&S->rad); }
      ShadeIO io; io.throughput = v3(tb.x, tb.y, tb.z); io.radiance = v3(rr.x, rr.y, rr.z); io.bitfield = f2u(tb.w); io.rng = f2u(rr.w);
      float* M = VOLUME ? st.media + (size_t)slot * st.mediaStride : nullptr; // this path's medium stack + walkSegmentPdf
      shade_segment<KLASS, TEXTURED, VOLUME, NEE, PACKED>(U, sc, M, h, rd, io);
      cont = io.cont; ended = !cont; shadow = io.shadow; shadowFirst = io.shadowFirst; no = io.no; k2 = io.k2; tMaxNext = io.tMaxNext;
      sdir = io.sdir; nee = io.nee; ld = io.ld; rngShadow = io.rngShadow;
      // NEE AOV (rp_main.rgen:431-435): an untraced shadow ray counts as "not shadowed"
      if (NEE && st.neeKey && shadowFirst && !shadow) nee_aov_record(st, slot, false);
      st4(&S->thr, io.throughput.x, io.throughput.y, io.throughput.z, u2f(io.bitfield));
      st4(&S->rad, io.radiance.x, io.radiance.y, io.radiance.z, u2f(io.rng));
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
      st4(&qs.b[Q_SHADOW][idx[2]], sdir.x, sdir.y, sdir.z, u2f(rngShadow)); // .w: rng state for the any-hit test of cutouts
      st4(&qs.c[Q_SHADOW][idx[2]], nee.x, nee.y, nee.z, shadowFirst ? 1.0f : 0.0f);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_debug_bsdf: the closed-form BSDF entry points on explicit shading frames (device-side known-answer tests)
// ------------------------------------------------------------------------------------------------
__global__ void k_debug_bsdf(const MaterialRec* mat, uint32_t shadeClass, uint32_t count, const float* __restrict__ in, float* __restrict__ out)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float* p = in + 22 * (size_t)i; float* o = out + 15 * (size_t)i;
  ShState st; st.normal = v3(p); st.tangentU = v3(p + 3); st.tangentV = v3(p + 6); st.geomNormal = v3(p + 9);
  st.position = v3(0.0f, 0.0f, 0.0f); st.frontFace = (p[21] < 0.5f); st.meshFlags = 0u; st.material = 0u;
  st.u = 0.0f; st.v = 0.0f; st.texMask = 0u; st.ior1 = 0.0f; st.ior2 = 0.0f;
      st.thinWalled = mat->klass == 2u && ((uint32_t)mat->p[MP_FEATURES] & MATF_THIN_WALLED) != 0u; st.sssVolume = false; st.hasCoatFrame = false; st.mesh = 0u;
      st.prim = 0u; st.vi[0] = st.vi[1] = st.vi[2] = 0u; st.instanceId = 0; st.hu = st.hv = 0.0f;
  if (mat->klass == 2u && ((uint32_t)mat->p[MP_FEATURES] & MATF_SPEC_ROTATION)) spec_turn_frame(mat, st); // geometry_tangent, as the shade stage does
  BsdfSample bs; BsdfEval ev;
  if (shadeClass == SHADE_CLASS_OPBR_BASE) { bsdf_sample<SHADE_CLASS_OPBR_BASE>(mat, st, v3(p + 12), p[18], p[19], p[20], bs);
      bsdf_evaluate<SHADE_CLASS_OPBR_BASE>(mat, st, v3(p + 12), v3(p + 15), ev); }
  else { bsdf_sample<KLASS_DYNAMIC>(mat, st, v3(p + 12), p[18], p[19], p[20], bs); bsdf_evaluate<KLASS_DYNAMIC>(mat, st, v3(p + 12), v3(p + 15), ev); }
  o[0] = bs.k2.x; o[1] = bs.k2.y; o[2] = bs.k2.z; o[3] = bs.overPdf.x; o[4] = bs.overPdf.y; o[5] = bs.overPdf.z; o[6] = bs.pdf; o[7] = (float)bs.event;
  o[8] = ev.diffuse.x; o[9] = ev.diffuse.y; o[10] = ev.diffuse.z; o[11] = ev.glossy.x; o[12] = ev.glossy.y; o[13] = ev.glossy.z; o[14] = ev.pdf;
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------------
void launchShade(hipStream_t s, uint32_t blocks, uint32_t klass, bool textured, bool volume, const FrameUniforms& U, const SceneView& sc, const PathState& st,
    const QueueSet& qs, Counters* cnt, uint32_t par)
{
  // `klass` is a SHADE class (gi_types.h): the HIT queue to read.  The OpenPBR BASE
  // variant's kernel exists for untextured materials in renders without a medium stack;
  // its hits go through the full OpenPBR kernel otherwise (same bits: gi_shading.h "BASE variant")
  const uint32_t hitClass = klass;
  if (klass == SHADE_CLASS_OPBR_BASE && (textured || volume)) klass = 2u;
#define GI_LAUNCH_SHADE4(K, T, V, N) do { \
    if (sc.shadePacked) hipLaunchKernelGGL((k_shade<K, T, V, N, true>), dim3(blocks), dim3(BLOCK), 0, s, U, sc, st, qs, cnt, par, hitClass); \
    else hipLaunchKernelGGL((k_shade<K, T, V, N, false>), dim3(blocks), dim3(BLOCK), 0, s, U, sc, st, qs, cnt, par, hitClass); } while (0)
#define GI_LAUNCH_SHADE3(K, T, V) do { if (nee) GI_LAUNCH_SHADE4(K, T, V, true); else GI_LAUNCH_SHADE4(K, T, V, false); } while (0)
#define GI_LAUNCH_SHADE(K) do { \
    if (volume) { if (textured) GI_LAUNCH_SHADE3(K, true, true); else GI_LAUNCH_SHADE3(K, false, true); } \
    else if (textured) GI_LAUNCH_SHADE3(K, true, false); \
    else GI_LAUNCH_SHADE3(K, false, false); } while (0)
  const bool nee = (U.flags & FLAG_NEE) != 0u;
  if (klass == 0u) GI_LAUNCH_SHADE(0u); else if (klass == 1u) GI_LAUNCH_SHADE(1u); else if (klass == 2u) GI_LAUNCH_SHADE(2u);
      else GI_LAUNCH_SHADE3(SHADE_CLASS_OPBR_BASE, false, false);
#undef GI_LAUNCH_SHADE4
#undef GI_LAUNCH_SHADE3
#undef GI_LAUNCH_SHADE
}

void launchDebugBsdf(hipStream_t s, const MaterialRec* mat, uint32_t shadeClass, uint32_t count, const float* in, float* out)
{
  hipLaunchKernelGGL(k_debug_bsdf, dim3((count + 63u) / 64u), dim3(64), 0, s, mat, shadeClass, count, in, out);
}

} // namespace gi
