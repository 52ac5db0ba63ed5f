// gi_stages.h -- per-path stage functions shared by the wavefront stage kernels (gi_kernels.hip: k_raygen, k_shade) and the fused
// persistent kernel (gi_path.hip: k_path): camera-ray generation, one shading step, the per-sample finish.  Device code, namespace gi.
#pragma once

#include "gi_shading.h"

namespace gi {

// Camera ray of (pixel, sample): RNG init, pixel jitter / filter importance sampling, thin lens, clip range
// (rp_main.rgen:215-288).  Returns the RNG state after the draws the reference makes here.
__device__ __forceinline__ void make_camera_ray(const FrameUniforms& U, uint32_t pixelIndex, uint32_t sampleIndex, V3& origin, V3& dir, float& tMin,
    float& tMax, uint32_t& rng)
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

// Per-sample finish (rp_main.rgen:489-496): hue-preserving clamp on the max channel, then max(0); NaNs are not filtered.
__device__ __forceinline__ V3 finish_sample(const FrameUniforms& U, V3 rad)
{
  const float mv = fmax2(rad.x, fmax2(rad.y, rad.z));
  if (mv > U.maxSampleValue) rad = rad * (U.maxSampleValue / mv);
  return v3(fmax2(0.0f, rad.x), fmax2(0.0f, rad.y), fmax2(0.0f, rad.z));
}

// ------------------------------------------------------------------------------------------------
// shade_segment: one closest-hit shading step + the post-trace part of the bounce loop for ONE path
// (rp_main.chit:132-493, rp_main.rgen:397-480), shared by the wavefront stage kernel k_shade and the fused persistent kernel
// k_path.  In: the hit record h = (t, u, v, triangle) / rd = (ray direction, -) [or a scattering event, see VOLUME_MISS] and the
// path state; out: the updated state, the next ray (no, k2, tMaxNext) if `cont`, and the shadow ray of this bounce if `shadow`.
// ------------------------------------------------------------------------------------------------
struct ShadeIO {
  V3 throughput, radiance; uint32_t bitfield, rng;                 // in / out: rp_main_payload.glsl:24-33
  bool cont, shadow, shadowFirst;                                    // out: path continues; a shadow ray is to be traced; this is bounce 0
  V3 no, k2; float tMaxNext;                                         // out: next ray
  V3 sdir, nee; float ld; uint32_t rngShadow;                        // out: shadow ray direction / distance, NEE contribution, rng copy (rp_main.rgen:399)
};
template <uint32_t KLASS, bool TEXTURED, bool VOLUME, bool NEE, bool PACKED = false>
__device__ __forceinline__ void shade_segment(const FrameUniforms& U, const SceneView& sc, float* M /* medium stack of the path (VOLUME) */, const F4& h,
    const F4& rd, ShadeIO& io)
{
  static_assert(KLASS != SHADE_CLASS_OPBR_BASE || (!TEXTURED && !VOLUME),
      "the BASE variant exists for untextured materials in renders without a medium stack (launchShade sends the others through the full kernel)");
  V3 throughput = io.throughput, radiance = io.radiance; uint32_t bitfield = io.bitfield, rng = io.rng;
  bool cont = false, shadow = false, shadowFirst = false; uint32_t rngShadow = 0u;
  V3 no = v3(0.0f, 0.0f, 0.0f), k2 = no, sdir = no, nee = no; float ld = 0.0f, tMaxNext = GI_FLT_MAX;
  const uint32_t bounce = bitfield & 0x00000fffu;

  const V3 rayDir = v3(rd.x, rd.y, rd.z);
  const uint32_t stackSize = VOLUME ? (U.mediumStackSize < MAX_MEDIUM_STACK ? U.mediumStackSize : MAX_MEDIUM_STACK) : 0u;
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
  setup_shading_state<PACKED>(sc, f2u(h.w), h.y, h.z, rayDir, ss);
  const MaterialRec* mat = &sc.materials[ss.material];
  if (TEXTURED && (mat->flags & MAT_FLAG_TEXTURED)) resolve_material_textures(sc, mat, rayDir, ss); // else ss.texMask stays 0 and folds away
  // geometry_tangent: the base lobes' tangent turned (full OpenPBR class only: an anisotropic material is never BASE); after the coat's own frame was made
  if ((KLASS == 2u || (KLASS == KLASS_DYNAMIC && mat->klass == 2u)) && ((uint32_t)mat->p[MP_FEATURES] & MATF_SPEC_ROTATION)) spec_turn_frame(mat, ss);
  const bool isDoubleSided = (ss.meshFlags & 2u) != 0u;
  // volume attenuation (rp_main.chit:160-186)
  float prevMediumIor = 1.0f, nextMediumIor = 1.0f;
  if (mediumIdx > 0u) {
    const float distance = h.x * U.metersPerSceneUnit;
    if (!VOLUME) { // empty medium stack: inside (1-bit toggle) -> Beer-Lambert with the HIT material's absorption coefficient (:169-173)
      if (((KLASS == KLASS_DYNAMIC) ? mat->klass : KLASS) == 2u
          || KLASS == SHADE_CLASS_OPBR_BASE) throughput = throughput * v3(gi_expf(-mat->p[MP_SIGMA_A] * distance), gi_expf(-mat->p[MP_SIGMA_A + 1] * distance),
          gi_expf(-mat->p[MP_SIGMA_A + 2] * distance));
    } else { // the medium on top of the stack (:174-184)
      const float* m = M + (mediumIdx - 1u) * MEDIUM_FLOATS;
      prevMediumIor = m[0];
      if (mediumIdx > 1u) nextMediumIor = M[(mediumIdx - 2u) * MEDIUM_FLOATS];
      throughput = throughput * v3(gi_expf(-m[5] * distance), gi_expf(-m[6] * distance), gi_expf(-m[7] * distance));
    }
  }
  // mdl_thin_walled (:155-157): OpenPBR geometry_thin_walled
  const bool thinWalled = ((KLASS == KLASS_DYNAMIC) ? mat->klass : KLASS) == 2u && ((uint32_t)mat->p[MP_FEATURES] & MATF_THIN_WALLED) != 0u;
  ss.thinWalled = thinWalled;
  ss.sssVolume = VOLUME; // a medium stack exists: OpenPBR's volumetric subsurface lobe is live
  // iorCurrent / iorOther (:188-189)
  if (VOLUME) { ss.ior1 = (ss.frontFace || thinWalled) ? prevMediumIor : -1.0f; ss.ior2 = (ss.frontFace || thinWalled) ? -1.0f : nextMediumIor; }
  // emission (rp_main.chit:293-343): uniform EDF, radiance == emission colour where cos > 0
  V3 em = (ss.texMask & (1u << TEX_EMISSION)) ? ss.texEmission : v3(mat->p[3], mat->p[4], mat->p[5]);
  if (em.x != 0.0f || em.y != 0.0f || em.z != 0.0f) {
    if (ss.frontFace || !isDoubleSided) {
      const float c = dot(-rayDir, ss.normal);
      if (c > 0.0f) {
        // emission_edf (open_pbr_surface.mtlx:590-619): seen through the coat (BASE variant: no coat, factor 1)
        if (((KLASS == KLASS_DYNAMIC) ? mat->klass : KLASS) == 2u)
          em = em * opbr_emission_factor(mat->p[12], v3(mat->p[19], mat->p[20], mat->p[21]), mat->p[MP_COAT_F0], c);
        radiance = radiance + throughput * (em * U.exposureScale);
      }
    }
  }
  // BSDF importance sampling (:361-389); xi = next4f, .w is drawn but unused by the closed forms
  const float x0 = gi_next1f(rng), x1 = gi_next1f(rng), x2 = gi_next1f(rng); (void)gi_next1f(rng);
  BsdfSample bs;
  OpbrBaseCtx baseCtx; baseCtx.have = false;
  if constexpr (KLASS == SHADE_CLASS_OPBR_BASE && NEE) { // sample and evaluate share their starting point (gi_shading.h OpbrBaseCtx)
    baseCtx = opbr_base_ctx(mat, ss, -rayDir);
    bs.event = EV_ABSORB; bs.pdf = 0.0f; bs.overPdf = v3(0.0f, 0.0f, 0.0f); bs.k2 = v3(0.0f, 0.0f, 0.0f); // (bsdf_sample's initialisation)
    opbr_base_sample(mat, ss, -rayDir, x0, x1, x2, bs, &baseCtx);
  } else bsdf_sample<KLASS>(mat, ss, -rayDir, x0, x1, x2, bs);
  throughput = throughput * bs.overPdf;
  k2 = bs.k2;
  const bool isTransmission = (bs.event & EV_TRANSMISSION) != 0u;
  // NEE (:394-444)
  if (NEE && (bs.event & (EV_DIFFUSE | EV_GLOSSY))) { // NEE is a compile-time variant: light sampling + BSDF evaluation cost registers even when off
    const float k0 = gi_next1f(rng), k1 = gi_next1f(rng), kk2 = gi_next1f(rng), k3 = gi_next1f(rng);
    V3 dirToLight, lightPower; float lightDist, invPdf; uint32_t ds;
    sample_light(sc, U, k0, k1, kk2, k3, ss.position, dirToLight, lightDist, lightPower, invPdf, ds);
    if ((lightDist > 0.0f) && dot(dirToLight, ss.geomNormal) > 0.0f) {
      BsdfEval ev;
      if constexpr (KLASS == SHADE_CLASS_OPBR_BASE && NEE) { // bsdf_evaluate's preamble, then the variant with the shared context
        ev.diffuse = v3(0.0f, 0.0f, 0.0f); ev.glossy = v3(0.0f, 0.0f, 0.0f); ev.pdf = 0.0f;
        if (dot(ss.normal, dirToLight) > 0.0f) opbr_base_evaluate(mat, ss, -rayDir, dirToLight, ev, &baseCtx);
      } else bsdf_evaluate<KLASS>(mat, ss, -rayDir, dirToLight, ev);
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
    rngShadow = rng; // the shadow payload gets a copy of the rng as it is HERE, before the Russian-roulette draw (rp_main.rgen:399)
  }
  // NEE AOV (rp_main.rgen:431-435): bounce 0 only; a shadow ray that is not traced counts as "not shadowed"
  shadowFirst = bounce == 0u; // the caller records "not shadowed" for the NEE AOV when shadowFirst && !shadow
  // (a BASE material has no transmissive lobe) medium stack (:447-480); a thin-walled surface has the same medium on both sides
  if (KLASS != SHADE_CLASS_OPBR_BASE && !thinWalled && isTransmission) {
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
            // entered through the subsurface lobe
            if (bs.event & EV_SUBSURFACE) { m[1] = mat->p[59]; m[2] = mat->sss[0]; m[3] = mat->sss[1]; m[4] = mat->sss[2]; m[5] = mat->sss[3];
                m[6] = mat->sss[4]; m[7] = mat->sss[5]; }
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
    const float sinTheta = gi_sqrt(fmax2(0.0f, 1.0f - cosTheta * cosTheta));
    float sp, cp; gi_sincos2pi(x1, &sp, &cp);
    V3 t, b; gi_orthonormal_basis(k2, t, b);
    k2 = ((t * sinTheta) * cp + (b * sinTheta) * sp) + k2 * cosTheta;
    bitfield &= ~0x40000000u;
  }
  bitfield++;
  cont = ((bitfield & 0x00000fffu) < U.maxBounces) && !(bitfield & 0x80000000u); // loop test :298-304
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
  io.throughput = throughput; io.radiance = radiance; io.bitfield = bitfield; io.rng = rng;
  io.cont = cont; io.shadow = shadow; io.shadowFirst = shadowFirst; io.no = no; io.k2 = k2; io.tMaxNext = tMaxNext;
  io.sdir = sdir; io.nee = nee; io.ld = ld; io.rngShadow = rngShadow;
}

// A new path's Slot (rp_main.rgen:274-276): throughput 1, bitfield 0, radiance 0, the rng state after the camera draws, its work item
__device__ __forceinline__ void slot_begin_path(Slot* S, uint32_t rng, uint32_t pixelLocal, uint32_t sLocal)
{
  st4(&S->thr, 1.0f, 1.0f, 1.0f, u2f(0u));
  st4(&S->rad, 0.0f, 0.0f, 0.0f, u2f(rng));
  st4(&S->id, u2f(pixelLocal), u2f(sLocal), u2f(1u), 0.0f);
}
// FLAG_DEFER_SLOT: the first segment of a camera path whose Slot k_raygen did not write has been traced.  A hit writes the Slot now (the path goes on exactly
// as if k_raygen had written it).  A miss retires the sample on the spot -- radiance 0 + throughput 1 x background, then the per-sample finish, the arithmetic
// of k_raygen's finish of a REGEN_MISSED entry (rp_main.miss:68-86, rp_main.rgen:489-496) -- and returns true: the slot goes back to the regen queue as
// REGEN_FRESH, "nothing to finish, memory unwritten".  Scenes with a dome image or medium stacks need the slot at a miss (dome_miss / the scattering test): the
// caller writes it first and takes the ordinary route.
__device__ __forceinline__ void retire_fresh_miss(const FrameUniforms& U, const FreshRec& f, F4* __restrict__ sampleBuf)
{
  uint32_t pixelLocal, sLocal; work_item(U, f.work, pixelLocal, sLocal);
  V3 rad = v3(0.0f, 0.0f, 0.0f) + v3(1.0f, 1.0f, 1.0f) * v3(U.background);
  const float mv = fmax2(rad.x, fmax2(rad.y, rad.z));
  if (mv > U.maxSampleValue) rad = rad * (U.maxSampleValue / mv);
  st4(&sampleBuf[sample_record(U, pixelLocal, sLocal)], fmax2(0.0f, rad.x), fmax2(0.0f, rad.y), fmax2(0.0f, rad.z), 0.0f);
}
__device__ __forceinline__ void begin_fresh_path(const FrameUniforms& U, const PathState& st, uint32_t slot, const FreshRec& f)
{
  uint32_t pixelLocal, sLocal; work_item(U, f.work, pixelLocal, sLocal);
  slot_begin_path(&st.slots[slot], f.rng, pixelLocal, sLocal);
}
// (a first segment that HIT is begun by k_shade, which gathers the ray record and the FreshRec beside it and completes the Slot in one go)

} // namespace gi
