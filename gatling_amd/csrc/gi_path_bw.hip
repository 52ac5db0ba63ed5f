// gi_path_bw.hip -- k_path_bw: the fused persistent path kernel as a WAVE-LOCAL WAVEFRONT, for LDS-resident scenes without next-event
// estimation (configs C1 / C2).
//
// k_path (gi_path.hip) keeps one path per lane in registers and walks all 64 lanes through camera ray -> traversal -> shading in lock
// step.  The counters say what that costs (profiles/r02d_bench_c2.log): the kernel saturates the VALU issue slots, but only 55 % of the
// lanes of an average instruction do useful work -- a traversal lasts as long as the wave's slowest ray (1.6 node steps on average, 3
// for the slowest), shading runs for the lanes that hit, regeneration for the ~45 % whose path just ended.  Here a wave owns PW > 64
// paths whose state lives in LDS (17 dwords each, structure of arrays) and three wave-private queues of path ids, one per stage:
//
//   TRAV   lanes walk rays through the BVH one step at a time (wave_step); a lane whose ray ends stores the hit in the path's state,
//          queues the path for SHADE (hit) or REGEN (miss) and takes the next ray from the TRAV queue -- the lanes' traversal state
//          stays in registers while the wave runs another stage
//   SHADE  64 queued hits at a time: closest-hit shading + the bounce loop's tail (shade_segment); survivors -> TRAV, ended -> REGEN
//   REGEN  64 ended paths at a time: per-sample finish -> per-sample colour buffer, next work item, camera ray -> TRAV
//
// A stage other than TRAV runs when it has a full wave's worth of paths queued (or when traversal runs dry), so shading and
// regeneration execute with every lane active.  Queues are private to a wave: no atomics, no barriers after the scene is staged.
// Replaces the same reference code as k_path (rp_main.rgen:185-521, traceRayEXT :381-393, rp_main.chit, rp_main.miss:68-86); all per-path
// arithmetic is the shared stage code (make_camera_ray, wave_step, shade_segment, finish_sample): images are bit-identical to k_path's,
// the wavefront pipeline's and the oracle's -- only the order in which independent paths advance differs.

#include <hip/hip_runtime.h>

#include "gi_device_math.h"
#include "gi_kernels.h"
#include "gi_types.h"
#include "gi_queues.h"
#include "gi_traversal.h"
#include "gi_shading.h"
#include "gi_stages.h"

namespace gi {

// per-path state in LDS, one array of PW dwords per field
enum : uint32_t { F_THR = 0, F_RAD = 3, F_BITS = 6, F_RNG = 7, F_WORK = 8, F_RO = 9 /* origin, or the hit (t, u, v) */, F_RD = 12,
    F_TMIN = 15 /* tMin, or the hit triangle */,
                  F_TMAX = 16, F_COUNT = 17 };
constexpr uint32_t NO_WORK = 0xffffffffu; // F_WORK of a path that carries no sample (initial state)
// paths per wave (a compile-time choice since the environment interface shrank): 3 blocks per
// CU; measured 96 / 128 / 160 / 192 / 256 -> 7426 / 6431 / 6621 / 6651 / 3814 Msamples/s on C2
constexpr uint32_t PATH_BW_PATHS_DEFAULT = 96u;

constexpr int PATH_BW_WAVES = 3; // resident waves per SIMD the register allocation aims for (168 VGPRs: 3; 4 needs <= 128 and spills 43 registers)
template <uint32_t KLASS, bool TEXTURED, bool CUTOUT, bool COUNT, uint32_t STACK>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(PATH_BW_WAVES, 8))) void k_path_bw(FrameUniforms U, SceneView sc, PathState st,
    Counters* cnt, F4* __restrict__ sampleBuf,
                                                         uint32_t ldsNodes, uint32_t ldsTris, uint32_t chunk, uint32_t PW, uint32_t thrShade, uint32_t thrRegen,
                                                             uint32_t thrDry)
{
  extern __shared__ uint4 s_dyn[];
  uint2 (*s_stack)[TRACE_BLOCK] = reinterpret_cast<uint2 (*)[TRACE_BLOCK]>(s_dyn);
  uint4* s_nodes = s_dyn + (STACK * TRACE_BLOCK * sizeof(uint2)) / sizeof(uint4);
  uint4* s_tris = s_nodes + ldsNodes * 5u;
  __shared__ WaveTri s_wave[TRACE_BLOCK / 64];
  const uint32_t wave = threadIdx.x >> 6, lane = __lane_id();
  WaveTri& W = s_wave[wave];
  // this wave's path state [F_COUNT][PW] and queues [3][PW]
  GI_LDS uint32_t* S = (GI_LDS uint32_t*)(s_tris + ldsTris * 3u) + (size_t)wave * (F_COUNT + 3u) * PW;
  GI_LDS uint32_t* qT = S + F_COUNT * PW; GI_LDS uint32_t* qS = qT + PW; GI_LDS uint32_t* qR = qS + PW;
  for (uint32_t i = threadIdx.x; i < ldsNodes * 5u; i += TRACE_BLOCK) s_nodes[i] = reinterpret_cast<const uint4*>(sc.nodes)[i];
  for (uint32_t i = threadIdx.x; i < ldsTris * 3u; i += TRACE_BLOCK) s_tris[i] = reinterpret_cast<const uint4*>(sc.tris)[(i / 3u) * 4u + (i % 3u)];
  for (uint32_t p = lane; p < PW; p += 64u) { S[F_WORK * PW + p] = NO_WORK; qR[p] = p; } // every path starts "ended, nothing to finish"
  __syncthreads(); // the only barrier

  auto ldf = [&](uint32_t f, uint32_t p) -> float { return u2f(S[f * PW + p]); };
  auto ldu = [&](uint32_t f, uint32_t p) -> uint32_t { return S[f * PW + p]; };
  auto stf = [&](uint32_t f, uint32_t p, float v) { S[f * PW + p] = f2u(v); };
  auto stu = [&](uint32_t f, uint32_t p, uint32_t v) { S[f * PW + p] = v; };

  const unsigned long long below = (1ull << lane) - 1ull;
  uint32_t nT = 0u, nS = 0u, nR = PW;                               // queue sizes (wave-uniform)
  uint32_t chunkNext = 0u, chunkEnd = 0u; bool exhausted = false;   // claimed work items not handed out yet (wave-uniform)
  uint32_t nSeg = 0u;
  TraceCounters tc{0u, 0u};
  uint2 overflow[1];
  RayTrav R; trav_init(R, v3(0.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, 1.0f), 0.0f, 0.0f);
  bool alive = false; uint32_t myPath = 0u, myRng = 0u;             // the ray this lane is walking belongs to path myPath

  for (;;) {
    __atomic_signal_fence(__ATOMIC_SEQ_CST); // (compiler only) path state and queues are exchanged between the lanes of this wave through LDS
    const unsigned long long aliveMask = __ballot(alive);
    const uint32_t nAlive = (uint32_t)__popcll(aliveMask);
    if (nAlive == 0u && nT == 0u && nS == 0u && nR == 0u) break;
    // traversal running dry: feed it from whatever the other stages hold, full wave or not
    const bool dry = nAlive + nT < thrDry;
    if (nS >= (dry ? 1u : thrShade)) {
      // --- SHADE (rp_main.chit + rp_main.rgen:441-480)
      const uint32_t take = nS < 64u ? nS : 64u;
      const bool act = lane < take;
      const uint32_t p = act ? qS[nS - 1u - lane] : 0u;
      nS -= take;
      bool cont = false;
      if (act) {
        ShadeIO io; io.shadow = false; io.shadowFirst = false; io.cont = false;
        io.throughput = v3(ldf(F_THR, p), ldf(F_THR + 1u, p), ldf(F_THR + 2u, p)); io.radiance = v3(ldf(F_RAD, p), ldf(F_RAD + 1u, p), ldf(F_RAD + 2u, p));
        io.bitfield = ldu(F_BITS, p); io.rng = ldu(F_RNG, p);
        const F4 h = F4{ldf(F_RO, p), ldf(F_RO + 1u, p), ldf(F_RO + 2u, p), ldf(F_TMIN, p)};
        const F4 rd = F4{ldf(F_RD, p), ldf(F_RD + 1u, p), ldf(F_RD + 2u, p), 0.0f};
        shade_segment<KLASS, TEXTURED, false, false>(U, sc, nullptr, h, rd, io);
        stf(F_THR, p, io.throughput.x); stf(F_THR + 1u, p, io.throughput.y); stf(F_THR + 2u, p, io.throughput.z);
        stf(F_RAD, p, io.radiance.x); stf(F_RAD + 1u, p, io.radiance.y); stf(F_RAD + 2u, p, io.radiance.z);
        stu(F_BITS, p, io.bitfield); stu(F_RNG, p, io.rng);
        cont = io.cont;
        if (cont) {
          stf(F_RO, p, io.no.x); stf(F_RO + 1u, p, io.no.y); stf(F_RO + 2u, p, io.no.z);
          stf(F_RD, p, io.k2.x); stf(F_RD + 1u, p, io.k2.y); stf(F_RD + 2u, p, io.k2.z);
          stf(F_TMIN, p, 0.0f); stf(F_TMAX, p, io.tMaxNext);
        }
      }
      const unsigned long long mT = __ballot(act && cont), mR = __ballot(act && !cont);
      if (act && cont) qT[nT + (uint32_t)__popcll(mT & below)] = p;
      if (act && !cont) qR[nR + (uint32_t)__popcll(mR & below)] = p;
      nT += (uint32_t)__popcll(mT); nR += (uint32_t)__popcll(mR);
      continue;
    }
    if (nR >= (dry ? 1u : thrRegen)) {
      // --- REGEN: per-sample finish (rp_main.rgen:483-496) and the next work item's camera ray (:213-283)
      const uint32_t take = nR < 64u ? nR : 64u;
      const bool act = lane < take;
      const uint32_t p = act ? qR[nR - 1u - lane] : 0u;
      nR -= take;
      if (act) {
        const uint32_t w = ldu(F_WORK, p);
        if (w != NO_WORK) {
          const uint32_t pixelLocal = w % U.pixelCount, sLocal = w / U.pixelCount;
          const uint32_t bounces = ldu(F_BITS, p) & 0x00000fffu;
          if (st.bouncesAov && U.batchFirstSample + sLocal == U.spp - 1u) { // Bounces AOV: the pixel's last sample (:483-486)
            const uint32_t maxB = U.maxBounces < 0x00000fffu ? U.maxBounces : 0x00000fffu;
            const V3 c = gi_colormap_inferno((float)bounces / (float)maxB);
            F4* dst = &st.bouncesAov[tile_to_image_pixel(U, pixelLocal)];
            dst->x = c.x; dst->y = c.y; dst->z = c.z;
          }
          if (st.pathSegments) atomicAdd(&st.pathSegments[pixelLocal], bounces); // ClockCycles proxy: integer sum, order-free
          const V3 c = finish_sample(U, v3(ldf(F_RAD, p), ldf(F_RAD + 1u, p), ldf(F_RAD + 2u, p)));
          st4(&sampleBuf[(size_t)sLocal * U.pixelCount + pixelLocal], c.x, c.y, c.z, 0.0f);
        }
      }
      // hand out work items: lane i of the batch gets the i-th of the next `take` unclaimed items (claims of `chunk` consecutive ids)
      uint32_t myWork = NO_WORK, handed = 0u;
      while (handed < take && !exhausted) {
        if (chunkNext == chunkEnd) {
          uint32_t b = 0u;
          if (lane == 0u) b = atomicAdd(&cnt->cursor[0][0].v, chunk);
          b = (uint32_t)__shfl((int)b, 0);
          if (b >= U.workTotal) { exhausted = true; break; }
          chunkNext = b; chunkEnd = (U.workTotal - b) < chunk ? U.workTotal : b + chunk;
        }
        const uint32_t avail = chunkEnd - chunkNext, n = (take - handed) < avail ? (take - handed) : avail;
        if (lane >= handed && lane < handed + n) myWork = chunkNext + (lane - handed);
        chunkNext += n; handed += n;
      }
      const bool go = act && myWork != NO_WORK;
      if (go) {
        const uint32_t pixelLocal = myWork % U.pixelCount, sLocal = myWork / U.pixelCount;
        const uint32_t pixelIndex = tile_to_image_pixel(U, pixelLocal); // :195 (global index: the RNG is tile independent)
        V3 ro, rdv; float tMin, tMax; uint32_t rng;
        make_camera_ray(U, pixelIndex, U.sampleOffset + U.batchFirstSample + sLocal, ro, rdv, tMin, tMax, rng);
        // :274-276
        stf(F_THR, p, 1.0f); stf(F_THR + 1u, p, 1.0f); stf(F_THR + 2u, p, 1.0f); stf(F_RAD, p, 0.0f); stf(F_RAD + 1u, p, 0.0f); stf(F_RAD + 2u, p, 0.0f);
        stu(F_BITS, p, 0u); stu(F_RNG, p, rng); stu(F_WORK, p, myWork);
        stf(F_RO, p, ro.x); stf(F_RO + 1u, p, ro.y); stf(F_RO + 2u, p, ro.z); stf(F_RD, p, rdv.x); stf(F_RD + 1u, p, rdv.y); stf(F_RD + 2u, p, rdv.z);
        stf(F_TMIN, p, tMin); stf(F_TMAX, p, tMax);
      }
      const unsigned long long mT = __ballot(go);
      if (go) qT[nT + (uint32_t)__popcll(mT & below)] = p;
      nT += (uint32_t)__popcll(mT); // (paths that found no work left are dead: in no queue)
      continue;
    }
    // --- TRAV: idle lanes take queued rays, then every walking ray advances one step (traceRayEXT, rp_main.rgen:381-393)
    const uint32_t nIdle = 64u - nAlive;
    if (nT > 0u && (nIdle >= 8u || nAlive == 0u)) {
      const uint32_t take = nIdle < nT ? nIdle : nT;
      const uint32_t rank = (uint32_t)__popcll(~aliveMask & below);
      if (!alive && rank < take) {
        const uint32_t p = qT[nT - 1u - rank];
        myPath = p; if (CUTOUT) myRng = ldu(F_RNG, p);
        trav_init(R, v3(ldf(F_RO, p), ldf(F_RO + 1u, p), ldf(F_RO + 2u, p)), v3(ldf(F_RD, p), ldf(F_RD + 1u, p), ldf(F_RD + 2u, p)), ldf(F_TMIN, p),
            ldf(F_TMAX, p));
        wave_ray_begin(W, R.tBest);
        alive = true;
      }
      nT -= take;
    }
    if (!__ballot(alive)) continue;
    const bool done = wave_step<false, COUNT, STACK, false, true, CUTOUT>(R, alive, W, sc, s_nodes, ldsNodes, s_tris, ldsTris, s_stack, overflow, tc, myRng);
    const bool fin = alive && done;
    bool hit = false;
    if (fin) {
      alive = false; nSeg++;
      wave_ray_end(W, R);
      const uint32_t p = myPath;
      hit = R.found;
      if (hit) { stf(F_RO, p, R.tBest); stf(F_RO + 1u, p, R.bestU); stf(F_RO + 2u, p, R.bestV); stu(F_TMIN, p, R.bestTri); }
      else { // rp_main.miss:68-86: uniform fallback dome == colour clear value; the loop's bounce++ still happens (rp_main.rgen:480)
        const V3 thr = v3(ldf(F_THR, p), ldf(F_THR + 1u, p), ldf(F_THR + 2u, p)), bg = v3(U.background);
        const V3 rad = v3(ldf(F_RAD, p), ldf(F_RAD + 1u, p), ldf(F_RAD + 2u, p)) + thr * bg;
        stf(F_RAD, p, rad.x); stf(F_RAD + 1u, p, rad.y); stf(F_RAD + 2u, p, rad.z);
        stu(F_BITS, p, ldu(F_BITS, p) + 1u);
      }
    }
    const unsigned long long mS = __ballot(fin && hit), mR = __ballot(fin && !hit);
    if (fin && hit) qS[nS + (uint32_t)__popcll(mS & below)] = myPath;
    if (fin && !hit) qR[nR + (uint32_t)__popcll(mR & below)] = myPath;
    nS += (uint32_t)__popcll(mS); nR += (uint32_t)__popcll(mR);
  }

  // statistics: one atomic per wave and counter
  unsigned long long a = nSeg, c = tc.nodes, d = tc.tris;
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); if (COUNT) { c += __shfl_down(c, off); d += __shfl_down(d, off); } }
  if (lane == 0u) {
    atomicAdd(&cnt->segments, a);
    if (COUNT) { atomicAdd(&cnt->nodesVisited, c); atomicAdd(&cnt->trisTested, d); }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
using PathBwKernel = void (*)(FrameUniforms, SceneView, PathState, Counters*, F4*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t);
template <uint32_t STACK>
static PathBwKernel pickPathBwKernel(uint32_t classMask, bool textured, bool cutout, bool count)
{
  const bool single = classMask == 1u || classMask == 2u || classMask == 4u;
  if (single && !textured && !cutout && !count) {
    if (classMask == 1u) return k_path_bw<0u, false, false, false, STACK>;
    if (classMask == 2u) return k_path_bw<1u, false, false, false, STACK>;
    return k_path_bw<2u, false, false, false, STACK>;
  }
  return count ? k_path_bw<KLASS_DYNAMIC, true, true, true, STACK> : k_path_bw<KLASS_DYNAMIC, true, true, false, STACK>;
}

int launchPathBw(hipStream_t s, uint32_t cuCount, uint32_t classMask, bool textured, bool count, uint32_t chunk, const FrameUniforms& U, const SceneView& sc,
                 const PathState& st, Counters* cnt, F4* sampleBuf)
{
  const uint32_t PW = PATH_BW_PATHS_DEFAULT;
  const uint32_t thrS = 48u, thrR = 32u, thrDry = 16u; // shade / regenerate / give-up thresholds (lanes); measured r02-r03
  const uint32_t ldsNodes = sc.nodeCount, ldsTris = sc.triCount;
  const uint32_t stack = sc.bvhDepth <= 4u ? 4u : 8u;
  const uint32_t bytes = stack * TRACE_BLOCK * (uint32_t)sizeof(uint2) + ldsNodes * 80u + ldsTris * 48u + (TRACE_BLOCK / 64u) * (F_COUNT + 3u) * PW * 4u;
  PathBwKernel k = stack == 4u
      ? pickPathBwKernel<4u>(classMask, textured, sc.hasCutouts != 0u, count) : pickPathBwKernel<8u>(classMask, textured, sc.hasCutouts != 0u, count);
  int perCu = 2;
  hipFuncAttributes fa{};
  if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k)) == hipSuccess && fa.numRegs > 0) {
    const uint32_t regs = ((uint32_t)fa.numRegs + 7u) & ~7u, byRegs = 512u / regs;
    const uint32_t byLds = (160u * 1024u) / (bytes + (uint32_t)fa.sharedSizeBytes + 256u);
    perCu = (int)(byRegs < byLds ? byRegs : byLds);
    if (perCu > 8) perCu = 8;
    if (perCu < 1) perCu = 1;
  }
  if (bytes + 8192u > 64u * 1024u) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  // persistent grid: what is resident, but never more waves than there are PW-sized shares of the work
  const uint64_t shares = ((uint64_t)U.workTotal + PW - 1u) / PW;
  uint64_t blocks = (uint64_t)cuCount * (uint64_t)perCu;
  const uint64_t needed = (shares + (TRACE_BLOCK / 64u) - 1u) / (TRACE_BLOCK / 64u);
  if (blocks > needed) blocks = needed;
  if (blocks == 0u) blocks = 1u;
  hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(TRACE_BLOCK), bytes, s, U, sc, st, cnt, sampleBuf, ldsNodes, ldsTris, chunk, PW, thrS, thrR, thrDry);
  return perCu;
}

} // namespace gi
