// gi_path.hip -- k_path: the fused persistent path kernel for scenes whose whole BVH lives in LDS (cornell: 5 nodes, 46 triangles).
//
// The wavefront pipeline (gi_kernels.hip) streams every path through HBM once per stage and bounce: ray record out, hit record in,
// 64-byte slot gathered and scattered -- on config C2 1.6 TB per frame for a scene of 2.6 KB.  When the scene is LDS-resident none
// of that traffic buys anything: there is no memory latency to hide behind a queue and no incoherent fetch to sort for.  k_path
// keeps the PATH in registers instead.  Every wave is persistent; a lane carries one (pixel, sample) work item through
// camera ray -> [closest hit -> shade -> shadow ray]* and, the moment its path ends, writes the finished sample and takes the next
// work item, so all 64 lanes trace on every trip ("persistent threads with path regeneration"; the regeneration replaces the
// wavefront loop's compaction).  HBM sees one 16-byte record per SAMPLE (per-sample colour buffer, summed in sample order by
// k_accumulate exactly as before) instead of ~340 bytes per SEGMENT.
//
// Replaces, for such scenes, the same reference code as the stage kernels: rp_main.rgen:185-521 (whole loop), traceRayEXT
// (:381-393, 412-424), rp_main.chit, rp_main.miss:68-86, rp_main_shadow.miss.  All per-path arithmetic is the SHARED stage code
// (make_camera_ray, wave_step, shade_segment, finish_sample), so images are bit-identical to the wavefront pipeline's and the
// oracle's; work items are claimed in chunks of consecutive ids (sample-major: adjacent pixels of one sample index), one atomic per
// chunk on one cursor.
//
// Not handled here (the host falls back to the wavefront pipeline): medium stacks (mediumStackSize > 0), dome-light images,
// scenes beyond LDS, trees deeper than 8 levels.

#include <hip/hip_runtime.h>

#define GI_LEAN_SQRT 1 // gi_device_math.h gi_sqrt: the correctly rounded square root without the steps ordinary arguments do not need
#include "gi_device_math.h"
#include "gi_kernels.h"
#include "gi_types.h"
#include "gi_queues.h"
#include "gi_traversal.h"
#include "gi_shading.h"
#include "gi_stages.h"

namespace gi {

constexpr uint32_t PRE_FIELDS = 10; // prepared camera ray: origin, direction, tMin, tMax, rng state, work item
constexpr uint32_t PATH_STACK_MAX = 8; // LDS traversal-stack entries per lane: 4 for trees of depth <= 4 (cornell), else 8 (the host checks bvhDepth <= 8)

constexpr int PATH_WAVES = 4; // resident waves per SIMD the register allocation aims for (114 VGPRs without a hint; 3 cost 11 %, 5 spill 26 registers: r03)
template <uint32_t KLASS, bool TEXTURED, bool NEE, bool CUTOUT, bool COUNT, uint32_t PATH_STACK>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(PATH_WAVES, 8))) void k_path(FrameUniforms U, SceneView sc, PathState st,
    Counters* cnt, F4* __restrict__ sampleBuf,
                                                      uint32_t ldsNodes, uint32_t ldsTris, uint32_t chunk)
{
  extern __shared__ uint4 s_dyn[];
  uint2 (*s_stack)[TRACE_BLOCK] = reinterpret_cast<uint2 (*)[TRACE_BLOCK]>(s_dyn);
  uint4* s_nodes = s_dyn + (PATH_STACK * TRACE_BLOCK * sizeof(uint2)) / sizeof(uint4);
  uint4* s_tris = s_nodes + ldsNodes * 5u;
  __shared__ WaveTri s_wave[TRACE_BLOCK / 64];
  WaveTri& W = s_wave[threadIdx.x >> 6];
  for (uint32_t i = threadIdx.x; i < ldsNodes * 5u; i += TRACE_BLOCK) s_nodes[i] = reinterpret_cast<const uint4*>(sc.nodes)[i];
  for (uint32_t i = threadIdx.x; i < ldsTris * 3u; i += TRACE_BLOCK) s_tris[i] = reinterpret_cast<const uint4*>(sc.tris)[(i / 3u) * 4u + (i % 3u)];
  __syncthreads(); // the only barrier: from here on the waves of a block are independent

  const uint32_t lane = __lane_id();
  const unsigned long long below = (1ull << lane) - 1ull;
  // the path this lane carries (rp_main_payload.glsl:20-33) and its next ray
  V3 thr = v3(0.0f, 0.0f, 0.0f), rad = thr, ro = thr, rdv = v3(0.0f, 0.0f, 1.0f);
  float tMin = 0.0f, tMax = 0.0f;
  uint32_t bitfield = 0u, rng = 0u, pixelLocal = 0u, sLocal = 0u;
  bool alive = false;
  uint32_t chunkNext = 0u, chunkEnd = 0u; bool exhausted = false; // wave-uniform: the claimed work items not handed out yet
  uint32_t nSeg = 0u, nShadow = 0u;
  TraceCounters tc{0u, 0u}, tcs{0u, 0u};
  uint2 overflow[1];
  RayTrav R;
  // Camera rays are generated 64 at a time, by ALL lanes, into a per-wave LDS ring (r03): a trip regenerates only the ~45 % of the lanes whose path just ended,
  // and make_camera_ray (hash, two draws, the Gaussian filter's log / sqrt / sincos, normalise) then ran at that lane utilisation on every trip.  Now the wave
  // prepares the next 64 work items' rays whenever fewer than 64 are pending (one full-width pass every ~2 trips) and idle lanes just pop them.
  __shared__ uint32_t s_pre[TRACE_BLOCK / 64][PRE_FIELDS][128]; // ring of 128 prepared rays per wave
  GI_LDS uint32_t (*pre)[128] = (GI_LDS uint32_t (*)[128])&s_pre[threadIdx.x >> 6][0][0];
  uint32_t preHead = 0u, preTail = 0u; // wave-uniform ring positions (monotonic; slot = position & 127)

  unsigned long long pc[4] = {0ull, 0ull, 0ull, 0ull}, pl[4] = {0ull, 0ull, 0ull, 0ull}, trips = 0ull, tPrev = COUNT ? __builtin_readcyclecounter() : 0ull;
  auto phase = [&](int k,
      unsigned long long lanes) { if (COUNT) { const unsigned long long t = __builtin_readcyclecounter(); pc[k] += t - tPrev; tPrev = t; pl[k] += lanes; } };
  for (;;) {
    __atomic_signal_fence(__ATOMIC_SEQ_CST); // (compiler only) the ring is exchanged between the lanes of this wave through LDS
    // --- regeneration (rp_main.rgen:213-283): idle lanes take the next work items w = sample * P + pixel
    unsigned long long idle = __ballot(!alive);
    const uint32_t nIdle = (uint32_t)__popcll(idle);
    if (nIdle) {
      // top the ring up to at least 64 prepared rays (or whatever work is left): every lane prepares one
      while (preTail - preHead < 64u && !exhausted) {
        if (chunkNext == chunkEnd) {
          uint32_t b = 0u;
          if (lane == 0u) b = atomicAdd(&cnt->cursor[0][0].v, chunk);
          b = (uint32_t)__shfl((int)b, 0);
          if (b >= U.workTotal) { exhausted = true; break; }
          chunkNext = b; chunkEnd = (U.workTotal - b) < chunk ? U.workTotal : b + chunk;
        }
        const uint32_t avail = chunkEnd - chunkNext, take = avail < 64u ? avail : 64u;
        if (lane < take) {
          const uint32_t w = chunkNext + lane;
          const uint32_t pl = w % U.pixelCount, sl = w / U.pixelCount;
          V3 o, d; float t0, t1; uint32_t r;
          // :195 (global pixel index: the RNG is tile independent)
          make_camera_ray(U, tile_to_image_pixel(U, pl), U.sampleOffset + U.batchFirstSample + sl, o, d, t0, t1, r);
          const uint32_t slot = (preTail + lane) & 127u;
          pre[0][slot] = f2u(o.x); pre[1][slot] = f2u(o.y); pre[2][slot] = f2u(o.z); pre[3][slot] = f2u(d.x); pre[4][slot] = f2u(d.y); pre[5][slot] = f2u(d.z);
          pre[6][slot] = f2u(t0); pre[7][slot] = f2u(t1); pre[8][slot] = r; pre[9][slot] = w;
        }
        chunkNext += take; preTail += take;
      }
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
      const uint32_t have = preTail - preHead, take = nIdle < have ? nIdle : have;
      const uint32_t rank = (uint32_t)__popcll(idle & below);
      if (!alive && rank < take) {
        const uint32_t slot = (preHead + rank) & 127u;
        ro = v3(u2f(pre[0][slot]), u2f(pre[1][slot]), u2f(pre[2][slot])); rdv = v3(u2f(pre[3][slot]), u2f(pre[4][slot]), u2f(pre[5][slot]));
        tMin = u2f(pre[6][slot]); tMax = u2f(pre[7][slot]); rng = pre[8][slot];
        const uint32_t w = pre[9][slot];
        pixelLocal = w % U.pixelCount; sLocal = w / U.pixelCount;
        thr = v3(1.0f, 1.0f, 1.0f); rad = v3(0.0f, 0.0f, 0.0f); bitfield = 0u; // :274-276
        alive = true;
      }
      preHead += take;
    }
    if (!__ballot(alive)) break;
    phase(0, nIdle); trips++;

    // --- closest hit (traceRayEXT, rp_main.rgen:381-393): all rays of the wave advance in steps, triangles are tested cooperatively
    trav_init(R, ro, rdv, tMin, alive ? tMax : 0.0f);
    wave_ray_begin(W, R.tBest);
    bool tAlive = alive;
    while (__ballot(tAlive)) {
      if (wave_step<false, COUNT, PATH_STACK, false, true,
          CUTOUT>(R, tAlive, W, sc, s_nodes, ldsNodes, s_tris, ldsTris, s_stack, overflow, tc, rng)) tAlive = false;
    }
    bool ended = false, missed = false;
    ShadeIO io; io.shadow = false; io.shadowFirst = false; io.cont = false;
    phase(1, (unsigned long long)__popcll(__ballot(alive)));
    if (alive) {
      nSeg++;
      wave_ray_end(W, R);
      if (R.found) { // rp_main.chit + rp_main.rgen:397-480
        io.throughput = thr; io.radiance = rad; io.bitfield = bitfield; io.rng = rng;
        const F4 h = F4{R.tBest, R.bestU, R.bestV, u2f(R.bestTri)}, rd = F4{rdv.x, rdv.y, rdv.z, 0.0f};
        shade_segment<KLASS, TEXTURED, false, NEE>(U, sc, nullptr, h, rd, io);
        thr = io.throughput; rad = io.radiance; bitfield = io.bitfield; rng = io.rng;
        // untraced shadow ray == "not shadowed" (rp_main.rgen:431-435)
        if (NEE && st.neeKey && io.shadowFirst && !io.shadow) nee_aov_record_px(st, pixelLocal, sLocal, false);
        ended = !io.cont;
      } else { // rp_main.miss:68-86: uniform fallback dome == colour clear value; the loop's bounce++ still happens (rp_main.rgen:480)
        rad = rad + thr * v3(U.background);
        if (st.neeKey && (bitfield & 0x00000fffu) == 0u) nee_aov_record_px(st, pixelLocal, sLocal, false);
        bitfield++;
        ended = true; missed = true;
      }
    }
    (void)missed;
    phase(2, (unsigned long long)__popcll(__ballot(alive && R.found)));

    // --- shadow ray of this bounce (rp_main.rgen:397-429): origin = next ray origin, tMin 0.01, tMax = distance to the light sample
    if (NEE) {
      bool sAlive = alive && io.shadow;
      if (__ballot(sAlive)) {
        trav_init(R, io.no, io.sdir, 0.01f, sAlive ? io.ld : 0.0f);
        wave_ray_begin(W, R.tBest);
        const bool traced = sAlive;
        while (__ballot(sAlive)) {
          if (wave_step<true, COUNT, PATH_STACK, false, true,
              CUTOUT>(R, sAlive, W, sc, s_nodes, ldsNodes, s_tris, ldsTris, s_stack, overflow, tcs, io.rngShadow)) sAlive = false;
        }
        if (traced) {
          nShadow++;
          if (!R.found) rad = rad + io.nee;
          if (st.neeKey && io.shadowFirst) nee_aov_record_px(st, pixelLocal, sLocal, R.found);
        }
      }
    }

    // --- next segment, or the per-sample finish (rp_main.rgen:483-496) -> per-sample colour buffer
    if (alive) {
      if (!ended) { ro = io.no; rdv = io.k2; tMin = 0.0f; tMax = io.tMaxNext; }
      else {
        const uint32_t bounces = bitfield & 0x00000fffu;
        if (st.bouncesAov && U.batchFirstSample + sLocal == U.spp - 1u) { // Bounces AOV: the pixel's last sample (:483-486)
          const uint32_t maxB = U.maxBounces < 0x00000fffu ? U.maxBounces : 0x00000fffu;
          const V3 c = gi_colormap_inferno((float)bounces / (float)maxB);
          F4* dst = &st.bouncesAov[tile_to_image_pixel(U, pixelLocal)];
          dst->x = c.x; dst->y = c.y; dst->z = c.z;
        }
        if (st.pathSegments) atomicAdd(&st.pathSegments[pixelLocal], bounces); // ClockCycles proxy: integer sum, order-free
        const V3 c = finish_sample(U, rad);
        st4(&sampleBuf[(size_t)sLocal * U.pixelCount + pixelLocal], c.x, c.y, c.z, 0.0f);
        alive = false;
      }
    }
    phase(3, (unsigned long long)__popcll(__ballot(ended)));
  }

  if (COUNT
      && lane == 0u) { for (int k = 0; k < 4; k++) { atomicAdd(&cnt->phaseCycles[k], pc[k]); atomicAdd(&cnt->phaseLanes[k], pl[k]);
      } atomicAdd(&cnt->phaseTrips, trips); }
  // statistics: one atomic per wave and counter
  unsigned long long a = nSeg, b = nShadow, c = tc.nodes, d = tc.tris, e = tcs.nodes, f = tcs.tris;
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off); b += __shfl_down(b, off);
    if (COUNT) { c += __shfl_down(c, off); d += __shfl_down(d, off); e += __shfl_down(e, off); f += __shfl_down(f, off); }
  }
  if (lane == 0u) {
    atomicAdd(&cnt->segments, a);
    if (NEE) atomicAdd(&cnt->shadowRays, b);
    if (COUNT) { atomicAdd(&cnt->nodesVisited, c); atomicAdd(&cnt->trisTested, d); atomicAdd(&cnt->shadowNodesVisited, e); atomicAdd(&cnt->shadowTrisTested, f);
        }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool pathKernelSupports(const SceneView& sc)
{
  return sc.triCount > 0u && sc.nodeCount <= LDS_NODES && sc.triCount <= LDS_TRIS && sc.bvhDepth <= PATH_STACK_MAX && sc.mediumStackSize == 0u
      && sc.domeTexture == 0u && !sc.twoLevel
         // (a scene whose shading records are packed -- never an LDS-resident one today:
         // TriRec::vi[0] is a shading-record index there, the fused kernels read vertex indices)
         && !sc.shadePacked;
}

using PathKernel = void (*)(FrameUniforms, SceneView, PathState, Counters*, F4*, uint32_t, uint32_t, uint32_t);
// Hot variants: one material class, no textures, no cutouts, no counters (the C1 / C2 paths).  Everything else runs the general
// variant (class read from the material record, textures and cutouts compiled in).
template <uint32_t STACK>
static PathKernel pickPathKernel(uint32_t classMask, bool textured, bool nee, bool cutout, bool count)
{
  const bool single = classMask == 1u || classMask == 2u || classMask == 4u;
  if (single && !textured && !cutout && !count) {
    if (classMask == 1u) return nee ? k_path<0u, false, true, false, false, STACK> : k_path<0u, false, false, false, false, STACK>;
    if (classMask == 2u) return nee ? k_path<1u, false, true, false, false, STACK> : k_path<1u, false, false, false, false, STACK>;
    return nee ? k_path<2u, false, true, false, false, STACK> : k_path<2u, false, false, false, false, STACK>;
  }
  if (count) return nee ? k_path<KLASS_DYNAMIC, true, true, true, true, STACK> : k_path<KLASS_DYNAMIC, true, false, true, true, STACK>;
  return nee ? k_path<KLASS_DYNAMIC, true, true, true, false, STACK> : k_path<KLASS_DYNAMIC, true, false, true, false, STACK>;
}

int launchPath(hipStream_t s, uint32_t cuCount, uint32_t classMask, bool textured, bool count, uint32_t chunk, const FrameUniforms& U, const SceneView& sc,
               const PathState& st, Counters* cnt, F4* sampleBuf)
{
  const uint32_t ldsNodes = sc.nodeCount, ldsTris = sc.triCount;
  const uint32_t stack = sc.bvhDepth <= 4u ? 4u : 8u;
  const uint32_t bytes = stack * TRACE_BLOCK * (uint32_t)sizeof(uint2) + ldsNodes * 80u + ldsTris * 48u;
  const bool neeOn = (U.flags & FLAG_NEE) != 0u;
  PathKernel k = stack == 4u
      ? pickPathKernel<4u>(classMask, textured, neeOn, sc.hasCutouts != 0u, count) : pickPathKernel<8u>(classMask, textured, neeOn, sc.hasCutouts != 0u, count);
  // Resident blocks per CU (4 waves per block = 1 wave per SIMD and block): the smaller of what the 160 KiB of LDS and the 512-entry
  // register file of a SIMD hold.  (hipOccupancyMaxActiveBlocksPerMultiprocessor answered 3 for a 35 KiB block: it does not know gfx950's LDS size.)
  int perCu = 2;
  hipFuncAttributes fa{};
  if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k)) == hipSuccess && fa.numRegs > 0) {
    const uint32_t regs = ((uint32_t)fa.numRegs + 7u) & ~7u, byRegs = 512u / regs;
    const uint32_t byLds = (160u * 1024u) / (bytes + (uint32_t)fa.sharedSizeBytes + 256u);
    perCu = (int)(byRegs < byLds ? byRegs : byLds);
    if (perCu > 8) perCu = 8;
    if (perCu < 1) perCu = 1;
  }
  // persistent grid: what is resident, but never more waves than chunks of work
  const uint64_t chunks = ((uint64_t)U.workTotal + chunk - 1u) / chunk;
  uint64_t blocks = (uint64_t)cuCount * (uint64_t)perCu;
  const uint64_t needed = (chunks + (TRACE_BLOCK / 64u) - 1u) / (TRACE_BLOCK / 64u);
  if (blocks > needed) blocks = needed;
  if (blocks == 0u) blocks = 1u;
  hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(TRACE_BLOCK), bytes, s, U, sc, st, cnt, sampleBuf, ldsNodes, ldsTris, chunk);
  return perCu;
}

// k_debug_sqrt: gi_sqrt (gi_device_math.h) against sqrtf over a range of bit patterns; counts the arguments whose results differ in a bit (NaN against NaN is equal
// whatever the payload: both come out of v_sqrt_f32 here, but the contract does not say so)
__global__ void k_debug_sqrt(uint32_t first, unsigned long long count, unsigned long long* mismatches)
{
  unsigned long long bad = 0ull;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = u2f(first + (uint32_t)i);
    const float a = gi_sqrt(x), b = sqrtf(x);
    if (f2u(a) != f2u(b) && !(a != a && b != b)) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
}

void launchDebugSqrt(hipStream_t s, uint32_t first, unsigned long long count, unsigned long long* mismatches)
{
  hipLaunchKernelGGL(k_debug_sqrt, dim3(4096), dim3(256), 0, s, first, count, mismatches);
}

} // namespace gi
