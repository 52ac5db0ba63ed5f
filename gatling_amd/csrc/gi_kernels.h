// gi_kernels.h -- host-side launch interface of the stage kernels (gi_kernels.hip,
// gi_trace.hip, gi_shade.hip, gi_aov.hip) and of the fused ones (gi_path.hip, gi_path_bw.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "gi_types.h"

namespace gi {

void launchInit(hipStream_t s, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t n, bool resetStats);
// `par` = iteration parity: k_raygen reads REGEN[par] and appends TRACE[par]; k_trace reads TRACE[par] and appends HIT and
// REGEN[par^1]; k_shade reads HIT and appends TRACE[par^1], REGEN[par^1], SHADOW.
void launchRaygen(hipStream_t s, uint32_t blocks, const FrameUniforms& U, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t par, F4* sampleBuf);
void launchAccumulate(hipStream_t s, const FrameUniforms& U, const F4* sampleBuf, F4* accum, F4* colorOut, bool firstBatch, bool lastBatch);
// LDS bytes one k_trace block needs for this scene (stack + staged nodes + staged triangles)
uint32_t traceStaticLdsBytes(); // static LDS of the traversal kernels on top of traceLdsLayout's dynamic bytes
void traceLdsLayout(const SceneView& sc, uint32_t& ldsNodes, uint32_t& ldsTris, uint32_t& bytes);
void launchTrace(hipStream_t s, uint32_t blocks, bool anyHit, bool count, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt,
                 uint32_t qIn, uint32_t qMiss, uint32_t dynRefill, uint32_t routeBlocks, const FrameUniforms& U, F4* sampleBuf);
void launchRoute(hipStream_t s, uint32_t blocks, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t qIn, uint32_t qMiss,
    const FrameUniforms& U,
                 F4* sampleBuf); // (called by launchTrace behind a k_trace_dyn launch)
// most records a thread appends per trip of a streaming kernel (k_route ROUTE_ITEMS,
// k_raygen RAYGEN_ITEMS): sizes the queue shards' slack, gi_render.cpp shardCapacity
constexpr uint32_t APPEND_ITEMS_MAX = 4u;
// flag in dynRefill (shadow launches): children are visited in slot order instead of near-to-far (k_trace_dyn: DYN_SLOT_ORDER)
constexpr uint32_t TRACE_DYN_SLOT_ORDER = 0x200u;
constexpr uint32_t TRACE_DYN_SPILL8 = 0x100u; // flag in dynRefill: 8 LDS stack entries + scratch overflow instead of 16 LDS entries
// dynRefill: 0 = block-synchronous k_trace; N = scenes that do not fit LDS use k_trace_dyn (a wave refills once N lanes are idle) + k_route
// one launch per material class present in the scene (the class is the sort key between k_trace and k_shade)
void launchShade(hipStream_t s, uint32_t blocks, uint32_t klass, bool textured /* some material of the class has textured inputs */,
    bool volume /* mediumStackSize > 0 */, const FrameUniforms& U, const SceneView& sc, const PathState& st, const QueueSet& qs, Counters* cnt, uint32_t par);

// Fused persistent path kernel (gi_path.hip) for LDS-resident scenes; launchPath returns the resident blocks per CU it launched with
bool pathKernelSupports(const SceneView& sc);
// the same kernel as a wave-local wavefront (gi_path_bw.hip): path state and stage queues in LDS, every stage runs on full waves; NEE off only
int launchPathBw(hipStream_t s, uint32_t cuCount, uint32_t classMask, bool textured, bool count, uint32_t chunk, const FrameUniforms& U, const SceneView& sc,
                 const PathState& st, Counters* cnt, F4* sampleBuf);
int launchPath(hipStream_t s, uint32_t cuCount, uint32_t classMask, bool textured, bool count, uint32_t chunk, const FrameUniforms& U, const SceneView& sc,
               const PathState& st, Counters* cnt, F4* sampleBuf);

void launchAov(hipStream_t s, const FrameUniforms& U, const SceneView& sc, const AovTargets& A);
void launchResolveNee(hipStream_t s, const FrameUniforms& U, const unsigned long long* key, F4* aov, uint32_t pixelCount);
void launchZeroClosest(hipStream_t s, Counters* cnt, uint32_t par); // FLAG_TWO_STREAM: in front of every closest-hit launch
void launchSpin(hipStream_t s, unsigned long long ns);              // test hook: occupies a stream for ~ns nanoseconds
void launchDebugBsdf(hipStream_t s, const MaterialRec* mat, uint32_t shadeClass, uint32_t count, const float* in, float* out);
void launchDebugSqrt(hipStream_t s, uint32_t first, unsigned long long count, unsigned long long* mismatches); // gi_sqrt against sqrtf over bit patterns
void launchDebugTex(hipStream_t s, const float* texels, uint32_t w, uint32_t h, uint32_t d, uint32_t count, const float* queries, float* out);

} // namespace gi
