// gi_options.h -- the library's environment interface in one place.
//
// Environment variables (all optional):
//   GATLING_DEVICE         HIP device ordinal for gtl::giInitialize (the reference picks its Vulkan device itself, CgpuVk.cpp:892-909)          [gtl_shim.cpp]
//   GATLING_DEVICES        "0,1,2,3" or "all": one process drives several devices inside the library                     [gi_c.cpp]
//   GATLING_BUILD_THREADS  host threads of the BVH build (default: all, at most 32; the tree does not depend on it)                      [bvh8.cpp, gi_c.cpp]
//   GATLING_BUILD_TIMING   log scene-build / transform-update timings to stderr
//   GATLING_ITER_LOG       with kernel timers on every iteration: one line per bounce iteration (queue sizes, stage times) to stderr
//   GATLING_OPTIONS        "key=value,key=value": the debug / test switches below.  None changes an image (tests hold every one of them to the oracle
//                          bit for bit); they select between equivalent schedules, or pin sizes the library otherwise plans itself.
//
//   key                  default   meaning
//   trace_dyn            8         refill threshold of k_trace_dyn; 0 = the block-synchronous k_trace for scenes beyond LDS too
//                                  (same as GI_C_SCENE_OPTION_TRACE_DYNAMIC)
//   trace_dyn_spill8     0         trees deeper than 8 levels keep 8 stack entries in LDS and spill the rest to scratch
//   two_level            -1        -1 = automatic (from 2^26 flattened triangles), 0 / 1 = force the flat / the two-level layout
//   work_order           1         1 = pixel-major work items, 0 = sample-major
//   defer_slot           1         path slots are written where a path first hits
//   bounds_retire        1         camera rays that cannot reach the scene's bounds retire in k_raygen
//   fused                1         LDS-resident scenes run the fused persistent kernels
//   path_bw              -1        which fused kernel: -1 = the scene option decides (k_path), 1 = k_path_bw
//   pool_slots           0         pin the path pool (slots); 0 = the memory plan decides
//   sample_buffer_mb     0         pin the per-sample buffer (MiB); 0 = the memory plan decides
//   assume_free_mb       0         tests: plan as if this many MiB were free on the device
//   incremental          1         transform-only edits update the tree in place
//   bvh_collapse         1         1 = cost-optimal collapse to 8-wide nodes, 0 = greedy
//   shadow_order         -1        visiting order of shadow walks: -1 = measured per scene (gi_render.cpp shadowOrder), 0 = near-to-far, 1 = slot order
//   peer_copies          1         multi-device gather: 0 = stage every device's row share through pinned host memory even where peer access exists
//   shade_variants       1         OpenPBR materials without optional lobes are binned and shaded by the BASE variant of k_shade (gi_shading.h);
//                                  0 = the full kernel for all
//   merge_shade_variants -1        -1 = thin batches (work fits the pool, <= 8 Mi items) bin BASE hits with the full OpenPBR class (one launch fewer
//                                  per iteration); 0 / 1 = never / always
//   two_stream           1         batches whose work fits the pool run their shadow launches on a second stream beside the next closest-hit launch
//                                  (gi_render.cpp "two streams")
//   two_stream_delay     0         tests: 1 / 2 = hold the main / the second stream back 0.3 ms per iteration so that the other one runs ahead
//   phase_stats          0         counting builds: print k_path's phase split / k_trace_dyn's lane accounting
#pragma once

#include <cstdlib>
#include <cstring>

namespace gi {

// value of `key` in $GATLING_OPTIONS, or `def`.  Read at every call (tests change the variable between renders of one process).
inline long optionValue(const char* key, long def)
{
  const char* s = getenv("GATLING_OPTIONS");
  if (!s) return def;
  const size_t n = strlen(key);
  while (*s) {
    while (*s == ',' || *s == ' ') s++;
    if (!strncmp(s, key, n) && s[n] == '=') return strtol(s + n + 1, nullptr, 10);
    while (*s && *s != ',') s++;
  }
  return def;
}
inline bool optionSet(const char* key) { return optionValue(key, -0x7fffffffL) != -0x7fffffffL; }

} // namespace gi
