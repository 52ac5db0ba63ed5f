// bvh8.h -- host-side builder of the 8-wide quantised BVH (replaces the driver's BLAS/TLAS build,
// cgpuCreateBlas/cgpuCreateTlas, /root/reference/src/cgpu/impl/CgpuVk.cpp:2561-2854).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "gi_types.h"

namespace gi {

struct Bvh8 {
  std::vector<Node8> nodes;   // breadth-first: the top of the tree comes first (LDS staging)
  std::vector<TriRec> tris;   // reordered so every node's leaf triangles are contiguous
  uint32_t maxDepth = 0;
};

// `tris` in scene order (origId is assigned from the position).  Degenerate input (0 triangles) produces a
// single empty root so traversal code needs no special case.
void buildBvh8(const std::vector<TriRec>& tris, Bvh8& out);

// The same tree over caller-supplied, already padded boxes (6 floats per item: min xyz, max xyz): the TLAS over instance bounds and
// the per-mesh BLAS of the two-level layout.  `order[k]` is the item a leaf slot's k-th reference (Node8::triBase + offset) names.
void buildBvh8Boxes(const float* boxes, size_t count, Bvh8& out, std::vector<uint32_t>& order);

} // namespace gi
