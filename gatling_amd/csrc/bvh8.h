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
  uint32_t activeTris = 0;    // items the tree references: tris[0, activeTris) (order[0, activeTris) in box mode); the rest are inactive, see below
};

// Inactive items.  A triangle with a vertex that is not finite or lies beyond 1e18 in magnitude (a box with such a plane, or an inverted box) cannot be hit:
// it is left out of the tree -- the Vulkan rule the reference inherits from its driver (a primitive whose first vertex has a NaN X is inactive,
// /root/reference/src/cgpu/impl/CgpuVk.cpp:2561-2670 hands the buffers over as they are), widened to every coordinate and to magnitudes whose extents,
// surface areas or dequantised planes would overflow fp32.  Inactive items keep their scene-order id (`origId`, `order[]`): they are stored BEHIND the
// active ones, so `tris.size()` is still the input's size and ids / per-triangle side tables do not shift.

// `tris` in scene order (origId is assigned from the position).  Degenerate input (0 triangles) produces a
// single empty root so traversal code needs no special case.
void buildBvh8(const std::vector<TriRec>& tris, Bvh8& out);

// The same tree over caller-supplied, already padded boxes (6 floats per item: min xyz, max xyz): the TLAS over instance bounds and
// the per-mesh BLAS of the two-level layout.  `order[k]` is the item a leaf slot's k-th reference (Node8::triBase + offset) names.
void buildBvh8Boxes(const float* boxes, size_t count, Bvh8& out, std::vector<uint32_t>& order);

// A tree over SUBTREES that already exist (incremental scene updates: one subtree per mesh instance, rebuilt alone when its transform changes).
// `boxes`: the items' padded bounds; `itemRoots[i]`: the root node of item i with ABSOLUTE child / triangle indices.  Every item becomes an internal child
// whose node is a copy of its root, so `out.nodes` (breadth-first, root at 0) together with the items' own node ranges is one ordinary BVH8; `out.maxDepth`
// counts the levels down to and including the copied roots.
void buildTopBvh8(const float* boxes, size_t count, const Node8* itemRoots, Bvh8& out);

} // namespace gi
