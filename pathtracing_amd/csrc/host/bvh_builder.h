// Host-side acceleration-structure build for triangle meshes.
//
// Replaces BVH::Node::Node / surface_area_heuristic (src/bvh.cpp:91-165): the reference builds a
// pointer-linked binary tree by fully sorting each node's triangles and scanning all triangles for
// every candidate split (O(n log^2 n) with large constants).  Here: a binned surface-area-heuristic
// binary build in O(n log n), run on several threads, then collapsed to 4-wide nodes with fp32
// boxes rounded outward and flattened for the GPU.  Any valid hierarchy yields the same closest
// hit, so the tree's shape is free; only exact ties between triangles can resolve differently.
#pragma once
#include <cstdint>
#include <vector>

#include "../device/scene_device.h"

namespace ptk {

struct BuiltBvh {
    std::vector<BvhNode4> nodes;  // node 0 is the root; child indices are local to this vector
    std::vector<uint32_t> order;  // order[k]: index in the input array of the k-th triangle in leaf order
    int max_depth = 0;            // of the 4-wide tree
    int max_stack = 0;            // most entries the ordered walk's stack can hold on this tree (exact, from the tree's shape)
    int sah_depth_cap = 0;        // binary levels that were split by the surface-area heuristic (see build_bvh4)
};

// `vertices` holds n triangles as 9 doubles each.  `max_leaf` is clamped to [1, 8].
// The tree never needs more than `stack_limit` (>= 42) stack entries from the walk: a mesh whose surface-area
// tree would need more is built again with the heuristic confined to fewer top levels (ranges are halved below
// them), which bounds the depth.
BuiltBvh build_bvh4(const double* vertices, int64_t n, int max_leaf, int n_bins, int threads, int stack_limit = 64);

} // namespace ptk
