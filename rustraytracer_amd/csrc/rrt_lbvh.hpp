// Host interface of the GPU LBVH builder (rrt_lbvh.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rrt_types.hpp"

namespace rrt {

// d_lo / d_hi: n x 3 f32 primitive bounds on the device (already conservative);
// d_wide: n - 1 wide nodes (node 0 is the root); d_prim_of_slot: n entries, sorted slot -> index into
// the bounds arrays.  Needs n >= 2.  Synchronises the stream.
// leaf_max (1..8): subtrees of at most this many primitives become one leaf.
hipError_t lbvh_build(const float* d_lo, const float* d_hi, int n, const float scene_lo[3], const float scene_hi[3],
                      int leaf_max, WideNode* d_wide, int* d_prim_of_slot, hipStream_t stream);

// d_wide (m entries) = the wide nodes of a binary tree of m nodes built on the host (rrt_sah.hpp).  use_marks: every wide
// node absorbs the descendants BvhNode::pad0 names (SAH_OPEN_*), else the largest children first.  Synchronises the stream.
hipError_t bvh_widen_host_tree(const BvhNode* h_nodes, int m, bool use_marks, WideNode* d_wide, hipStream_t stream);

// out[slot] = leaf record of primitive prim_of_slot[slot] (scan-order indices), n entries.
hipError_t lbvh_pack_leaves(const PrimRec<float>* prims, const int* prim_of_slot, int n, LeafRec<float>* out, hipStream_t s);
hipError_t lbvh_pack_leaves(const PrimRec<double>* prims, const int* prim_of_slot, int n, LeafRec<double>* out, hipStream_t s);

}  // namespace rrt
