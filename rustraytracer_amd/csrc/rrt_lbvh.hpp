// Host interface of the GPU tree builder (rrt_lbvh.hip, rrt_ploc.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rrt_types.hpp"

namespace rrt {

// The tree on the GPU.  d_lo / d_hi: n x 3 f32 primitive bounds on the device (already conservative);
// d_wide: n - 1 wide nodes (node 0 is the root); d_prim_of_slot: n entries, sorted slot -> index into
// the bounds arrays (the Morton order).  One primitive per leaf.  Needs n >= 2.  Synchronises the stream.
// *depth (here and below) = the levels of wide nodes of the finished tree.
hipError_t lbvh_build(const float* d_lo, const float* d_hi, int n, const float scene_lo[3], const float scene_hi[3],
                      WideNode* d_wide, int* d_prim_of_slot, hipStream_t stream, int* depth);

// The binary tree by clustering along the Morton order (rrt_ploc.hip): d_nodes = n - 1 records, node 0 the root,
// children numbered above their parents, wide-node marks in BvhNode::pad0; d_prim_of_slot = the Morton order.
// Synchronises the stream.
hipError_t ploc_build(const float* d_lo, const float* d_hi, const int* d_prim_of_slot, int n, BvhNode* d_nodes, hipStream_t stream);

// d_wide (m entries) = the wide nodes of a binary tree of m nodes built and marked on the host (rrt_sah.hpp).
// Synchronises the stream.
hipError_t bvh_widen_host_tree(const BvhNode* h_nodes, int m, WideNode* d_wide, hipStream_t stream, int* depth);

// The binary surface-area tree built on the GPU (rrt_gsah.hip): the tree SahBuilder (rrt_sah.hpp) makes from the same
// boxes -- same cut coordinates, bins, costs and tie rules -- level by level, one work-group per range.  d_size[i] = log2 of
// the half area of box i, computed by the host like SahBuilder does.  d_nodes: n - 1 records, node 0 the root, a node's
// children numbered above it (depth first), wide-node marks in BvhNode::pad0.  Needs n >= 2.  Synchronises the stream.
hipError_t gsah_build(const float* d_lo, const float* d_hi, const float* d_size, int n, BvhNode* d_nodes, int* d_prim_of_slot,
                      hipStream_t stream, int* levels);
// ... and its wide nodes, packed level by level from the root (the first wide nodes of the array are the top of the tree).
hipError_t gsah_tree(const float* d_lo, const float* d_hi, const float* d_size, int n, WideNode* d_wide, int* d_prim_of_slot,
                     hipStream_t stream, int* depth);

// out[slot] = leaf record of primitive prim_of_slot[slot] (scan-order indices), n entries.
hipError_t lbvh_pack_leaves(const PrimRec<float>* prims, const int* prim_of_slot, int n, LeafRec<float>* out, hipStream_t s);
hipError_t lbvh_pack_leaves(const PrimRec<double>* prims, const int* prim_of_slot, int n, LeafRec<double>* out, hipStream_t s);

}  // namespace rrt
