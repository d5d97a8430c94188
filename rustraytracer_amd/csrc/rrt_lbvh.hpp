// Host interface of the GPU LBVH builder (rrt_lbvh.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rrt_types.hpp"

namespace rrt {

// d_lo / d_hi: n x 3 f32 primitive bounds on the device (already conservative);
// d_nodes: n - 1 nodes (node 0 is the root); d_prim_of_slot: n entries, sorted slot -> index into
// the bounds arrays.  Needs n >= 2.  Synchronises the stream.
hipError_t lbvh_build(const float* d_lo, const float* d_hi, int n, const float scene_lo[3], const float scene_hi[3],
                      BvhNode* d_nodes, int* d_prim_of_slot, hipStream_t stream);

}  // namespace rrt
