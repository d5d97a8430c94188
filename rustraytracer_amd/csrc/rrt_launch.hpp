// Host-callable launch wrappers of the kernels in rrt_kernels.hip / rrt_lbvh.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "rrt_types.hpp"

namespace rrt {

struct PostParams {
    int32_t tone_mapping;  // 0 none, 1 reinhard
    int32_t gamma;
    int32_t has_wb;
    int32_t pad;
    double wb[3];
};

// a, b: PRIMARY ? (u[n], v[n]) : (origins[3n], dirs[3n]); all pointers are device memory
// (the f32 query repeats in f64, on sv64, the rays it cannot decide within f32's error bounds: closest_hit_exact)
hipError_t launch_intersect(const SceneView<float>& sv, const SceneView<double>& sv64, const ErrRec* err, bool bvh, bool primary, size_t n, const double* a,
                            const double* b, int32_t* oid, int32_t* pid, double* t, hipStream_t s);
hipError_t launch_intersect(const SceneView<double>& sv, bool bvh, bool primary, size_t n, const double* a,
                            const double* b, int32_t* oid, int32_t* pid, double* t, hipStream_t s);

hipError_t launch_render_fused(const SceneView<float>& sv, const SceneView<double>& sv64, const ErrRec* err, const LightView<float>& lv, bool bvh, const RenderParams& rp, int n_blocks, hipStream_t s);
hipError_t launch_render_fused(const SceneView<double>& sv, const LightView<double>& lv, bool bvh, const RenderParams& rp, int n_blocks, hipStream_t s);

hipError_t launch_sampler_values(const RenderParams& rp, uint32_t pixel, float* out, hipStream_t stream);

// all pointers are device memory: d_in, normal, u, d_out, weight hold 3 doubles per sample
hipError_t launch_material_probe(const MatRec<float>& m, size_t n, const double* d_in, const double* normal, const double* u,
                                 double* d_out, double* weight, int32_t* ok, hipStream_t stream);
hipError_t launch_material_probe(const MatRec<double>& m, size_t n, const double* d_in, const double* normal, const double* u,
                                 double* d_out, double* weight, int32_t* ok, hipStream_t stream);

hipError_t launch_postprocess(const float* sum, double inv_spp, const PostParams& pp, size_t n_components,
                              uint8_t* out, hipStream_t stream);

}  // namespace rrt
