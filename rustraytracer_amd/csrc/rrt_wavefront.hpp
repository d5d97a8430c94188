// Wavefront path tracer: path state lives in HBM queues between per-bounce kernels.
//
// One trip handles every live path once:
//   wf_closest_hit   ray queue -> hit queue            (Scene::intersect, src/scene.rs:35-47)
//   wf_shade         ray + hit queue -> next ray queue (src/tracers/utils.rs:41-87, one loop body)
// (for scenes small enough for the wave-uniform linear scan the two are one kernel, wf_shade<true>)
// wf_shade also starts new camera samples in every lane that has no surviving path, so the queue
// stays full until the render runs out of samples, and writes survivors and new paths densely
// (wave64 ballot + lane rank inside a wave, one counter reservation per block).
#pragma once
#include <hip/hip_runtime.h>

#include "rrt_types.hpp"

namespace rrt {

// Element types of the queues in the arithmetic type of the render: f32 is the product path; f64 (the reference's
// type, for checking the pipeline path by path against a CPU statement of the reference) takes twice the bytes.  The fourth component of a
// 4-vector and the second of a 2-vector carry integers as bit patterns.
template <typename R> struct WfElem;
template <> struct WfElem<float> {
    typedef float4 V4;
    typedef float2 V2;
    static __host__ __device__ V4 v4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
    static __host__ __device__ V2 v2(float x, float y) { return make_float2(x, y); }
    static __device__ float of_int(int v) { return __int_as_float(v); }
    static __device__ int to_int(float v) { return __float_as_int(v); }
    // two 32-bit words in one element: f32 has room for the first only (the caller stores the second elsewhere)
    static __device__ float of_words(uint32_t lo, uint32_t) { return __uint_as_float(lo); }
    static __device__ uint32_t word_lo(float v) { return __float_as_uint(v); }
    static __device__ uint32_t word_hi(float) { return 0u; }
};
template <> struct WfElem<double> {
    typedef double4 V4;
    typedef double2 V2;
    static __host__ __device__ V4 v4(double x, double y, double z, double w) { return make_double4(x, y, z, w); }
    static __host__ __device__ V2 v2(double x, double y) { return make_double2(x, y); }
    static __device__ double of_int(int v) { return __longlong_as_double((long long)v); }
    static __device__ int to_int(double v) { return (int)__double_as_longlong(v); }
    static __device__ double of_words(uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }
    static __device__ uint32_t word_lo(double v) { return (uint32_t)(unsigned long long)__double_as_longlong(v); }
    static __device__ uint32_t word_hi(double v) { return (uint32_t)((unsigned long long)__double_as_longlong(v) >> 32); }
};

// Structure-of-arrays queue of `capacity` paths.  A path takes 56 bytes in f32 (104 in f64); the closest-hit kernel
// touches 24 (read) + 8 (write) of them.
template <typename R>
struct WfQueue {
    typename WfElem<R>::V4* o_dx;      // ray origin xyz, direction x
    typename WfElem<R>::V2* dyz;       // direction y, z
    typename WfElem<R>::V2* hit;       // t, scan-order primitive index as int bits (WF_MISS, WF_EMPTY)
    typename WfElem<R>::V4* beta_pix;  // throughput rgb, pixel index as uint bits
    uint2* smp;        // sample index inside the pixel; draws | lights sampled at the previous vertex << 19 | dimension << 20 | depth << 23
};

enum : int { WF_MISS = -1, WF_EMPTY = -2 };
constexpr uint32_t WF_MAX_DEPTH = 511;  // 9 bits of the packed sampler word (draws: 19 bits, lights sampled: 1, dimension: 3)

struct WfControl {
    uint32_t count[4];        // live paths: count[parity] (two kernels per trip) or count[trip % 3] (one kernel per trip)
    unsigned long long rays;  // closest-hit queries so far
    uint32_t dyn[2];          // tree kernel: blocks of rays handed out dynamically in the trip of this parity
    uint32_t scount[2];       // next-event estimation: shadow rays queued by the wf_shade that WROTE queue `parity`
    uint32_t sdyn[2];         // ... and the tree kernel's dynamic share of them
    uint32_t pad[2];
};

// Where a wave takes its next camera samples from.  Sample n of the render (n = s * n_pixels +
// pixel: sample-major, so neighbouring lanes get neighbouring pixels) belongs to wave
// (n / 64) % n_waves; a wave walks its own blocks of 64 in order, so no counter is shared.
struct WfWaveCursor {
    uint32_t s, p;  // sample index and pixel of the first sample of the current block
    uint32_t off;   // samples of the current block already started
    uint32_t pad;
};

template <typename R>
struct WfParams {
    WfQueue<R> q[2];
    // Next-event estimation through the tree (RRT_RENDER_NEE): the shadow rays of a trip wait here for their own
    // closest-hit launch.  o_dx / dyz: the ray; hit: what it met; beta_pix: the radiance it carries if nothing is in its
    // way, and the pixel; smp: f32 (the light's primitive + 2, the distance to the sampled point as float bits), f64 the
    // distance's 64 bits, with the light's primitive + 2 in the upper word beside the pixel.
    WfQueue<R> sq;
    WfControl* ctl;
    WfWaveCursor* cursors;              // capacity / 64
    unsigned long long* block_counters; // rows of COUNTER_COUNT, one per block of the widest launch
    uint32_t capacity;                  // paths per queue, a multiple of the shade block size
    uint32_t stride_s, stride_p;        // (64 * n_waves) / n_pixels and % n_pixels
    uint32_t pad;
};

constexpr int WF_SHADE_BLOCK = 1024;
constexpr int WF_SHADE_SCAN_BLOCK = 512;
constexpr int WF_HIT_BLOCK = 256;

// (every launcher exists for R = float and R = double)
template <typename R> hipError_t wf_launch_init(const RenderParams& rp, const WfParams<R>& wp, hipStream_t stream);
// n_cu: compute units of the device (the f32 tree kernel's grid is what they hold at once)
// shadow = false: the rays of queue `parity`; true: the shadow rays queued by the wf_shade that wrote queue `parity`
template <typename R> hipError_t wf_launch_closest_hit(const SceneView<R>& sv, bool bvh, const WfParams<R>& wp, int parity, bool shadow,
                                                       int n_cu, hipStream_t stream);
// lights.n > 0 with RENDER_FLAG_NEE in rp.flags: next-event estimation (shadow rays into wp.sq)
template <typename R> hipError_t wf_launch_shade(const SceneView<R>& sv, const LightView<R>& lights, const RenderParams& rp, const WfParams<R>& wp,
                                                 int parity, hipStream_t stream);
// adds the radiance of the unblocked shadow rays of wp.sq (after their closest-hit launch) to the image
template <typename R> hipError_t wf_launch_shadow_resolve(const RenderParams& rp, const WfParams<R>& wp, int parity, hipStream_t stream);
// One kernel per trip for linear-scan scenes: closest hit + shade (+ the shadow ray of next-event estimation, traced on
// the spot).  Trip t reads queue t & 1 and count[t % 3].
template <typename R> hipError_t wf_launch_shade_scan(const SceneView<R>& sv, const LightView<R>& lights, const RenderParams& rp,
                                                      const WfParams<R>& wp, unsigned long long trip, hipStream_t stream);

}  // namespace rrt
