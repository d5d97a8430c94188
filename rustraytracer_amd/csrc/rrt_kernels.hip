// gfx950 kernels of the path tracer and their launch wrappers.
//
//   intersect_kernel   Scene::intersect for arbitrary rays (parity gate)
//   primary_kernel     Camera::create_ray + Scene::intersect for (u, v) pairs
//   render_fused       the pixel/sample loop of render() with the MCPT bounce
//                      loop inside: one path per lane, kept in registers for
//                      all of its bounces; a lane whose path ended starts the
//                      next camera sample at once, so every closest-hit scan
//                      runs with (almost) all 64 lanes live
//   postprocess_kernel mean -> reinhard -> gamma -> white balance -> u8
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rrt_bvh.hpp"
#include "rrt_launch.hpp"
#include "rrt_shade.hpp"

namespace rrt {

// ------------------------------------------------------------------ parity kernels
// R = double: the query in f64.  R = float: the f32 query of the product kernels, tracked, with the rays it could not
// decide within f32's error bounds repeated in f64 from the caller's f64 ray (closest_hit_exact): the primitive is the
// one the f64 query names, the distance the f32 one unless the primitive changed.
template <typename R, bool BVH, bool PRIMARY>
__global__ void __launch_bounds__(256)
intersect_kernel(SceneView<R> sv, SceneView<double> sv64, const ErrRec* err, size_t n, const double* __restrict__ a, const double* __restrict__ b,
                 int32_t* __restrict__ object_id, int32_t* __restrict__ prim_id, double* __restrict__ t_out) {
    // every lane of a wave runs the scan (wave-uniform loads, wave votes); lanes past n are masked at the store
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t j = i < n ? i : n - 1;
    V3<double> o64, d64;
    if (PRIMARY) {
        o64 = mk<double>(sv64.cam_o[0], sv64.cam_o[1], sv64.cam_o[2]);
        d64 = camera_dir<double>(sv64, a[j], b[j]);
    } else {
        o64 = mk<double>(a[3 * j], a[3 * j + 1], a[3 * j + 2]);
        d64 = mk<double>(b[3 * j], b[3 * j + 1], b[3 * j + 2]);
    }
    HitCounters hc{0, 0};
    R t;
    if (BVH) tree_top_load(sv);
    int g;
    if constexpr (sizeof(R) == 4) {
        bool redone;
        g = closest_hit_exact<BVH>(sv, sv64, err, mk<float>((float)o64.x, (float)o64.y, (float)o64.z),
                                   mk<float>((float)d64.x, (float)d64.y, (float)d64.z), o64, d64, 1.f, t, hc, redone);
    } else {
        g = closest_hit<R, BVH>(sv, o64, d64, t, hc);
    }
    if (i < n) {
        if (g >= 0) {
            object_id[i] = sv.shade[g].object_id;
            prim_id[i] = sv.shade[g].prim_id;
            t_out[i] = (double)t;
        } else {
            object_id[i] = -1;
            prim_id[i] = -1;
            t_out[i] = __builtin_inf();
        }
    }
}

// ------------------------------------------------------------------ fused renderer
// One path per lane, kept in registers for all of its bounces.  Work item =
// (pixel, chunk of consecutive sample indices); items are handed out through one
// global counter, and the lanes of a wave that need an item at the same time
// share one atomic (ballot + popcount + lane rank).
//
// One trip of the loop is, for every lane:
//   E  closest hit of the lane's current ray            (src/scene.rs:35-47)
//   F  hit processing: emission, Russian roulette        (src/tracers/utils.rs:41-64)
//   A  a lane whose sample just ended closes its item / takes a new one
//   B  ... and starts its next camera sample             (src/renderer.rs:118-119)
//   C  ONE 2D sampler draw for all lanes: the pixel jitter of a lane that starts
//      a sample, the lobe sample of a lane that scatters
//   D  the new ray: Camera::create_ray or Material::scatter
// so a lane whose path ended in F has a fresh camera ray by the end of the same
// trip and (almost) all 64 lanes are live in every scan; the expensive sampler
// code (hash, stratum permutation) runs once per trip at full occupancy instead
// of once for ray generation and once for scattering.
//
// LDS_TABLES: the shading tables (ShadeRec per primitive, MatRec per object) of a scene with at most
// FUSED_LDS_PRIMS primitives are copied into LDS once per block, so the two dependent gathers of
// every bounce (primitive -> surface, object -> material) have LDS latency.
constexpr int FUSED_LDS_PRIMS = 64;

// NEE: next-event estimation (RRT_RENDER_NEE) as a separate instantiation, so that the kernels of the reference's
// estimator carry none of it: after the hit processing a lane at a Lambertian or Phong vertex samples a light and
// traces the shadow ray (a second closest-hit query in the same trip).
#ifdef RRT_EXP_WAVES8
#define RRT_FUSED_OCCUPANCY __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define RRT_FUSED_OCCUPANCY
#endif
// EXACT (RRT_RENDER_EXACT_HITS, f32 only): every closest-hit query is tracked and repeated in f64 where f32 cannot
// decide it (closest_hit_exact, with the f32 ray widened); a separate instantiation like NEE.
template <typename R, bool BVH, bool LDS_TABLES, bool NEE = false, bool EXACT = false>
__global__ void __launch_bounds__(256) RRT_FUSED_OCCUPANCY
render_fused(SceneView<R> sv, RenderParams rp, LightView<R> lights, SceneView<double> sv64, const ErrRec* err) {
    __shared__ ShadeRec<R> s_shade[LDS_TABLES ? FUSED_LDS_PRIMS : 1];
    __shared__ MatRec<R> s_mats[LDS_TABLES ? FUSED_LDS_PRIMS : 1];
    if (LDS_TABLES) {
        for (int i = threadIdx.x; i < sv.n_prims; i += 256) s_shade[i] = sv.shade[i];
        for (int i = threadIdx.x; i < sv.n_objects; i += 256) s_mats[i] = sv.mats[i];
        __syncthreads();
    }
    if (BVH) tree_top_load(sv);
    const uint32_t lane = threadIdx.x & 63u;
    // The kernel wants more scalar registers than a wave has (scene view, render parameters, 24 for a
    // pair of primitive records, wave-uniform counters); what does not fit is parked in lanes of a
    // vector register and every use costs a v_readlane in the VALU stream.  Parameters that only the
    // item bookkeeping reads (one trip in six) are therefore re-read from the kernel argument
    // segment where they are used (scalar loads, no VALU slot) and hidden from loop-invariant
    // hoisting by an empty asm: +2 % on the Cornell box, +8 % on smallpt.
    static_assert(alignof(SceneView<R>) == 8 && alignof(RenderParams) == 8, "kernel arguments are laid out back to back at 8-byte alignment");
    const char RRT_CONSTANT* kernarg = (const char RRT_CONSTANT*)__builtin_amdgcn_kernarg_segment_ptr();
    const RenderParams RRT_CONSTANT* rpk = (const RenderParams RRT_CONSTANT*)(kernarg + ((sizeof(SceneView<R>) + 7) & ~(size_t)7));

    constexpr uint32_t POOL_ITEMS = 64;  // items per refill of the wave's pool
    uint32_t pool_next = 0, pool_end = 0;  // wave-uniform

    bool have_item = false, have_ray = false, exhausted = false;
    uint32_t pixel = 0, pixkey = 0, s_next = 0, s_end = 0;
    R px_c = 0, py_c = 0;        // pixel coordinates + 0.5 of the current item
    V3<R> acc = mk<R>(0, 0, 0);  // radiance sum of the current item
    Path<R> path;
    path.o = path.d = path.beta = mk<R>(0, 0, 0);
    path.depth = 0;
    path.smp = Sampler{0, 0, 0, 0, 4};

    // wave-uniform event counters: bumped by ballots taken outside the divergent
    // regions, so they stay in scalar registers
    uint32_t c_rays = 0, c_nonfinite = 0, c_samples = 0, c_redos = 0;
    uint32_t shadow_rays = 0;       // per lane (NEE only)
    bool lights_sampled = false;    // NEE: the path's previous vertex sampled the lights, so an emitter found now adds nothing
    HitCounters hc{0, 0};
    BounceCounters bc{{0, 0, 0, 0, 0}};

    for (;;) {
        // -- E + F
        c_rays += (uint32_t)__popcll(__ballot(have_ray));
        int mode = MODE_NEW, counted = -1;
        bool bad = false;
        V3<R> n;     // surface of this trip's hit: set by process_hit, read by scatter, both only
        MatRec<R> m;  // for lanes whose path goes on (mode != MODE_NEW)
        if (have_ray) {
            R t;
#ifdef RRT_EXP_NOSCAN
            int g = (int)(path.smp.key % (uint32_t)sv.n_prims);
            t = R(0.3);
#else
            int g;
            if constexpr (EXACT && sizeof(R) == 4) {
                bool redone;
                g = closest_hit_exact<BVH>(sv, sv64, err, path.o, path.d, mk<double>(path.o.x, path.o.y, path.o.z),
                                           mk<double>(path.d.x, path.d.y, path.d.z), 0.f, t, hc, redone);
                c_redos += redone ? 1u : 0u;
            } else {
                g = closest_hit<R, BVH>(sv, path.o, path.d, t, hc);
            }
#endif
#ifdef RRT_EXP_NOSHADE
            acc.x += t; acc.y += (R)g;
#else
            if (g >= 0) {
                const ShadeRec<R> sh = LDS_TABLES ? s_shade[g] : sv.shade[g];
                m = LDS_TABLES ? s_mats[sh.object_id] : sv.mats[sh.object_id];
                mode = process_hit<R, NEE>(rp, path, sh, m, t, acc, n, bad, counted, !(NEE && lights_sampled));
            }
#endif
            if (NEE) {
                if (mode == MAT_LAMBERTIAN || mode == MAT_PHONG) {
                    // (process_hit has moved path.o to the vertex; path.d is still the incoming direction)
                    const R pick = get_1d<R>(path.smp, rp, true);
                    R lu, lv;
                    get_draw<R>(path.smp, rp, true, lu, lv);
                    V3<R> wi, le_w;
                    R dist;
                    int light_prim;
                    if (sample_light<R>(lights, path.o, pick, lu, lv, wi, dist, le_w, light_prim)) {
                        const V3<R> f = lobe_bxdf_cos<R>(mode, path.d, n, m, wi);
                        if (f.x != R(0) || f.y != R(0) || f.z != R(0)) {
                            R ts;
                            const int gs = closest_hit<R, BVH>(sv, path.o, wi, ts, hc);
                            shadow_rays += 1;
                            if (gs < 0 || gs == light_prim || ts >= dist * (R(1) - nee_shadow_slack<R>())) {
                                const V3<R> c = mk<R>(path.beta.x * f.x * le_w.x, path.beta.y * f.y * le_w.y, path.beta.z * f.z * le_w.z);
                                bad = bad || !(isfinite(c.x) && isfinite(c.y) && isfinite(c.z));
                                acc = acc + c;
                            }
                        }
                    }
                    lights_sampled = true;
                } else if (mode != MODE_NEW) {
                    lights_sampled = false;  // mirror, glass: the next vertex is found by the scattered ray alone
                }
            }
        }
        c_nonfinite += (uint32_t)__popcll(__ballot(bad));
#pragma unroll
        for (int k = 1; k < 5; ++k) bc.bounce[k] += (uint32_t)__popcll(__ballot(counted == k));

        // -- A: the sample is over: close the item after its last sample, take the next one
        const bool fresh = mode == MODE_NEW && !exhausted;
        if (fresh && have_item && s_next == s_end) {
            const RenderParams RRT_CONSTANT* q = rpk;
            asm volatile("" : "+s"(q));
            float* accum = q->accum;
            atomicAdd(&accum[3 * (size_t)pixel + 0], (float)acc.x);
            atomicAdd(&accum[3 * (size_t)pixel + 1], (float)acc.y);
            atomicAdd(&accum[3 * (size_t)pixel + 2], (float)acc.z);
            have_item = false;
        }
        const bool need = fresh && !have_item;
        const unsigned long long need_mask = __ballot(need);
        if (need_mask) {
            const RenderParams RRT_CONSTANT* q = rpk;
            asm volatile("" : "+s"(q));
            const uint32_t p_width = q->width, p_chunk = q->chunk, p_n_samples = q->n_samples, p_sample_begin = q->sample_begin;
            const uint32_t p_seed_lo = q->seed_lo, p_seed_hi = q->seed_hi;
            const uint32_t dp_m = q->div_pixels_m, dp_s = q->div_pixels_s, dw_m = q->div_width_m, dw_s = q->div_width_s;
            const uint32_t p_big = q->n_big_chunks, p_small = q->small_chunk;
            const uint32_t n_pixels = p_width * q->height;
            const unsigned long long p_n_items = q->n_items;
            unsigned long long* p_work_counter = q->work_counter;
            // Items come from a wave-private pool [pool_next, pool_end) held in scalar registers; one
            // global atomic refills it with POOL_ITEMS consecutive items.  (One atomic per requesting
            // wave and item, on a single counter word, saturates at well under 100 M/s chip-wide and
            // was the bottleneck of the whole kernel.)
            const uint32_t cnt = (uint32_t)__popcll(need_mask);
            const uint32_t avail = pool_end - pool_next;
            uint32_t fresh_base = 0;
            if (cnt > avail) {
                const int leader = __ffsll((long long)need_mask) - 1;
                unsigned long long got = 0;
                if ((int)lane == leader) got = atomicAdd(p_work_counter, (unsigned long long)POOL_ITEMS);
                uint32_t lo = __builtin_amdgcn_readlane((uint32_t)got, leader);
                uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(got >> 32), leader);
                // the counter only ever exceeds n_items (< 2^31) by one pool per wave: no wrap
                fresh_base = hi ? 0xfffffff0u : lo;
            }
            if (need) {
                uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need_mask >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)need_mask, 0u));
                uint32_t w32 = rank < avail ? pool_next + rank : fresh_base + (rank - avail);
                if ((unsigned long long)w32 < p_n_items) {
                    uint32_t chunk = fast_div(w32, dp_m, dp_s);
                    pixel = w32 - chunk * n_pixels;
                    pixkey = pixel_key(p_seed_lo, p_seed_hi, pixel);
                    uint32_t py = fast_div(pixel, dw_m, dw_s);
                    px_c = (R)(pixel - py * p_width) + R(0.5);
                    py_c = (R)py + R(0.5);
                    const bool big = chunk < p_big;
                    uint32_t first = big ? chunk * p_chunk : p_big * p_chunk + (chunk - p_big) * p_small;
                    uint32_t last = first + (big ? p_chunk : p_small);
                    if (last > p_n_samples) last = p_n_samples;
                    s_next = p_sample_begin + first;
                    s_end = p_sample_begin + last;
                    acc = mk<R>(0, 0, 0);
                    have_item = true;
                } else {
                    exhausted = true;
                }
            }
            if (cnt > avail) {
                pool_next = fresh_base + (cnt - avail);
                pool_end = fresh_base + POOL_ITEMS;
            } else {
                pool_next += cnt;
            }
        }
        if (__ballot(!exhausted) == 0ull) break;

        // -- B: the head of the sample loop, src/renderer.rs:118-119
        const bool start = fresh && have_item;
        c_samples += (uint32_t)__popcll(__ballot(start));
        if (start) {
            sampler_start(path.smp, rp, pixkey, s_next);
            s_next += 1;
        }

        // -- C: one draw for every lane that needs one
        R u2 = R(0), v2 = R(0);
        if (start || mode_draws(mode)) get_draw<R>(path.smp, rp, start || mode_draws_2d(mode), u2, v2);

        // -- D: the next ray
        have_ray = false;
        if (start) {
            // u = (x + u_off + 0.5) / W, v = 1 - (y + v_off + 0.5) / H (src/renderer.rs:120-121);
            // the f32 kernels multiply by a reciprocal prepared on the host
            const RenderParams RRT_CONSTANT* q = rpk;
            const SceneView<R> RRT_CONSTANT* cam = (const SceneView<R> RRT_CONSTANT*)kernarg;  // the first kernel argument
            asm volatile("" : "+s"(q), "+s"(cam));
            R u, v;
            if (sizeof(R) == 4) {
                u = (px_c + u2) * (R)q->inv_width;
                v = R(1) - (py_c + v2) * (R)q->inv_height;
            } else {
                u = (px_c + u2) / (R)q->width;
                v = R(1) - (py_c + v2) / (R)q->height;
            }
            path.o = mk<R>(cam->cam_o[0], cam->cam_o[1], cam->cam_o[2]);
            // Camera::create_ray (src/camera.rs:38-45); cam_ll already has the origin subtracted
            path.d = normalize(mk<R>(cam->cam_ll[0] + u * cam->cam_h[0] + v * cam->cam_v[0],
                                     cam->cam_ll[1] + u * cam->cam_h[1] + v * cam->cam_v[1],
                                     cam->cam_ll[2] + u * cam->cam_h[2] + v * cam->cam_v[2]));
            path.beta = mk<R>(1, 1, 1);
            path.depth = 0;
            have_ray = true;
            lights_sampled = false;
        } else if (mode != MODE_NEW) {
            have_ray = scatter<R>(path, mode, n, m, u2, v2);
        }
    }
    if (NEE) {
        for (int off = 32; off > 0; off >>= 1) shadow_rays += __shfl_xor(shadow_rays, off);
        c_rays += shadow_rays;
    }
    if (EXACT) {
        for (int off = 32; off > 0; off >>= 1) c_redos += __shfl_xor(c_redos, off);
    }
    if (!BVH) {
        hc.prim_tests = c_rays * (uint32_t)sv.n_prims;  // a linear scan tests every primitive
    } else {
        // tree traversal counts per lane: fold the wave's 64 counters into lane 0
        for (int off = 32; off > 0; off >>= 1) {
            hc.prim_tests += __shfl_xor(hc.prim_tests, off);
            hc.node_tests += __shfl_xor(hc.node_tests, off);
        }
    }

    if (lane == 0) {
        const RenderParams RRT_CONSTANT* q = rpk;
        asm volatile("" : "+s"(q));
        // (a row picked by the number of the wave's last item pool: spread, and already in a scalar register)
        unsigned long long* counters = q->counters + (size_t)((pool_end >> 6) & (uint32_t)(COUNTER_ROWS - 1)) * COUNTER_COUNT;
        atomicAdd(&counters[COUNTER_RAYS], (unsigned long long)c_rays);
        atomicAdd(&counters[COUNTER_PRIM_TESTS], (unsigned long long)hc.prim_tests);
        atomicAdd(&counters[COUNTER_NODE_TESTS], (unsigned long long)hc.node_tests);
        for (int k = 0; k < 5; ++k)
            atomicAdd(&counters[COUNTER_BOUNCE0 + k], (unsigned long long)bc.bounce[k]);
        atomicAdd(&counters[COUNTER_NONFINITE], (unsigned long long)c_nonfinite);
        atomicAdd(&counters[COUNTER_SAMPLES], (unsigned long long)c_samples);
        if (EXACT) atomicAdd(&counters[COUNTER_EXACT_REDOS], (unsigned long long)c_redos);
    }
}

// ------------------------------------------------------------------ sampler probe
__global__ void __launch_bounds__(256) sampler_values_kernel(RenderParams rp, uint32_t pixel, float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rp.n_samples) return;
    Sampler s;
    sampler_start(s, rp, pixel_key(rp.seed_lo, rp.seed_hi, pixel), rp.sample_begin + i);
    float* o = out + 8 * (size_t)i;
    float unused;
    get_draw<float>(s, rp, true, o[0], o[1]);
    o[2] = get_1d<float>(s, rp, true);  // the two forms of a 1D draw: get_1d (Russian roulette) ...
    get_draw<float>(s, rp, true, o[3], o[4]);
    get_draw<float>(s, rp, false, o[5], unused);  // ... and get_draw (the dielectric's number)
    get_draw<float>(s, rp, true, o[6], o[7]);
}

// ------------------------------------------------------------------ material probe
// Material::scatter + the throughput factor for n independent (incoming direction, normal) pairs of
// one material, with the random numbers supplied by the caller: the code the renderers run per
// bounce, made observable one sample at a time.
template <typename R>
__global__ void __launch_bounds__(256)
material_probe_kernel(MatRec<R> m, size_t n, const double* __restrict__ d_in, const double* __restrict__ normal,
                      const double* __restrict__ u, double* __restrict__ d_out, double* __restrict__ weight,
                      int32_t* __restrict__ ok) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Path<R> p;
    p.o = mk<R>(0, 0, 0);
    p.d = mk<R>((R)d_in[3 * i], (R)d_in[3 * i + 1], (R)d_in[3 * i + 2]);
    p.beta = mk<R>(1, 1, 1);
    p.depth = 0;
    p.smp = Sampler{0, 0, 0, 0, 4};
    const V3<R> nn = mk<R>((R)normal[3 * i], (R)normal[3 * i + 1], (R)normal[3 * i + 2]);
    const R r1 = (R)u[3 * i];
    bool good = false;
    if (m.type != MAT_EMISSIVE)  // an emitter's scatter returns None (src/material.rs:44-46)
        good = scatter_with<R>(p, m.type, nn, m, (R)u[3 * i + 1], (R)u[3 * i + 2], [&]() { return r1; });
    ok[i] = good ? 1 : 0;
    d_out[3 * i] = (double)p.d.x;
    d_out[3 * i + 1] = (double)p.d.y;
    d_out[3 * i + 2] = (double)p.d.z;
    weight[3 * i] = (double)p.beta.x;
    weight[3 * i + 1] = (double)p.beta.y;
    weight[3 * i + 2] = (double)p.beta.z;
}

// ------------------------------------------------------------------ post-process
// src/renderer.rs:39-71,128-136 in f64 (the reference's arithmetic type); one
// thread per colour component.
__global__ void __launch_bounds__(256)
postprocess_kernel(const float* __restrict__ sum, double inv_spp, PostParams pp, size_t n_components,
                   uint8_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_components) return;
    double c = (double)sum[i] * inv_spp;
    if (pp.tone_mapping == 1) c = c / (c + 1.0);
    if (pp.gamma) c = ::pow(c, 1.0 / 2.2);
    if (pp.has_wb) c = c * pp.wb[i % 3];
    double s = c * 255.0;
    // Rust: (c * 255.0).min(255.0) as u8 -- min drops a NaN operand, the cast saturates
    double m = (s != s) ? 255.0 : (s < 255.0 ? s : 255.0);
    out[i] = m <= 0.0 ? (uint8_t)0 : (uint8_t)m;
}

// ------------------------------------------------------------------ launchers
template <typename R>
static hipError_t launch_intersect_t(const SceneView<R>& sv, const SceneView<double>& sv64, const ErrRec* err, bool bvh, bool primary, size_t n, const double* a,
                                     const double* b, int32_t* oid, int32_t* pid, double* t,
                                     hipStream_t stream) {
    if (n == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    if (bvh) {
        if (primary)
            hipLaunchKernelGGL((intersect_kernel<R, true, true>), grid, block, 0, stream, sv, sv64, err, n, a, b, oid, pid, t);
        else
            hipLaunchKernelGGL((intersect_kernel<R, true, false>), grid, block, 0, stream, sv, sv64, err, n, a, b, oid, pid, t);
    } else {
        if (primary)
            hipLaunchKernelGGL((intersect_kernel<R, false, true>), grid, block, 0, stream, sv, sv64, err, n, a, b, oid, pid, t);
        else
            hipLaunchKernelGGL((intersect_kernel<R, false, false>), grid, block, 0, stream, sv, sv64, err, n, a, b, oid, pid, t);
    }
    return hipGetLastError();
}

hipError_t launch_intersect(const SceneView<float>& sv, const SceneView<double>& sv64, const ErrRec* err, bool bvh, bool primary, size_t n, const double* a,
                            const double* b, int32_t* oid, int32_t* pid, double* t, hipStream_t s) {
    return launch_intersect_t<float>(sv, sv64, err, bvh, primary, n, a, b, oid, pid, t, s);
}
hipError_t launch_intersect(const SceneView<double>& sv, bool bvh, bool primary, size_t n, const double* a,
                            const double* b, int32_t* oid, int32_t* pid, double* t, hipStream_t s) {
    return launch_intersect_t<double>(sv, sv, nullptr, bvh, primary, n, a, b, oid, pid, t, s);
}

template <typename R>
static hipError_t launch_render_fused_t(const SceneView<R>& sv, const SceneView<double>& sv64, const ErrRec* err, const LightView<R>& lv, bool bvh, const RenderParams& rp, int n_blocks,
                                        hipStream_t stream) {
    // RRT_DEBUG_LDS_BYTES: occupancy experiments only (unused dynamic LDS caps the blocks per CU)
    static const size_t lds = [] { const char* e = getenv("RRT_DEBUG_LDS_BYTES"); return e ? (size_t)atol(e) : (size_t)0; }();
    const bool tables = sv.n_prims <= FUSED_LDS_PRIMS && sv.n_objects <= FUSED_LDS_PRIMS;
    if (rp.flags & RENDER_FLAG_NEE) {
        if (bvh)
            hipLaunchKernelGGL((render_fused<R, true, false, true>), dim3(n_blocks), dim3(256), lds, stream, sv, rp, lv, sv64, err);
        else
            hipLaunchKernelGGL((render_fused<R, false, false, true>), dim3(n_blocks), dim3(256), lds, stream, sv, rp, lv, sv64, err);
        return hipGetLastError();
    }
    if constexpr (sizeof(R) == 4) if (rp.flags & RENDER_FLAG_EXACT) {
        if (bvh)
            hipLaunchKernelGGL((render_fused<R, true, false, false, true>), dim3(n_blocks), dim3(256), lds, stream, sv, rp, lv, sv64, err);
        else if (tables)
            hipLaunchKernelGGL((render_fused<R, false, true, false, true>), dim3(n_blocks), dim3(256), lds, stream, sv, rp, lv, sv64, err);
        else
            hipLaunchKernelGGL((render_fused<R, false, false, false, true>), dim3(n_blocks), dim3(256), lds, stream, sv, rp, lv, sv64, err);
        return hipGetLastError();
    }
    if (bvh)
        hipLaunchKernelGGL((render_fused<R, true, false>), dim3(n_blocks), dim3(256), lds, stream, sv, rp, lv, sv64, err);
    else if (tables)
        hipLaunchKernelGGL((render_fused<R, false, true>), dim3(n_blocks), dim3(256), lds, stream, sv, rp, lv, sv64, err);
    else
        hipLaunchKernelGGL((render_fused<R, false, false>), dim3(n_blocks), dim3(256), lds, stream, sv, rp, lv, sv64, err);
    return hipGetLastError();
}
hipError_t launch_render_fused(const SceneView<float>& sv, const SceneView<double>& sv64, const ErrRec* err, const LightView<float>& lv, bool bvh, const RenderParams& rp, int n_blocks, hipStream_t s) {
    return launch_render_fused_t<float>(sv, sv64, err, lv, bvh, rp, n_blocks, s);
}
hipError_t launch_render_fused(const SceneView<double>& sv, const LightView<double>& lv, bool bvh, const RenderParams& rp, int n_blocks, hipStream_t s) {
    return launch_render_fused_t<double>(sv, sv, nullptr, lv, bvh, rp, n_blocks, s);
}

template <typename R>
static hipError_t launch_material_probe_t(const MatRec<R>& m, size_t n, const double* d_in, const double* normal, const double* u,
                                          double* d_out, double* weight, int32_t* ok, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(material_probe_kernel<R>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, m, n, d_in, normal, u,
                       d_out, weight, ok);
    return hipGetLastError();
}
hipError_t launch_material_probe(const MatRec<float>& m, size_t n, const double* d_in, const double* normal, const double* u,
                                 double* d_out, double* weight, int32_t* ok, hipStream_t stream) {
    return launch_material_probe_t<float>(m, n, d_in, normal, u, d_out, weight, ok, stream);
}
hipError_t launch_material_probe(const MatRec<double>& m, size_t n, const double* d_in, const double* normal, const double* u,
                                 double* d_out, double* weight, int32_t* ok, hipStream_t stream) {
    return launch_material_probe_t<double>(m, n, d_in, normal, u, d_out, weight, ok, stream);
}

hipError_t launch_sampler_values(const RenderParams& rp, uint32_t pixel, float* out, hipStream_t stream) {
    dim3 block(256), grid((rp.n_samples + 255) / 256);
    hipLaunchKernelGGL(sampler_values_kernel, grid, block, 0, stream, rp, pixel, out);
    return hipGetLastError();
}

hipError_t launch_postprocess(const float* sum, double inv_spp, const PostParams& pp, size_t n_components,
                              uint8_t* out, hipStream_t stream) {
    if (n_components == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n_components + 255) / 256));
    hipLaunchKernelGGL(postprocess_kernel, grid, block, 0, stream, sum, inv_spp, pp, n_components, out);
    return hipGetLastError();
}

}  // namespace rrt
