// Wavefront path tracer kernels for gfx950 (see rrt_wavefront.hpp for the queue layout).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rrt_bvh.hpp"
#include "rrt_shade.hpp"
#include "rrt_wavefront.hpp"

namespace rrt {

namespace {

RRT_DEV uint32_t lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Per-block event counts: waves add into LDS, then one thread per counter adds the block's total to
// the block's own row in global memory (rows are summed on the host after the render), so the
// per-trip kernels issue no global atomics for statistics.
struct BlockCounters {
    uint32_t v[COUNTER_COUNT];
};

RRT_DEV void counters_flush(const BlockCounters& lds, unsigned long long* rows) {
    if (threadIdx.x < COUNTER_COUNT && lds.v[threadIdx.x])
        rows[(size_t)blockIdx.x * COUNTER_COUNT + threadIdx.x] += lds.v[threadIdx.x];
}

}  // namespace

// ------------------------------------------------------------------ set-up
// Marks every slot of queue 0 empty (the first wf_shade then fills it with camera samples),
// positions the per-wave sample cursors and clears the control block.
template <typename R>
__global__ void __launch_bounds__(256) wf_init(RenderParams rp, WfParams<R> wp) {
    typedef WfElem<R> E;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < wp.capacity) wp.q[0].hit[i] = E::v2(R(0), E::of_int(WF_EMPTY));
    const uint32_t n_waves = wp.capacity / 64u;
    if (i < n_waves) {
        const unsigned long long first = (unsigned long long)i * 64ull;
        const unsigned long long n_pixels = (unsigned long long)rp.width * rp.height;
        WfWaveCursor c;
        c.s = (uint32_t)(first / n_pixels);
        c.p = (uint32_t)(first % n_pixels);
        c.off = 0;
        c.pad = 0;
        wp.cursors[i] = c;
    }
    if (i == 0) {
        wp.ctl->count[0] = wp.capacity;
        wp.ctl->count[1] = wp.ctl->count[2] = wp.ctl->count[3] = 0;
        wp.ctl->dyn[0] = wp.ctl->dyn[1] = 0;
        wp.ctl->scount[0] = wp.ctl->scount[1] = wp.ctl->sdyn[0] = wp.ctl->sdyn[1] = 0;
        wp.ctl->rays = 0;
    }
}

// ------------------------------------------------------------------ closest hit
// One ray per lane.  The linear scan reads primitive records wave-uniformly, so all 64 lanes of a
// wave that holds at least one ray run it; lanes past the end of the queue repeat the last ray.
// shadow: the queue of shadow rays (WfParams::sq) instead of the ray queue
template <typename R, bool BVH>
__global__ void __launch_bounds__(WF_HIT_BLOCK) wf_closest_hit(SceneView<R> sv, WfParams<R> wp, int parity, int shadow) {
    typedef WfElem<R> E;
    __shared__ BlockCounters s_ctr;
    const uint32_t n_in = shadow ? wp.ctl->scount[parity] : wp.ctl->count[parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (!shadow) {
            wp.ctl->count[parity ^ 1] = 0;   // the output counter of the wf_shade that follows
            wp.ctl->scount[parity ^ 1] = 0;  // ... and its shadow rays
        }
        wp.ctl->rays += n_in;
    }
    if (blockIdx.x * (uint32_t)WF_HIT_BLOCK >= n_in) return;
    if (BVH) {
        if (threadIdx.x < COUNTER_COUNT) s_ctr.v[threadIdx.x] = 0;
        tree_top_load(sv);
    }
    const WfQueue<R>& q = shadow ? wp.sq : wp.q[parity];
    const uint32_t idx = blockIdx.x * (uint32_t)WF_HIT_BLOCK + threadIdx.x;
    const uint32_t j = idx < n_in ? idx : n_in - 1u;
    const typename E::V4 a = q.o_dx[j];
    const typename E::V2 b = q.dyz[j];
    HitCounters hc{0, 0};
    R t;
    const int g = closest_hit<R, BVH>(sv, mk<R>(a.x, a.y, a.z), mk<R>(a.w, b.x, b.y), t, hc);
    if (idx < n_in) q.hit[idx] = E::v2(t, E::of_int(g >= 0 ? g : WF_MISS));
    if (BVH) {
        if (idx >= n_in) hc.prim_tests = hc.node_tests = 0;
        for (int off = 32; off > 0; off >>= 1) {
            hc.prim_tests += __shfl_xor(hc.prim_tests, off);
            hc.node_tests += __shfl_xor(hc.node_tests, off);
        }
        if ((threadIdx.x & 63u) == 0) {
            atomicAdd(&s_ctr.v[COUNTER_PRIM_TESTS], hc.prim_tests);
            atomicAdd(&s_ctr.v[COUNTER_NODE_TESTS], hc.node_tests);
        }
        __syncthreads();
        counters_flush(s_ctr, wp.block_counters);
    }
}

// ------------------------------------------------------------------ closest hit through the tree
// Tree traversal lengths differ by an order of magnitude between the rays of one wave (a ray that
// only sees the room's walls is done after a few nodes, one that grazes a mesh visits hundreds), so
// one-ray-per-lane leaves most lanes idle most of the time.  Here a wave is persistent: it walks its
// blocks of 64 rays and hands the next rays to lanes as they finish, whenever at least
// WF_REFILL_MIN lanes are idle.
constexpr int WF_REFILL_MIN = 20;  // (64 Mi-path queues, million-triangle scene; radix tree: 8 -> 5.80, 12 -> 5.95, 16 -> 6.00, 20 -> 5.99 Grays/s; surface-area tree: 16 -> 6.36, 20 -> 6.40, 24 -> 6.30)

constexpr int WF_STACK_LEVELS = BVH_LDS_LEVELS;
#ifndef RRT_EXP_TREE_WAVES
#define RRT_EXP_TREE_WAVES 8  // experiment: waves per SIMD the tree kernel is compiled for
#endif
__global__ void __launch_bounds__(BVH_BLOCK) __attribute__((amdgpu_waves_per_eu(RRT_EXP_TREE_WAVES)))
wf_closest_hit_tree(SceneView<float> sv, WfParams<float> wp, int parity, int refill_min, int shadow) {
    typedef float R;
    __shared__ int s_stack[WF_STACK_LEVELS][BVH_BLOCK];
    __shared__ BlockCounters s_ctr;
    tree_top_load(sv);
    const int top_n = tree_top_count(sv);
    const uint32_t n_in = shadow ? wp.ctl->scount[parity] : wp.ctl->count[parity];
    uint32_t* const dyn = shadow ? wp.ctl->sdyn : wp.ctl->dyn;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (!shadow) {
            wp.ctl->count[parity ^ 1] = 0;
            wp.ctl->scount[parity ^ 1] = 0;
        }
        wp.ctl->rays += n_in;
    }
    // Which blocks of 64 rays this wave walks.  Nearly all of the queue is dealt out statically (wave w takes blocks
    // w, w + n_waves, ...: no shared counter); the last 1/64 is handed out block by block through an atomic counter, so
    // that waves whose rays happened to be cheap help the others finish.  Every dynamic block costs its wave the round
    // trip of the atomic (a tenth of the time it spends on the block's rays): a quarter of the queue handed out this
    // way, which paid while the kernel was slower, cost 4 % at the end (1/8: 2 % less, 1/64 = none at all).
    const uint32_t n_ray_blocks = (n_in + 63u) / 64u;
    if (blockIdx.x * (uint32_t)(BVH_BLOCK / 64) >= n_ray_blocks) return;  // more waves than blocks of rays
    const uint32_t stride = gridDim.x * (BVH_BLOCK / 64);
#ifndef RRT_EXP_DYNAMIC_SHARE
#define RRT_EXP_DYNAMIC_SHARE 64u  // 1 / this share of a trip's blocks is handed out dynamically
#endif
    const uint32_t static_end = (n_ray_blocks - n_ray_blocks / RRT_EXP_DYNAMIC_SHARE) / stride * stride;
    if (blockIdx.x == 0 && threadIdx.x == 0) dyn[parity ^ 1] = 0;  // the next trip's counter
    uint32_t blk = __builtin_amdgcn_readfirstlane(blockIdx.x * (BVH_BLOCK / 64) + (threadIdx.x >> 6));
    bool dynamic = false;
    auto next_block = [&]() {  // wave-uniform
        if (!dynamic) {
            blk += stride;
            if (blk < static_end) return;
            dynamic = true;
        }
        uint32_t got = 0;
        if ((threadIdx.x & 63u) == 0) got = atomicAdd(&dyn[parity], 1u);
        blk = static_end + __builtin_amdgcn_readfirstlane(got);
    };
    if (blk >= static_end) {
        dynamic = true;
        uint32_t got = 0;
        if ((threadIdx.x & 63u) == 0) got = atomicAdd(&dyn[parity], 1u);
        blk = static_end + __builtin_amdgcn_readfirstlane(got);
    }
    if (threadIdx.x < COUNTER_COUNT) s_ctr.v[threadIdx.x] = 0;
    __syncthreads();
    uint32_t off = 0;
    const WfQueue<float>& q = shadow ? wp.sq : wp.q[parity];

    int spill[BVH_STACK_ENTRIES - WF_STACK_LEVELS];  // (see TreeStackSpill)
    TreeWalk<R> w;
    w.o = mk<R>(0, 0, 0);
    w.d = w.inv_d = mk<R>(0, 0, 1);
    hit_init(w.acc);
    w.node = BVH_DONE;
    w.sp = 0;
    HitCounters hc{0, 0};
    uint32_t mine = 0;
    bool have = false;
    auto push = [&](int node) {
        if (w.sp < WF_STACK_LEVELS) {
            s_stack[w.sp][threadIdx.x] = node;
        } else if (w.sp < BVH_STACK_ENTRIES) {
            spill[w.sp - WF_STACK_LEVELS] = node;
        } else {  // cannot happen with a tree the host admitted (rrt_types.hpp): counted, never written out of bounds
            atomicAdd(sv.stack_overflow, 1ull);
            return;
        }
        w.sp += 1;
    };
    auto pop = [&]() {
        if (w.sp == 0) {
            w.node = BVH_DONE;
            return;
        }
        w.sp -= 1;
        w.node = w.sp < WF_STACK_LEVELS ? s_stack[w.sp][threadIdx.x] : spill[w.sp - WF_STACK_LEVELS];
    };
#ifdef RRT_EXP_TREE_STATS  // experiment: one wave in 509 blocks prints how its lanes were occupied (lanes with work, lanes at a node, refills)
    uint32_t st_trips = 0, st_work = 0, st_node = 0, st_refill = 0, st_refill_idle = 0, st_mixed = 0;
#endif

    for (;;) {
        const bool idle = w.node == BVH_DONE;
        const unsigned long long idle_mask = __ballot(idle);
        const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        if (n_idle >= (uint32_t)refill_min || idle_mask == __ballot(true)) {
            if (idle && have) {
                const int g = hit_winner(w.acc);
                q.hit[mine] = make_float2(w.acc.t, __int_as_float(g >= 0 ? g : WF_MISS));
                have = false;
            }
            if (blk < n_ray_blocks) {
                // idle lanes take the rest of the current block in lane order; what does not fit waits
                // for the next refill, which starts on the wave's next block
                const uint32_t k = off + lane_rank(idle_mask);
                const unsigned long long idx64 = (unsigned long long)blk * 64ull + k;
                if (idle && k < 64u && idx64 < (unsigned long long)n_in) {
                    mine = (uint32_t)idx64;
                    const float4 a = q.o_dx[mine];
                    const float2 b = q.dyz[mine];
                    hit_init(w.acc);
                    walk_start(w, mk<R>(a.x, a.y, a.z), mk<R>(a.w, b.x, b.y));
                    scan_beside_tree(sv, w.o, w.d, w.acc, hc);
                    have = true;
                }
                off += n_idle;
                if (off >= 64u) {
                    off = 0;
                    next_block();
                }
            }
            if (__ballot(w.node != BVH_DONE) == 0ull) break;
#ifdef RRT_EXP_TREE_STATS
            st_refill += 1;
            st_refill_idle += n_idle;
#endif
        }
#ifdef RRT_EXP_TREE_STATS
        {
            const uint32_t nw = (uint32_t)__popcll(__ballot(w.node != BVH_DONE)), nn = (uint32_t)__popcll(__ballot(w.node >= 0));
            st_trips += 1;
            st_work += nw;
            st_node += nn;
            st_mixed += (nn > 0 && nn < nw) ? 1u : 0u;
        }
#endif
        // One trip: the lanes at a node do a node step; then the lanes at a leaf -- most of them because that node step
        // just took them there -- test one primitive.  A lane so spends no trip of its own on most leaves (14.9 -> ~12
        // trips per ray on the million-triangle scene, +2 %; with one step of either kind per lane and trip, sharing
        // the four 16-byte loads between the two kinds of record, a sixth of the lanes stood at a leaf in every trip).
        if (__ballot(w.node >= 0) != 0ull) {
            if (w.node >= 0) {
                uint4 q0, q1, q2, q3;
                tree_node_words(sv, w.node, top_n, q0, q1, q2, q3);
#ifdef RRT_EXP_NODE128
                {  // sensitivity experiment: a node step fetches 128 bytes (the record and its neighbour in the same 128-byte line pair)
                    const uint4* nb = reinterpret_cast<const uint4*>(sv.nodes + (w.node ^ 1));
                    const uint4 r0 = nb[0], r1 = nb[1], r2 = nb[2], r3 = nb[3];
                    if ((r0.x ^ r1.y ^ r2.z ^ r3.w) == 0x12345678u) hc.node_tests += 1;
                }
#endif
#ifdef RRT_EXP_NODE80
                {  // sensitivity experiment: one more 16-byte load per node step (from the neighbouring record)
                    const uint4 r0 = reinterpret_cast<const uint4*>(sv.nodes + (w.node ^ 1))[0];
                    if ((r0.x ^ r0.y) == 0x12345678u) hc.node_tests += 1;
                }
#endif
#ifdef RRT_EXP_EXTRA_VALU
                {  // sensitivity experiment: extra dependent FMAs per node step
                    float x = __uint_as_float(q0.x);
#pragma unroll
                    for (int k = 0; k < RRT_EXP_EXTRA_VALU; ++k) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
                    if (x == 12345.678f) hc.node_tests += 1;
                }
#endif
                R dist[4];
                int child[4];
                node_children(w, q0, q1, q2, q3, dist, child, w.acc.t);
                // (a node pushes at most three entries: with room for three in LDS no push needs the spill test)
                if (w.sp <= WF_STACK_LEVELS - 3)
                    node_descend(w, dist, child, [&](int c) { s_stack[w.sp][threadIdx.x] = c; w.sp += 1; }, pop);
                else
                    node_descend(w, dist, child, push, pop);
                hc.node_tests += 1;
            }
        }
        if (__ballot(w.node < 0 && w.node != BVH_DONE) != 0ull) {
            if (w.node < 0 && w.node != BVH_DONE) {
                const int v = ~w.node;
                const int slot = v & BVH_SLOT_MASK;
                const uint4* rec = reinterpret_cast<const uint4*>(sv.leaves + slot);
                const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
#ifdef RRT_EXP_LEAF80
                {  // sensitivity experiment: one more 16-byte load per leaf step
                    const uint4 r0 = reinterpret_cast<const uint4*>(sv.leaves + (slot ^ 1))[0];
                    if ((r0.x ^ r0.y) == 0x12345678u) hc.node_tests += 1;
                }
#endif
                leaf_test_words<R>(sv, q0, q1, q2, q3, w.o, w.d, w.acc);
                hc.prim_tests += 1;
                const int left = v >> BVH_SLOT_BITS;
                if (left > 0) w.node = ~((slot + 1) | ((left - 1) << BVH_SLOT_BITS));
                else pop();
            }
        }
    }
#ifdef RRT_EXP_TREE_STATS
    if (blockIdx.x % 509u == 3u && threadIdx.x == 64 && n_in > 40000000u)
        printf("TREESTAT trips %u work %u node %u refills %u refill_idle %u mixed %u\n", st_trips, st_work, st_node, st_refill, st_refill_idle, st_mixed);
#endif
    for (int sh = 32; sh > 0; sh >>= 1) {
        hc.prim_tests += __shfl_xor(hc.prim_tests, sh);
        hc.node_tests += __shfl_xor(hc.node_tests, sh);
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&s_ctr.v[COUNTER_PRIM_TESTS], hc.prim_tests);
        atomicAdd(&s_ctr.v[COUNTER_NODE_TESTS], hc.node_tests);
    }
    __syncthreads();
    counters_flush(s_ctr, wp.block_counters);
}

// ------------------------------------------------------------------ shade + regenerate + compact
// Lane i owns slot i of the input queue.  After the hit is processed (emission, Russian roulette)
// the lane either scatters its path or, if it holds none any more, starts the next camera sample of
// its wave's cursor.  Every lane that ends the trip with a ray writes it to the output queue at
//   block base (one atomicAdd per block) + rays of the earlier waves of the block + rank in the wave.
//
// FUSE_HIT (scenes whose closest hit is the wave-uniform linear scan): the scan runs here, on the ray
// just read, instead of in a kernel of its own -- no hit queue, the ray is read once, and the scan's
// arithmetic overlaps the queue traffic.  The live counts then rotate through three words: this
// trip reads count[c_in], adds to count[c_out] and clears the word the NEXT trip will add to (no
// block of this launch touches that one).
// NEE (RRT_RENDER_NEE, a separate instantiation): a lane at a Lambertian or Phong vertex samples a light and sends a
// shadow ray -- traced on the spot by the wave-uniform scan (FUSE_HIT), or written to the shadow queue for the tree
// kernel's next launch; wf_shadow_resolve then adds what the unblocked ones carry.  Whether the previous vertex sampled
// the lights (an emitter found now then adds nothing) travels in bit 19 of the packed sampler word.
template <typename R, bool FUSE_HIT, int BLOCK, bool NEE>
__global__ void __launch_bounds__(BLOCK)
wf_shade(SceneView<R> sv, RenderParams rp, WfParams<R> wp, LightView<R> lights, int parity, int c_in, int c_out, int c_clear, int first_trip) {
    typedef WfElem<R> E;
    typedef typename E::V4 Q4;
    typedef typename E::V2 Q2;
    constexpr int WAVES = BLOCK / 64;
    __shared__ BlockCounters s_ctr;
    __shared__ uint32_t s_wave_rays[WAVES];
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_wave_shadow[NEE && !FUSE_HIT ? WAVES : 1];
    __shared__ uint32_t s_shadow_base;

    const uint32_t idx = blockIdx.x * (uint32_t)BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t gwave = __builtin_amdgcn_readfirstlane(idx >> 6);
    const WfQueue<R>& qi = wp.q[parity];
    const WfQueue<R>& qo = wp.q[parity ^ 1];
    const uint32_t n_in = wp.ctl->count[c_in];
    const uint32_t n_pixels = rp.width * rp.height;
    if (FUSE_HIT && blockIdx.x == 0 && threadIdx.x == 0) {
        wp.ctl->count[c_clear] = 0;
        if (!first_trip) wp.ctl->rays += n_in;
    }

    if (threadIdx.x < COUNTER_COUNT) s_ctr.v[threadIdx.x] = 0;
    __syncthreads();

    Path<R> path;
    path.o = path.d = path.beta = mk<R>(0, 0, 0);
    path.depth = 0;
    path.smp = Sampler{0, 0, 0, 0, 4};
    uint32_t pixel = 0;
    int mode = MODE_NEW, counted = -1;
    bool bad = false;
    V3<R> n = mk<R>(0, 0, 0);
    MatRec<R> m;
    m.r = m.g = m.b = m.param = R(0);
    m.type = 0;
    bool lights_sampled = false;
    // the shadow ray of this lane's vertex (NEE)
    bool shadow = false;
    V3<R> s_wi = mk<R>(0, 0, 1), s_c = mk<R>(0, 0, 0);
    R s_dist = R(0);
    int s_prim = -2;

    int g = WF_EMPTY;
    R t_hit = R(0);
    if (FUSE_HIT) {
        // every lane of a wave that holds at least one ray runs the scan (wave-uniform loads)
        const bool have = idx < n_in && !first_trip;
        if (__ballot(have)) {
            const uint32_t j = have ? idx : 0u;
            const Q4 a = qi.o_dx[j];
            const Q2 b = qi.dyz[j];
            path.o = mk<R>(a.x, a.y, a.z);
            path.d = mk<R>(a.w, b.x, b.y);
            HitCounters hc{0, 0};
            const int found = closest_hit<R, false>(sv, path.o, path.d, t_hit, hc);
            g = have ? (found >= 0 ? found : WF_MISS) : WF_EMPTY;
        }
    } else if (idx < n_in) {
        const Q2 h = qi.hit[idx];
        t_hit = h.x;
        g = E::to_int(h.y);
    }
    {
        if (g >= 0) {
            if (!FUSE_HIT) {
                const Q4 a = qi.o_dx[idx];
                const Q2 b = qi.dyz[idx];
                path.o = mk<R>(a.x, a.y, a.z);
                path.d = mk<R>(a.w, b.x, b.y);
            }
            const Q4 c = qi.beta_pix[idx];
            const uint2 s = qi.smp[idx];
            path.beta = mk<R>(c.x, c.y, c.z);
            pixel = E::word_lo(c.w);
            path.smp.pixkey = pixel_key(rp.seed_lo, rp.seed_hi, pixel);
            path.smp.sample = s.x;
            path.smp.key = sample_key(path.smp.pixkey, s.x);
            path.smp.ctr = s.y & 0x7ffffu;
            lights_sampled = NEE && ((s.y >> 19) & 1u) != 0u;
            path.smp.dim = (s.y >> 20) & 7u;
            path.depth = s.y >> 23;
            V3<R> acc = mk<R>(0, 0, 0);
            const ShadeRec<R> sh = sv.shade[g];
            m = sv.mats[sh.object_id];
            mode = process_hit<R, NEE>(rp, path, sh, m, t_hit, acc, n, bad, counted, !(NEE && lights_sampled));
            // only an emitter adds radiance, and it ends the path: at most one add per sample
            if (acc.x != R(0)) atomicAdd(&rp.accum[3 * (size_t)pixel + 0], (float)acc.x);
            if (acc.y != R(0)) atomicAdd(&rp.accum[3 * (size_t)pixel + 1], (float)acc.y);
            if (acc.z != R(0)) atomicAdd(&rp.accum[3 * (size_t)pixel + 2], (float)acc.z);
            if (NEE) {
                if (mode == MAT_LAMBERTIAN || mode == MAT_PHONG) {
                    // (process_hit has moved path.o to the vertex; path.d is still the incoming direction)
                    const R pick = get_1d<R>(path.smp, rp, true);
                    R lu, lv;
                    get_draw<R>(path.smp, rp, true, lu, lv);
                    V3<R> le_w;
                    if (sample_light<R>(lights, path.o, pick, lu, lv, s_wi, s_dist, le_w, s_prim)) {
                        const V3<R> f = lobe_bxdf_cos<R>(mode, path.d, n, m, s_wi);
                        if (f.x != R(0) || f.y != R(0) || f.z != R(0)) {
                            shadow = true;
                            s_c = mk<R>(path.beta.x * f.x * le_w.x, path.beta.y * f.y * le_w.y, path.beta.z * f.z * le_w.z);
                        }
                    }
                    lights_sampled = true;
                } else if (mode != MODE_NEW) {
                    lights_sampled = false;  // mirror, glass: the next vertex is found by the scattered ray alone
                }
            }
        }
    }
    if (NEE && FUSE_HIT) {
        // the shadow rays of this wave, by the same wave-uniform scan
        const unsigned long long sm = __ballot(shadow);
        if (sm) {
            HitCounters hc{0, 0};
            R ts;
            const int gs = closest_hit<R, false>(sv, path.o, s_wi, ts, hc);
            if (shadow && (gs < 0 || gs == s_prim || ts >= s_dist * (R(1) - nee_shadow_slack<R>()))) {
                bad = bad || !(isfinite(s_c.x) && isfinite(s_c.y) && isfinite(s_c.z));
                atomicAdd(&rp.accum[3 * (size_t)pixel + 0], (float)s_c.x);
                atomicAdd(&rp.accum[3 * (size_t)pixel + 1], (float)s_c.y);
                atomicAdd(&rp.accum[3 * (size_t)pixel + 2], (float)s_c.z);
            }
            if (lane == 0) atomicAdd(&s_ctr.v[COUNTER_RAYS], (uint32_t)__popcll(sm));
        }
    }

    // -- the next camera samples of this wave go to the lanes without a path
    const bool vacant = mode == MODE_NEW;
    const unsigned long long vacant_mask = __ballot(vacant);
    bool start = false;
    if (vacant_mask) {
        WfWaveCursor cur = wp.cursors[gwave];
        cur.s = __builtin_amdgcn_readfirstlane(cur.s);
        cur.p = __builtin_amdgcn_readfirstlane(cur.p);
        cur.off = __builtin_amdgcn_readfirstlane(cur.off);
        // first sample of the wave's next block of 64
        uint32_t s1 = cur.s + wp.stride_s, p1 = cur.p + wp.stride_p;
        if (p1 >= n_pixels) {
            p1 -= n_pixels;
            s1 += 1;
        }
        const uint32_t o = cur.off + lane_rank(vacant_mask);
        uint32_t s = o < 64u ? cur.s : s1;
        uint32_t p = o < 64u ? cur.p + o : p1 + (o - 64u);
        while (p >= n_pixels) {  // more than once only for images smaller than a wave
            p -= n_pixels;
            s += 1;
        }
        start = vacant && s < rp.n_samples;
        // sample numbers grow with the lane rank, so the started lanes are a prefix of the vacant ones
        const uint32_t taken = (uint32_t)__popcll(__ballot(start));
        if (start) {
            pixel = p;
            sampler_start(path.smp, rp, pixel_key(rp.seed_lo, rp.seed_hi, pixel), rp.sample_begin + s);
            lights_sampled = false;
        }
        if (taken && lane == 0) {
            uint32_t off = cur.off + taken;
            if (off >= 64u) {
                cur.s = s1;
                cur.p = p1;
                off -= 64u;
            }
            cur.off = off;
            wp.cursors[gwave] = cur;
        }
    }

    // -- one draw for every lane that needs one, then the new ray
    // (the vertex a shadow ray starts from is path.o; the scatter below replaces only the direction)
    const V3<R> vertex = path.o;
    R u2 = R(0), v2 = R(0);
    if (start || mode_draws(mode)) get_draw<R>(path.smp, rp, start || mode_draws_2d(mode), u2, v2);
    bool have_ray = false;
    if (start) {
        const uint32_t py = pixel / rp.width;  // (a multiply-shift with two more scalar registers costs this kernel 3 %)
        const R px_c = (R)(pixel - py * rp.width) + R(0.5);
        const R py_c = (R)py + R(0.5);
        // src/renderer.rs:120-121; f32 multiplies by a reciprocal prepared on the host, f64 divides like the reference
        R u, v;
        if constexpr (sizeof(R) == 4) {
            u = (px_c + u2) * rp.inv_width;
            v = R(1) - (py_c + v2) * rp.inv_height;
        } else {
            u = (px_c + u2) / (R)rp.width;
            v = R(1) - (py_c + v2) / (R)rp.height;
        }
        path.o = mk<R>(sv.cam_o[0], sv.cam_o[1], sv.cam_o[2]);
        path.d = camera_dir<R>(sv, u, v);
        path.beta = mk<R>(1, 1, 1);
        path.depth = 0;
        have_ray = true;
    } else if (mode != MODE_NEW) {
        have_ray = scatter<R>(path, mode, n, m, u2, v2);
    }

    // -- compaction
    const unsigned long long ray_mask = __ballot(have_ray);
    const unsigned long long shadow_mask = (NEE && !FUSE_HIT) ? __ballot(shadow) : 0ull;
    const uint32_t c_start = (uint32_t)__popcll(__ballot(start));
    const uint32_t c_bad = (uint32_t)__popcll(__ballot(bad));
    if (lane == 0) {
        s_wave_rays[wave] = (uint32_t)__popcll(ray_mask);
        if (NEE && !FUSE_HIT) s_wave_shadow[wave] = (uint32_t)__popcll(shadow_mask);
        if (c_start) atomicAdd(&s_ctr.v[COUNTER_SAMPLES], c_start);
        if (c_bad) atomicAdd(&s_ctr.v[COUNTER_NONFINITE], c_bad);
    }
#pragma unroll
    for (int k = 1; k < 5; ++k) {
        const uint32_t c = (uint32_t)__popcll(__ballot(counted == k));
        if (lane == 0 && c) atomicAdd(&s_ctr.v[COUNTER_BOUNCE0 + k], c);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < WAVES; ++w) total += s_wave_rays[w];
        s_base = total ? atomicAdd(&wp.ctl->count[c_out], total) : 0u;
        if (NEE && !FUSE_HIT) {
            uint32_t st = 0;
            for (int w = 0; w < WAVES; ++w) st += s_wave_shadow[w];
            s_shadow_base = st ? atomicAdd(&wp.ctl->scount[c_out], st) : 0u;
        }
    }
    __syncthreads();
    if (have_ray) {
        uint32_t slot = s_base + lane_rank(ray_mask);
        for (uint32_t w = 0; w < wave; ++w) slot += s_wave_rays[w];
        qo.o_dx[slot] = E::v4(path.o.x, path.o.y, path.o.z, path.d.x);
        qo.dyz[slot] = E::v2(path.d.y, path.d.z);
        qo.beta_pix[slot] = E::v4(path.beta.x, path.beta.y, path.beta.z, E::of_words(pixel, 0u));
        qo.smp[slot] = make_uint2(path.smp.sample, path.smp.ctr | ((NEE && lights_sampled) ? (1u << 19) : 0u) | (path.smp.dim << 20) | (path.depth << 23));
    }
    if (NEE && !FUSE_HIT) {
        if (shadow) {
            uint32_t slot = s_shadow_base + lane_rank(shadow_mask);
            for (uint32_t w = 0; w < wave; ++w) slot += s_wave_shadow[w];
            wp.sq.o_dx[slot] = E::v4(vertex.x, vertex.y, vertex.z, s_wi.x);
            wp.sq.dyz[slot] = E::v2(s_wi.y, s_wi.z);
            wp.sq.beta_pix[slot] = E::v4(s_c.x, s_c.y, s_c.z, E::of_words(pixel, (uint32_t)(s_prim + 2)));
            if constexpr (sizeof(R) == 4) {
                wp.sq.smp[slot] = make_uint2((uint32_t)(s_prim + 2), __float_as_uint(s_dist));
            } else {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(s_dist);
                wp.sq.smp[slot] = make_uint2((uint32_t)bits, (uint32_t)(bits >> 32));
            }
        }
    }
    counters_flush(s_ctr, wp.block_counters);
}

// The shadow rays of one trip after their closest-hit launch: what an unblocked one carries goes to its pixel.
// Unblocked: nothing on its way, the flat light's own primitive first, or nothing nearer than the sampled point.
template <typename R>
__global__ void __launch_bounds__(256) wf_shadow_resolve(RenderParams rp, WfParams<R> wp, int parity) {
    typedef WfElem<R> E;
    __shared__ BlockCounters s_ctr;
    const uint32_t n_in = wp.ctl->scount[parity];
    if (blockIdx.x * 256u >= n_in) return;
    if (threadIdx.x < COUNTER_COUNT) s_ctr.v[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool bad = false;
    if (i < n_in) {
        const typename E::V2 h = wp.sq.hit[i];
        const uint2 s = wp.sq.smp[i];
        const typename E::V4 c = wp.sq.beta_pix[i];
        const int g = E::to_int(h.y);
        int light_prim;
        R dist;
        if constexpr (sizeof(R) == 4) {
            light_prim = (int)s.x - 2;
            dist = __uint_as_float(s.y);
        } else {
            light_prim = (int)E::word_hi(c.w) - 2;
            dist = __longlong_as_double((long long)(((unsigned long long)s.y << 32) | s.x));
        }
        if (g < 0 || g == light_prim || h.x >= dist * (R(1) - nee_shadow_slack<R>())) {
            const uint32_t pixel = E::word_lo(c.w);
            bad = !(isfinite(c.x) && isfinite(c.y) && isfinite(c.z));
            atomicAdd(&rp.accum[3 * (size_t)pixel + 0], (float)c.x);
            atomicAdd(&rp.accum[3 * (size_t)pixel + 1], (float)c.y);
            atomicAdd(&rp.accum[3 * (size_t)pixel + 2], (float)c.z);
        }
    }
    const uint32_t c_bad = (uint32_t)__popcll(__ballot(bad));
    if ((threadIdx.x & 63u) == 0 && c_bad) atomicAdd(&s_ctr.v[COUNTER_NONFINITE], c_bad);
    __syncthreads();
    counters_flush(s_ctr, wp.block_counters);
}

// ------------------------------------------------------------------ launchers
template <typename R>
hipError_t wf_launch_init(const RenderParams& rp, const WfParams<R>& wp, hipStream_t stream) {
    hipLaunchKernelGGL(wf_init<R>, dim3((wp.capacity + 255u) / 256u), dim3(256), 0, stream, rp, wp);
    return hipGetLastError();
}

namespace {
// persistent waves through the tree (f32: the product path)
void launch_tree_walk(const SceneView<float>& sv, const WfParams<float>& wp, int parity, int sh, int n_cu, uint32_t grid_x, hipStream_t stream) {
    static const bool plain = getenv("RRT_DEBUG_WF_PLAIN_WALK") != nullptr;  // experiments: one ray per lane from start to end
    if (plain) {
        hipLaunchKernelGGL((wf_closest_hit<float, true>), dim3(grid_x), dim3(WF_HIT_BLOCK), 0, stream, sv, wp, parity, sh);
        return;
    }
    // as many blocks as the device really holds at once (8 per CU with the kernel's registers and LDS; a block more
    // than fits would run after all the others, alone), never more than there are rays
    static const int per_cu = [] {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wf_closest_hit_tree, BVH_BLOCK, 0) != hipSuccess || nb < 1) nb = 8;
        return nb < 8 ? nb : 8;
    }();
    const uint32_t persistent_blocks = (uint32_t)n_cu * (uint32_t)per_cu;
    const uint32_t want = grid_x < persistent_blocks ? grid_x : persistent_blocks;
    static const int refill_min = [] { const char* e = getenv("RRT_DEBUG_REFILL_MIN"); return e ? atoi(e) : WF_REFILL_MIN; }();  // experiments
    hipLaunchKernelGGL(wf_closest_hit_tree, dim3(want), dim3(BVH_BLOCK), 0, stream, sv, wp, parity, refill_min, sh);
}
// f64 (the checking mode): one ray per lane from start to end, the walk the parity kernels use
void launch_tree_walk(const SceneView<double>& sv, const WfParams<double>& wp, int parity, int sh, int, uint32_t grid_x, hipStream_t stream) {
    hipLaunchKernelGGL((wf_closest_hit<double, true>), dim3(grid_x), dim3(WF_HIT_BLOCK), 0, stream, sv, wp, parity, sh);
}
}  // namespace

template <typename R>
hipError_t wf_launch_closest_hit(const SceneView<R>& sv, bool bvh, const WfParams<R>& wp, int parity, bool shadow,
                                 int n_cu, hipStream_t stream) {
    const dim3 grid(wp.capacity / (uint32_t)WF_HIT_BLOCK), block(WF_HIT_BLOCK);
    const int sh = shadow ? 1 : 0;
    if (bvh && sv.n_bvh_prims >= 2)
        launch_tree_walk(sv, wp, parity, sh, n_cu, grid.x, stream);
    else if (bvh)
        hipLaunchKernelGGL((wf_closest_hit<R, true>), grid, block, 0, stream, sv, wp, parity, sh);
    else
        hipLaunchKernelGGL((wf_closest_hit<R, false>), grid, block, 0, stream, sv, wp, parity, sh);
    return hipGetLastError();
}

template <typename R>
hipError_t wf_launch_shade(const SceneView<R>& sv, const LightView<R>& lights, const RenderParams& rp, const WfParams<R>& wp, int parity,
                           hipStream_t stream) {
    const dim3 grid(wp.capacity / (uint32_t)WF_SHADE_BLOCK), block(WF_SHADE_BLOCK);
    if (rp.flags & RENDER_FLAG_NEE)
        hipLaunchKernelGGL((wf_shade<R, false, WF_SHADE_BLOCK, true>), grid, block, 0, stream, sv, rp, wp, lights, parity, parity, parity ^ 1, 0, 0);
    else
        hipLaunchKernelGGL((wf_shade<R, false, WF_SHADE_BLOCK, false>), grid, block, 0, stream, sv, rp, wp, lights, parity, parity, parity ^ 1, 0, 0);
    return hipGetLastError();
}

template <typename R>
hipError_t wf_launch_shadow_resolve(const RenderParams& rp, const WfParams<R>& wp, int parity, hipStream_t stream) {
    hipLaunchKernelGGL(wf_shadow_resolve<R>, dim3((wp.capacity + 255u) / 256u), dim3(256), 0, stream, rp, wp, parity);
    return hipGetLastError();
}

template <typename R>
hipError_t wf_launch_shade_scan(const SceneView<R>& sv, const LightView<R>& lights, const RenderParams& rp, const WfParams<R>& wp,
                                unsigned long long trip, hipStream_t stream) {
    // 512 threads: with the scan's scalar registers a SIMD holds 7 waves, i.e. three blocks of 512 per CU but only one of 1024
    const dim3 grid(wp.capacity / (uint32_t)WF_SHADE_SCAN_BLOCK), block(WF_SHADE_SCAN_BLOCK);
    const int parity = (int)(trip & 1), c_in = (int)(trip % 3), c_out = (int)((trip + 1) % 3), c_clear = (int)((trip + 2) % 3), first = trip == 0 ? 1 : 0;
    if (rp.flags & RENDER_FLAG_NEE)
        hipLaunchKernelGGL((wf_shade<R, true, WF_SHADE_SCAN_BLOCK, true>), grid, block, 0, stream, sv, rp, wp, lights, parity, c_in, c_out, c_clear, first);
    else
        hipLaunchKernelGGL((wf_shade<R, true, WF_SHADE_SCAN_BLOCK, false>), grid, block, 0, stream, sv, rp, wp, lights, parity, c_in, c_out, c_clear, first);
    return hipGetLastError();
}

#define RRT_WF_INSTANTIATE(R)                                                                                                              \
    template hipError_t wf_launch_init<R>(const RenderParams&, const WfParams<R>&, hipStream_t);                                           \
    template hipError_t wf_launch_closest_hit<R>(const SceneView<R>&, bool, const WfParams<R>&, int, bool, int, hipStream_t);              \
    template hipError_t wf_launch_shade<R>(const SceneView<R>&, const LightView<R>&, const RenderParams&, const WfParams<R>&, int, hipStream_t); \
    template hipError_t wf_launch_shadow_resolve<R>(const RenderParams&, const WfParams<R>&, int, hipStream_t);                            \
    template hipError_t wf_launch_shade_scan<R>(const SceneView<R>&, const LightView<R>&, const RenderParams&, const WfParams<R>&, unsigned long long, hipStream_t);
RRT_WF_INSTANTIATE(float)
RRT_WF_INSTANTIATE(double)
#undef RRT_WF_INSTANTIATE

}  // namespace rrt
