// The surface-area tree on the GPU: the binary tree of rrt_sah.hpp (binned surface-area heuristic, the same four cut
// coordinates, bins, cost and tie rules; ranges of up to 32 primitives by the exact sweep), built level by level.
//
//   * a range of more than GS_SWEEP_MAX primitives is one TASK of its level.  A range of up to GS_WAVE_MAX primitives
//     belongs to one wave: it finds the extent of the four cut coordinates, fills 4 x 16 bins in LDS (box and count,
//     LDS atomics on order-preserving integer images of the floats), lets its 64 lanes price the 64 cuts (4 coordinates
//     x 16 bin borders), takes the cheapest (the first one in coordinate-major order among equals, as a serial loop
//     with `<` would), and partitions its range stably into the other index buffer.  Children that are still tasks go
//     onto the next level's lists.
//   * a larger range is cut into CHUNKS of GS_CHUNK primitives, one work-group of 1024 threads each, so that the few
//     huge ranges at the top of the tree keep the whole chip busy: the same steps as separate launches over the
//     chunks (extent, bins -- both reduced in LDS per chunk, then merged with global atomics --, left counts, scatter)
//     with one wave per task in between (the choice of the cut; chunk offsets, node, children).
//   * node numbers need no counter and no scan: the subtree over `count` primitives has count - 1 nodes, so the task
//     with node number k and nl primitives on its left gives its left child k + 1 and its right child k + nl.
//     Children have larger numbers than their parents; the tree does not depend on scheduling.
//   * the ranges of 2..GS_SWEEP_MAX primitives are finished by one THREAD each: every cut of the order along every
//     coordinate (sorted by (coordinate, primitive), so ties do not depend on the order in memory), recursively.
//   * which descendants every 4-wide node absorbs (sah_mark_node) is decided bottom-up: inside its subtree by the
//     sweep thread, then level by level for the nodes the tasks made.
// The host launches a fixed pattern of kernels per level (grids sized by the bound 2^level and the primitive count;
// groups beyond the level's task count leave at once) and reads the task counts back only every few levels.
#include <hip/hip_runtime.h>

#include "rrt_lbvh.hpp"
#include "rrt_sah.hpp"

namespace rrt {

namespace {

constexpr int GS_AXES = 4;
constexpr int GS_BINS = 16;
constexpr int GS_SWEEP_MAX = 32;    // = SahBuilder::SWEEP_MAX
constexpr int GS_WAVE_MAX = 2048;   // larger ranges are worked on in chunks
constexpr int GS_BIG_BLOCK = 1024;
constexpr int GS_CHUNK = 4096;      // primitives per chunk of a large range (one group of GS_BIG_BLOCK threads)
constexpr int GS_BIN_WORDS = GS_AXES * GS_BINS * 7;  // per bin: box lo xyz (min), hi xyz (max) as ordered integers, count

// one chunk of a large task
struct GsChunk {
    int task;   // slot in the level's list of large tasks
    int index;  // chunk number inside the task
};
// what the launches of a level share about one large task
struct GsBigState {
    uint32_t ext[2 * GS_AXES];
    uint32_t bins[GS_BIN_WORDS];
    uint32_t part[2][6];       // degenerate ranges: the boxes of the two halves
    int best;                  // coordinate * 16 + last left bin, or -1 (degenerate: cut the order in the middle)
    int nl;
    int chunk0, n_chunks;      // this task's entries in the level's chunk table
};

struct GsTask {
    int node, first, count, pad;
};

// control block in device memory
struct GsControl {
    int n_big[2], n_wave[2];  // tasks of the two list parities
    int n_chunks[2];          // chunks of the large tasks of the two parities
    int n_sweep;              // ranges left to the sweep threads
    int n_top;                // nodes made by tasks (marked level by level afterwards)
    int max_level;
    int error;                // 1: a list overflowed (cannot happen: the ranges of a level are disjoint, so a level holds at
                              // most n / (smallest range of the class + 1) tasks of a class, and the lists are that long)
};

struct GsParams {
    const float* lo;    // 3n, primitive boxes
    const float* hi;
    const float* size;  // n: the fourth cut coordinate (log2 of the box's half area, from the host)
    int* idx[2];        // n each: primitive order, ping-pong between levels
    int* fin;           // n: the final order (ranges that left the levels, then what the sweep threads make of them)
    BvhNode* nodes;     // n - 1
    float* f;           // n - 1: sah_mark_node's expected steps
    GsTask* big[2];
    GsTask* wave[2];
    GsTask* sweep;
    GsChunk* chunks[2];
    GsBigState* big_state[2];  // cap_big each
    int* chunk_left;        // per chunk of the level: primitives going left, then (exclusive prefix) those of the task's earlier chunks
    int* top_node;
    int* top_level;
    GsControl* ctl;
    int n;
    int cap_big, cap_wave, cap_sweep, cap_top, cap_chunks;
};

// float -> unsigned with the same order, and back
__device__ __forceinline__ uint32_t f2o(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__device__ __forceinline__ float gs_coord(const GsParams& P, int p, int a) {
    return a < 3 ? 0.5f * (P.lo[3 * p + a] + P.hi[3 * p + a]) : P.size[p];
}

// dx dy + dy dz + dz dx with every product and sum rounded (what the host builder computes)
__device__ __forceinline__ float gs_half_area(const float* l, const float* h) {
#pragma clang fp contract(off)
    const float dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
    return dx * dy + dy * dz + dz * dx;
}

__device__ __forceinline__ int gs_bin(float c, float clo, float scale) {
#pragma clang fp contract(off)
    int b = (int)((c - clo) * scale);
    return b < 0 ? 0 : (b >= GS_BINS ? GS_BINS - 1 : b);
}

__global__ void gs_init_kernel(GsParams P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n) P.idx[0][i] = i;
    if (i == 0) {
        GsControl c;
        c.n_big[0] = c.n_big[1] = c.n_wave[0] = c.n_wave[1] = 0;
        c.n_sweep = c.n_top = c.max_level = c.error = 0;
        c.n_chunks[0] = c.n_chunks[1] = 0;
        const GsTask root{0, 0, P.n, 0};
        if (P.n > GS_WAVE_MAX) {
            P.big[0][0] = root;
            c.n_big[0] = 1;
            c.n_chunks[0] = (P.n + GS_CHUNK - 1) / GS_CHUNK;
            P.big_state[0][0].chunk0 = 0;
            P.big_state[0][0].n_chunks = c.n_chunks[0];
        }
        else if (P.n > GS_SWEEP_MAX) { P.wave[0][0] = root; c.n_wave[0] = 1; }
        else { P.sweep[0] = root; c.n_sweep = 1; }
        *P.ctl = c;
    }
    // (a range that goes straight to the sweep threads is read from `fin`)
    if (i < P.n && P.n <= GS_SWEEP_MAX) P.fin[i] = i;
    if (P.n > GS_WAVE_MAX && i < (P.n + GS_CHUNK - 1) / GS_CHUNK) P.chunks[0][i] = GsChunk{0, i};
}

__global__ void gs_clear_lists_kernel(GsControl* ctl, int parity) {
    ctl->n_big[parity] = 0;
    ctl->n_wave[parity] = 0;
    ctl->n_chunks[parity] = 0;
}

// ---------------------------------------------------------------- pieces shared by the wave path and the chunk path
// bins: GS_BIN_WORDS words (LDS or global); bin (a, b) = bins + (a * GS_BINS + b) * 7
__device__ __forceinline__ void gs_bins_clear(uint32_t* bins, int tid, int stride) {
    for (int k = tid; k < GS_BIN_WORDS; k += stride) bins[k] = (k % 7) < 3 ? 0xffffffffu : 0u;
}

__device__ __forceinline__ void gs_scales(const uint32_t* ext, float* clo, float* scale) {
#pragma clang fp contract(off)
    for (int a = 0; a < GS_AXES; ++a) {
        clo[a] = o2f(ext[a]);
        const float chi = o2f(ext[GS_AXES + a]);
        scale[a] = chi > clo[a] ? (float)GS_BINS * (1.f - 1e-6f) / (chi - clo[a]) : 0.f;
    }
}

// primitive p into LDS bins
__device__ __forceinline__ void gs_bin_add(const GsParams& P, uint32_t* bins, int p, const float* clo, const float* scale) {
    uint32_t bl[3], bh[3];
    for (int k = 0; k < 3; ++k) { bl[k] = f2o(P.lo[3 * p + k]); bh[k] = f2o(P.hi[3 * p + k]); }
    for (int a = 0; a < GS_AXES; ++a) {
        uint32_t* bin = bins + (a * GS_BINS + gs_bin(gs_coord(P, p, a), clo[a], scale[a])) * 7;
        atomicMin(&bin[0], bl[0]); atomicMin(&bin[1], bl[1]); atomicMin(&bin[2], bl[2]);
        atomicMax(&bin[3], bh[0]); atomicMax(&bin[4], bh[1]); atomicMax(&bin[5], bh[2]);
        atomicAdd(&bin[6], 1u);
    }
}

// union of the bins [b0, b1] of coordinate a; returns the primitive count
__device__ __forceinline__ int gs_bins_union(const uint32_t* bins, int a, int b0, int b1, float* l, float* h) {
    uint32_t ul[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, uh[3] = {0u, 0u, 0u};
    int cnt = 0;
    for (int b = b0; b <= b1; ++b) {
        const uint32_t* bin = bins + (a * GS_BINS + b) * 7;
        if (bin[6] == 0u) continue;
        cnt += (int)bin[6];
        for (int k = 0; k < 3; ++k) { ul[k] = min(ul[k], bin[k]); uh[k] = max(uh[k], bin[3 + k]); }
    }
    for (int k = 0; k < 3; ++k) { l[k] = o2f(ul[k]); h[k] = o2f(uh[k]); }
    return cnt;
}

// The 64 lanes of a wave price the 64 cuts (lane = coordinate * 16 + last bin of the left part) and agree on the
// cheapest; among equals the lowest lane, the order in which a serial loop with `<` meets them.  -1: no cut has a cost
// (every coordinate of every primitive coincides).
__device__ __forceinline__ int gs_price_cuts(const uint32_t* bins, const float* scale, int lane) {
#pragma clang fp contract(off)
    const int a = lane >> 4, b = lane & 15;
    float cost = 3.0e38f;
    if (b < GS_BINS - 1 && scale[a] > 0.f) {
        float l0[3], h0[3], l1[3], h1[3];
        const int nl = gs_bins_union(bins, a, 0, b, l0, h0);
        const int nr = gs_bins_union(bins, a, b + 1, GS_BINS - 1, l1, h1);
        if (nl > 0 && nr > 0) cost = gs_half_area(l0, h0) * (float)nl + gs_half_area(l1, h1) * (float)nr;
    }
    float bc = cost;
    int bl = lane;
    for (int off = 32; off > 0; off >>= 1) {
        const float oc = __shfl_xor(bc, off);
        const int ol = __shfl_xor(bl, off);
        if (oc < bc || (oc == bc && ol < bl)) { bc = oc; bl = ol; }
    }
    return bc < 3.0e38f ? bl : -1;
}

// The node of task t, cut after nl primitives, and what becomes of its two parts.  Called by ONE wave (all 64 lanes):
// lane 0 writes, the lanes share the chunk table of a child that is a large task again.
__device__ __forceinline__ void gs_finish_task(const GsParams& P, const GsTask& t, int nl, const float* box_l, const float* box_h,
                                               const float* box_rl, const float* box_rh, int parity, int level, int lane) {
    const int next = parity ^ 1;
    const GsTask child[2] = {GsTask{t.node + 1, t.first, nl, 0}, GsTask{t.node + nl, t.first + nl, t.count - nl, 0}};
    int link[2];
    for (int c = 0; c < 2; ++c) {
        if (child[c].count == 1) {
            link[c] = ~child[c].first;  // one primitive: slot | (1 - 1) << BVH_SLOT_BITS
            continue;
        }
        link[c] = child[c].node;
        if (child[c].count <= GS_SWEEP_MAX) {
            if (lane == 0) {
                const int s = atomicAdd(&P.ctl->n_sweep, 1);
                if (s < P.cap_sweep) P.sweep[s] = child[c]; else P.ctl->error = 1;
            }
        } else if (child[c].count <= GS_WAVE_MAX) {
            if (lane == 0) {
                const int s = atomicAdd(&P.ctl->n_wave[next], 1);
                if (s < P.cap_wave) P.wave[next][s] = child[c]; else P.ctl->error = 1;
            }
        } else {
            const int nc = (child[c].count + GS_CHUNK - 1) / GS_CHUNK;
            int s = 0, c0 = 0;
            if (lane == 0) {
                s = atomicAdd(&P.ctl->n_big[next], 1);
                c0 = atomicAdd(&P.ctl->n_chunks[next], nc);
            }
            s = __shfl(s, 0);
            c0 = __shfl(c0, 0);
            if (s < P.cap_big && c0 + nc <= P.cap_chunks) {
                if (lane == 0) {
                    P.big[next][s] = child[c];
                    P.big_state[next][s].chunk0 = c0;
                    P.big_state[next][s].n_chunks = nc;
                }
                for (int j = lane; j < nc; j += 64) P.chunks[next][c0 + j] = GsChunk{s, j};
            } else if (lane == 0) {
                P.ctl->error = 1;
            }
        }
    }
    if (lane == 0) {
        BvhNode N;
        for (int k = 0; k < 3; ++k) { N.lo0[k] = box_l[k]; N.hi0[k] = box_h[k]; N.lo1[k] = box_rl[k]; N.hi1[k] = box_rh[k]; }
        N.c0 = link[0];
        N.c1 = link[1];
        N.pad0 = N.pad1 = 0;
        P.nodes[t.node] = N;
        const int k = atomicAdd(&P.ctl->n_top, 1);
        if (k < P.cap_top) { P.top_node[k] = t.node; P.top_level[k] = level; } else P.ctl->error = 1;
        atomicMax(&P.ctl->max_level, level);
    }
}

// ---------------------------------------------------------------- ranges of up to GS_WAVE_MAX primitives: one wave each
// Splits the range [first, first + count) of idx[parity] and writes the two parts to idx[parity ^ 1].
__global__ void __launch_bounds__(64) gs_wave_level_kernel(GsParams P, int parity, int level) {
#pragma clang fp contract(off)
    __shared__ uint32_t s_ext[2 * GS_AXES];
    __shared__ uint32_t s_bins[GS_BIN_WORDS];
    __shared__ uint32_t s_part[2][6];  // degenerate ranges: the boxes of the two halves
    if ((int)blockIdx.x >= P.ctl->n_wave[parity]) return;
    const GsTask t = P.wave[parity][blockIdx.x];
    const int* idx_in = P.idx[parity];
    int* idx_out = P.idx[parity ^ 1];
    const int lane = threadIdx.x;

    // -- the extent of the four cut coordinates
    if (lane < GS_AXES) { s_ext[lane] = 0xffffffffu; s_ext[GS_AXES + lane] = 0u; }
    if (lane < 12) s_part[lane / 6][lane % 6] = (lane % 6) < 3 ? 0xffffffffu : 0u;
    gs_bins_clear(s_bins, lane, 64);
    __syncthreads();
    {
        float mlo[GS_AXES], mhi[GS_AXES];
        for (int a = 0; a < GS_AXES; ++a) { mlo[a] = 3.0e38f; mhi[a] = -3.0e38f; }
        for (int i = lane; i < t.count; i += 64) {
            const int p = idx_in[t.first + i];
            for (int a = 0; a < GS_AXES; ++a) {
                const float c = gs_coord(P, p, a);
                mlo[a] = fminf(mlo[a], c);
                mhi[a] = fmaxf(mhi[a], c);
            }
        }
        for (int a = 0; a < GS_AXES; ++a) {
            atomicMin(&s_ext[a], f2o(mlo[a]));
            atomicMax(&s_ext[GS_AXES + a], f2o(mhi[a]));
        }
    }
    __syncthreads();
    float clo[GS_AXES], scale[GS_AXES];
    gs_scales(s_ext, clo, scale);

    // -- the bins, the cut
    for (int i = lane; i < t.count; i += 64) gs_bin_add(P, s_bins, idx_in[t.first + i], clo, scale);
    __syncthreads();
    const int best = gs_price_cuts(s_bins, scale, lane);
    const bool degenerate = best < 0;  // cut the order in the middle
    const int axis = degenerate ? 0 : best >> 4, cut = degenerate ? 0 : best & 15;
    float box_l[3], box_h[3], box_rl[3], box_rh[3];
    int nl;
    if (!degenerate) {
        nl = gs_bins_union(s_bins, axis, 0, cut, box_l, box_h);
        (void)gs_bins_union(s_bins, axis, cut + 1, GS_BINS - 1, box_rl, box_rh);
    } else {
        nl = t.count / 2;
        for (int i = lane; i < t.count; i += 64) {
            const int p = idx_in[t.first + i];
            uint32_t* part = s_part[i < nl ? 0 : 1];
            for (int k = 0; k < 3; ++k) { atomicMin(&part[k], f2o(P.lo[3 * p + k])); atomicMax(&part[3 + k], f2o(P.hi[3 * p + k])); }
        }
        __syncthreads();
        for (int k = 0; k < 3; ++k) {
            box_l[k] = o2f(s_part[0][k]); box_h[k] = o2f(s_part[0][3 + k]);
            box_rl[k] = o2f(s_part[1][k]); box_rh[k] = o2f(s_part[1][3 + k]);
        }
    }
    const int nr = t.count - nl;

    // -- stable partition into the other buffer; a part that leaves the levels here (a leaf, or a range for the sweep
    //    threads) also goes to the final order
    const bool fin_l = nl <= GS_SWEEP_MAX, fin_r = nr <= GS_SWEEP_MAX;
    int run_l = 0, run_r = 0;
    for (int base = 0; base < t.count; base += 64) {
        const int i = base + lane;
        const bool active = i < t.count;
        const int p = active ? idx_in[t.first + i] : 0;
        const bool left = active && (degenerate ? i < nl : gs_bin(gs_coord(P, p, axis), clo[axis], scale[axis]) <= cut);
        const bool right = active && !left;
        const unsigned long long ml = __ballot(left), mr = __ballot(right);
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        if (left) {
            const int at = t.first + run_l + __popcll(ml & below);
            idx_out[at] = p;
            if (fin_l) P.fin[at] = p;
        }
        if (right) {
            const int at = t.first + nl + run_r + __popcll(mr & below);
            idx_out[at] = p;
            if (fin_r) P.fin[at] = p;
        }
        run_l += __popcll(ml);
        run_r += __popcll(mr);
    }
    gs_finish_task(P, t, nl, box_l, box_h, box_rl, box_rh, parity, level, lane);
}

// ---------------------------------------------------------------- larger ranges: one work-group per chunk
__global__ void __launch_bounds__(256) gs_big_reset_kernel(GsParams P, int parity) {
    if ((int)blockIdx.x >= P.ctl->n_big[parity]) return;
    GsBigState& S = P.big_state[parity][blockIdx.x];
    const int tid = threadIdx.x;
    if (tid < 2 * GS_AXES) S.ext[tid] = tid < GS_AXES ? 0xffffffffu : 0u;
    if (tid < 12) S.part[tid / 6][tid % 6] = (tid % 6) < 3 ? 0xffffffffu : 0u;
    gs_bins_clear(S.bins, tid, 256);
}

__global__ void __launch_bounds__(GS_BIG_BLOCK) gs_big_extent_kernel(GsParams P, int parity) {
    __shared__ uint32_t s_ext[2 * GS_AXES];
    if ((int)blockIdx.x >= P.ctl->n_chunks[parity]) return;
    const GsChunk c = P.chunks[parity][blockIdx.x];
    const GsTask t = P.big[parity][c.task];
    const int* idx_in = P.idx[parity];
    const int tid = threadIdx.x;
    const int begin = c.index * GS_CHUNK, end = min(begin + GS_CHUNK, t.count);
    if (tid < GS_AXES) { s_ext[tid] = 0xffffffffu; s_ext[GS_AXES + tid] = 0u; }
    __syncthreads();
    float mlo[GS_AXES], mhi[GS_AXES];
    for (int a = 0; a < GS_AXES; ++a) { mlo[a] = 3.0e38f; mhi[a] = -3.0e38f; }
    for (int i = begin + tid; i < end; i += GS_BIG_BLOCK) {
        const int p = idx_in[t.first + i];
        for (int a = 0; a < GS_AXES; ++a) {
            const float v = gs_coord(P, p, a);
            mlo[a] = fminf(mlo[a], v);
            mhi[a] = fmaxf(mhi[a], v);
        }
    }
    // the wave's extent first (shuffles), then one LDS atomic per wave and value, then one global atomic per chunk and value
    for (int a = 0; a < GS_AXES; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            mlo[a] = fminf(mlo[a], __shfl_xor(mlo[a], off));
            mhi[a] = fmaxf(mhi[a], __shfl_xor(mhi[a], off));
        }
    if ((tid & 63) == 0)
        for (int a = 0; a < GS_AXES; ++a) {
            atomicMin(&s_ext[a], f2o(mlo[a]));
            atomicMax(&s_ext[GS_AXES + a], f2o(mhi[a]));
        }
    __syncthreads();
    GsBigState& S = P.big_state[parity][c.task];
    if (tid < GS_AXES) atomicMin(&S.ext[tid], s_ext[tid]);
    else if (tid < 2 * GS_AXES) atomicMax(&S.ext[tid], s_ext[tid]);
}

__global__ void __launch_bounds__(GS_BIG_BLOCK) gs_big_bins_kernel(GsParams P, int parity) {
    __shared__ uint32_t s_bins[GS_BIN_WORDS];
    if ((int)blockIdx.x >= P.ctl->n_chunks[parity]) return;
    const GsChunk c = P.chunks[parity][blockIdx.x];
    const GsTask t = P.big[parity][c.task];
    GsBigState& S = P.big_state[parity][c.task];
    const int* idx_in = P.idx[parity];
    const int tid = threadIdx.x;
    const int begin = c.index * GS_CHUNK, end = min(begin + GS_CHUNK, t.count);
    gs_bins_clear(s_bins, tid, GS_BIG_BLOCK);
    float clo[GS_AXES], scale[GS_AXES];
    gs_scales(S.ext, clo, scale);
    __syncthreads();
    for (int i = begin + tid; i < end; i += GS_BIG_BLOCK) gs_bin_add(P, s_bins, idx_in[t.first + i], clo, scale);
    __syncthreads();
    // the chunk's bins into the task's: only bins that hold something
    for (int k = tid; k < GS_BIN_WORDS; k += GS_BIG_BLOCK) {
        const int r = k % 7;
        if (s_bins[k - r + 6] == 0u) continue;
        if (r < 3) atomicMin(&S.bins[k], s_bins[k]);
        else if (r < 6) atomicMax(&S.bins[k], s_bins[k]);
        else atomicAdd(&S.bins[k], s_bins[k]);
    }
}

// one wave per task: the cut
__global__ void __launch_bounds__(64) gs_big_cut_kernel(GsParams P, int parity) {
    __shared__ uint32_t s_bins[GS_BIN_WORDS];
    if ((int)blockIdx.x >= P.ctl->n_big[parity]) return;
    const GsTask t = P.big[parity][blockIdx.x];
    GsBigState& S = P.big_state[parity][blockIdx.x];
    const int lane = threadIdx.x;
    for (int k = lane; k < GS_BIN_WORDS; k += 64) s_bins[k] = S.bins[k];
    float clo[GS_AXES], scale[GS_AXES];
    gs_scales(S.ext, clo, scale);
    __syncthreads();
    const int best = gs_price_cuts(s_bins, scale, lane);
    if (lane == 0) {
        float l[3], h[3];
        S.best = best;
        S.nl = best < 0 ? t.count / 2 : gs_bins_union(s_bins, best >> 4, 0, best & 15, l, h);
    }
}

__device__ __forceinline__ bool gs_big_goes_left(const GsParams& P, const GsBigState& S, int p, int i, const float* clo, const float* scale) {
    return S.best < 0 ? i < S.nl : gs_bin(gs_coord(P, p, S.best >> 4), clo[S.best >> 4], scale[S.best >> 4]) <= (S.best & 15);
}

// per chunk: how many of its primitives go left (and, for a degenerate range, the boxes of the two halves)
__global__ void __launch_bounds__(GS_BIG_BLOCK) gs_big_count_kernel(GsParams P, int parity) {
    __shared__ int s_left;
    __shared__ uint32_t s_part[2][6];
    if ((int)blockIdx.x >= P.ctl->n_chunks[parity]) return;
    const GsChunk c = P.chunks[parity][blockIdx.x];
    const GsTask t = P.big[parity][c.task];
    GsBigState& S = P.big_state[parity][c.task];
    const int* idx_in = P.idx[parity];
    const int tid = threadIdx.x;
    const int begin = c.index * GS_CHUNK, end = min(begin + GS_CHUNK, t.count);
    if (tid == 0) s_left = 0;
    if (tid < 12) s_part[tid / 6][tid % 6] = (tid % 6) < 3 ? 0xffffffffu : 0u;
    float clo[GS_AXES], scale[GS_AXES];
    gs_scales(S.ext, clo, scale);
    const bool degenerate = S.best < 0;
    __syncthreads();
    int mine = 0;
    for (int i = begin + tid; i < end; i += GS_BIG_BLOCK) {
        const int p = idx_in[t.first + i];
        const bool left = gs_big_goes_left(P, S, p, i, clo, scale);
        mine += left ? 1 : 0;
        if (degenerate) {
            uint32_t* part = s_part[left ? 0 : 1];
            for (int k = 0; k < 3; ++k) { atomicMin(&part[k], f2o(P.lo[3 * p + k])); atomicMax(&part[3 + k], f2o(P.hi[3 * p + k])); }
        }
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if ((tid & 63) == 0 && mine) atomicAdd(&s_left, mine);
    __syncthreads();
    if (tid == 0) P.chunk_left[blockIdx.x] = s_left;
    if (degenerate && tid < 12) {
        if ((tid % 6) < 3) atomicMin(&S.part[tid / 6][tid % 6], s_part[tid / 6][tid % 6]);
        else atomicMax(&S.part[tid / 6][tid % 6], s_part[tid / 6][tid % 6]);
    }
}

// one wave per task: where every chunk's left and right parts start, the node, the children
__global__ void __launch_bounds__(64) gs_big_finish_kernel(GsParams P, int parity, int level) {
    __shared__ uint32_t s_bins[GS_BIN_WORDS];
    if ((int)blockIdx.x >= P.ctl->n_big[parity]) return;
    const GsTask t = P.big[parity][blockIdx.x];
    GsBigState& S = P.big_state[parity][blockIdx.x];
    const int lane = threadIdx.x;
    for (int k = lane; k < GS_BIN_WORDS; k += 64) s_bins[k] = S.bins[k];
    __syncthreads();
    // exclusive prefix of the chunks' left counts (a task has count / GS_CHUNK chunks: a few hundred at most)
    if (lane == 0) {
        int run = 0;
        for (int j = 0; j < S.n_chunks; ++j) {
            const int v = P.chunk_left[S.chunk0 + j];
            P.chunk_left[S.chunk0 + j] = run;
            run += v;
        }
        if (run != S.nl) P.ctl->error = 1;
    }
    float box_l[3], box_h[3], box_rl[3], box_rh[3];
    if (S.best >= 0) {
        (void)gs_bins_union(s_bins, S.best >> 4, 0, S.best & 15, box_l, box_h);
        (void)gs_bins_union(s_bins, S.best >> 4, (S.best & 15) + 1, GS_BINS - 1, box_rl, box_rh);
    } else {
        for (int k = 0; k < 3; ++k) {
            box_l[k] = o2f(S.part[0][k]); box_h[k] = o2f(S.part[0][3 + k]);
            box_rl[k] = o2f(S.part[1][k]); box_rh[k] = o2f(S.part[1][3 + k]);
        }
    }
    gs_finish_task(P, t, S.nl, box_l, box_h, box_rl, box_rh, parity, level, lane);
}

// per chunk: its primitives to their places in the other buffer (stable: chunk after chunk, in order inside a chunk)
__global__ void __launch_bounds__(GS_BIG_BLOCK) gs_big_scatter_kernel(GsParams P, int parity) {
    constexpr int WAVES = GS_BIG_BLOCK / 64;
    __shared__ int s_wave_left[WAVES], s_wave_right[WAVES];
    __shared__ int s_run[2];
    if ((int)blockIdx.x >= P.ctl->n_chunks[parity]) return;
    const GsChunk c = P.chunks[parity][blockIdx.x];
    const GsTask t = P.big[parity][c.task];
    const GsBigState& S = P.big_state[parity][c.task];
    const int* idx_in = P.idx[parity];
    int* idx_out = P.idx[parity ^ 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int begin = c.index * GS_CHUNK, end = min(begin + GS_CHUNK, t.count);
    const int nl = S.nl, nr = t.count - nl;
    const int left_before = P.chunk_left[blockIdx.x];  // of the task's earlier chunks
    const bool fin_l = nl <= GS_SWEEP_MAX, fin_r = nr <= GS_SWEEP_MAX;
    float clo[GS_AXES], scale[GS_AXES];
    gs_scales(S.ext, clo, scale);
    if (tid == 0) { s_run[0] = left_before; s_run[1] = begin - left_before; }
    __syncthreads();
    for (int base = begin; base < end; base += GS_BIG_BLOCK) {
        const int i = base + tid;
        const bool active = i < end;
        const int p = active ? idx_in[t.first + i] : 0;
        const bool left = active && gs_big_goes_left(P, S, p, i, clo, scale);
        const bool right = active && !left;
        const unsigned long long ml = __ballot(left), mr = __ballot(right);
        if (lane == 0) { s_wave_left[wave] = __popcll(ml); s_wave_right[wave] = __popcll(mr); }
        __syncthreads();
        int before_l = s_run[0], before_r = s_run[1];
        for (int w = 0; w < wave; ++w) { before_l += s_wave_left[w]; before_r += s_wave_right[w]; }
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        if (left) {
            const int at = t.first + before_l + __popcll(ml & below);
            idx_out[at] = p;
            if (fin_l) P.fin[at] = p;
        }
        if (right) {
            const int at = t.first + nl + before_r + __popcll(mr & below);
            idx_out[at] = p;
            if (fin_r) P.fin[at] = p;
        }
        __syncthreads();
        if (tid == 0) {
            int tl = 0, tr = 0;
            for (int w = 0; w < WAVES; ++w) { tl += s_wave_left[w]; tr += s_wave_right[w]; }
            s_run[0] += tl;
            s_run[1] += tr;
        }
        __syncthreads();
    }
}

// One thread per range of 2..GS_SWEEP_MAX primitives: the whole subtree, SahBuilder::partition_sweep cut by cut.
__global__ void __launch_bounds__(64) gs_sweep_kernel(GsParams P) {
#pragma clang fp contract(off)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.ctl->n_sweep) return;
    const GsTask t = P.sweep[j];
    int ids[GS_SWEEP_MAX];
    for (int k = 0; k < t.count; ++k) ids[k] = P.fin[t.first + k];
    struct Range { int node, off, cnt; } stack[GS_SWEEP_MAX];
    int sp = 0;
    stack[sp++] = Range{t.node, 0, t.count};
    float key[GS_SWEEP_MAX], right_area[GS_SWEEP_MAX];
    int kid[GS_SWEEP_MAX];
    auto grow = [&](float* l, float* h, int p) {
        for (int k = 0; k < 3; ++k) { l[k] = fminf(l[k], P.lo[3 * p + k]); h[k] = fmaxf(h[k], P.hi[3 * p + k]); }
    };
    auto clear = [](float* l, float* h) { for (int k = 0; k < 3; ++k) { l[k] = 3.0e38f; h[k] = -3.0e38f; } };
    while (sp > 0) {
        const Range r = stack[--sp];
        auto sort_along = [&](int a) {  // by (coordinate, primitive): insertion sort of at most 32 pairs
            for (int i = 0; i < r.cnt; ++i) {
                const int p = ids[r.off + i];
                const float c = gs_coord(P, p, a);
                int q = i;
                while (q > 0 && (key[q - 1] > c || (key[q - 1] == c && kid[q - 1] > p))) {
                    key[q] = key[q - 1];
                    kid[q] = kid[q - 1];
                    --q;
                }
                key[q] = c;
                kid[q] = p;
            }
        };
        float best_cost = 3.0e38f;
        int best_nl = r.cnt / 2, best_axis = -1;
        for (int a = 0; a < GS_AXES; ++a) {
            sort_along(a);
            float l[3], h[3];
            clear(l, h);
            for (int i = r.cnt - 1; i > 0; --i) {
                grow(l, h, kid[i]);
                right_area[i] = gs_half_area(l, h);
            }
            clear(l, h);
            for (int i = 0; i < r.cnt - 1; ++i) {
                grow(l, h, kid[i]);
                const int nl = i + 1, nr = r.cnt - nl;
                // (one primitive per leaf: a part of one primitive costs its box once)
                const float cost = gs_half_area(l, h) * (nl <= 1 ? 1.f : (float)nl) + right_area[i + 1] * (nr <= 1 ? 1.f : (float)nr);
                if (cost < best_cost) { best_cost = cost; best_nl = nl; best_axis = a; }
            }
        }
        if (best_axis >= 0) {
            if (best_axis != GS_AXES - 1) sort_along(best_axis);
            for (int k = 0; k < r.cnt; ++k) ids[r.off + k] = kid[k];
        }
        const int nl = best_nl, nr = r.cnt - nl;
        BvhNode N;
        float l[3], h[3];
        clear(l, h);
        for (int k = 0; k < nl; ++k) grow(l, h, ids[r.off + k]);
        for (int k = 0; k < 3; ++k) { N.lo0[k] = l[k]; N.hi0[k] = h[k]; }
        clear(l, h);
        for (int k = nl; k < r.cnt; ++k) grow(l, h, ids[r.off + k]);
        for (int k = 0; k < 3; ++k) { N.lo1[k] = l[k]; N.hi1[k] = h[k]; }
        N.c0 = nl == 1 ? ~(t.first + r.off) : r.node + 1;
        N.c1 = nr == 1 ? ~(t.first + r.off + nl) : r.node + nl;
        N.pad0 = N.pad1 = 0;
        P.nodes[r.node] = N;
        if (nl >= 2) stack[sp++] = Range{r.node + 1, r.off, nl};
        if (nr >= 2) stack[sp++] = Range{r.node + nl, r.off + nl, nr};
    }
    for (int k = 0; k < t.count; ++k) P.fin[t.first + k] = ids[k];
    // the wide-node marks of this subtree: its nodes are t.node .. t.node + count - 2, children above their parents
    for (int k = t.node + t.count - 2; k >= t.node; --k) sah_mark_node(P.nodes, P.f, (size_t)k);
}

__global__ void __launch_bounds__(256) gs_mark_level_kernel(GsParams P, int level) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.ctl->n_top || P.top_level[i] != level) return;
    sah_mark_node(P.nodes, P.f, (size_t)P.top_node[i]);
}

}  // namespace

// d_nodes: n - 1 binary nodes (node 0 the root, one primitive per leaf, wide-node marks in BvhNode::pad0);
// d_prim_of_slot: n entries, sorted slot -> index into the bounds arrays.  d_size[i] = log2 of the half area of box i
// (the builder's fourth cut coordinate; computed by the host so that both builders bin the very same numbers).
// *levels = levels of tasks the build went through.  Needs n >= 2.  Synchronises the stream.
hipError_t gsah_build(const float* d_lo, const float* d_hi, const float* d_size, int n, BvhNode* d_nodes, int* d_prim_of_slot,
                      hipStream_t stream, int* levels) {
    if (n < 2 || n > BVH_MAX_PRIMS) return hipErrorInvalidValue;
    hipError_t e;
    GsParams P{};
    P.lo = d_lo; P.hi = d_hi; P.size = d_size; P.n = n;
    P.nodes = d_nodes;
    P.fin = d_prim_of_slot;
    // a level holds at most n / (smallest range of the class + 1) tasks of a class; two children per task
    P.cap_big = n / GS_WAVE_MAX + 2;
    P.cap_wave = n / GS_SWEEP_MAX + 2;
    P.cap_sweep = n / 2 + 2;
    P.cap_top = n;  // every task makes one of the tree's n - 1 nodes
    P.cap_chunks = n / GS_CHUNK + P.cap_big + 2;  // every large task ends in at most one chunk that is not full
    void* mem[20] = {};
    auto cleanup = [&]() { for (void* m : mem) (void)hipFree(m); };
#define GS_TRY(x) do { e = (x); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    GS_TRY(hipMalloc(&mem[0], sizeof(int) * (size_t)n)); P.idx[0] = (int*)mem[0];
    GS_TRY(hipMalloc(&mem[1], sizeof(int) * (size_t)n)); P.idx[1] = (int*)mem[1];
    GS_TRY(hipMalloc(&mem[2], sizeof(float) * (size_t)n)); P.f = (float*)mem[2];
    GS_TRY(hipMalloc(&mem[3], sizeof(GsTask) * (size_t)P.cap_big)); P.big[0] = (GsTask*)mem[3];
    GS_TRY(hipMalloc(&mem[4], sizeof(GsTask) * (size_t)P.cap_big)); P.big[1] = (GsTask*)mem[4];
    GS_TRY(hipMalloc(&mem[5], sizeof(GsTask) * (size_t)P.cap_wave)); P.wave[0] = (GsTask*)mem[5];
    GS_TRY(hipMalloc(&mem[6], sizeof(GsTask) * (size_t)P.cap_wave)); P.wave[1] = (GsTask*)mem[6];
    GS_TRY(hipMalloc(&mem[7], sizeof(GsTask) * (size_t)P.cap_sweep)); P.sweep = (GsTask*)mem[7];
    GS_TRY(hipMalloc(&mem[8], sizeof(int) * (size_t)P.cap_top)); P.top_node = (int*)mem[8];
    GS_TRY(hipMalloc(&mem[9], sizeof(int) * (size_t)P.cap_top)); P.top_level = (int*)mem[9];
    GS_TRY(hipMalloc(&mem[10], sizeof(GsControl))); P.ctl = (GsControl*)mem[10];
    GS_TRY(hipMalloc(&mem[11], sizeof(GsChunk) * (size_t)P.cap_chunks)); P.chunks[0] = (GsChunk*)mem[11];
    GS_TRY(hipMalloc(&mem[12], sizeof(GsChunk) * (size_t)P.cap_chunks)); P.chunks[1] = (GsChunk*)mem[12];
    GS_TRY(hipMalloc(&mem[13], sizeof(GsBigState) * (size_t)P.cap_big)); P.big_state[0] = (GsBigState*)mem[13];
    GS_TRY(hipMalloc(&mem[14], sizeof(GsBigState) * (size_t)P.cap_big)); P.big_state[1] = (GsBigState*)mem[14];
    GS_TRY(hipMalloc(&mem[15], sizeof(int) * (size_t)P.cap_chunks)); P.chunk_left = (int*)mem[15];
    GS_TRY(hipMemsetAsync(P.f, 0, sizeof(float) * (size_t)n, stream));
    hipLaunchKernelGGL(gs_init_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, P);
    GS_TRY(hipGetLastError());
    GsControl ctl{};
    int level = 0;
    bool large_left = n > GS_WAVE_MAX;  // ranges only shrink: once a level has no large task, none follows
    for (;; ++level) {
        if (level > 4096) GS_TRY(hipErrorUnknown);
        const int parity = level & 1;
        hipLaunchKernelGGL(gs_clear_lists_kernel, dim3(1), dim3(1), 0, stream, P.ctl, parity ^ 1);
        // level L holds at most 2^L tasks, and never more than the lists hold
        const long long bound = level < 30 ? (1ll << level) : (1ll << 30);
        const int g_big = (int)(bound < P.cap_big ? bound : P.cap_big), g_wave = (int)(bound < P.cap_wave ? bound : P.cap_wave);
        if (large_left) {
            const dim3 tasks(g_big), chunks(P.cap_chunks);
            hipLaunchKernelGGL(gs_big_reset_kernel, tasks, dim3(256), 0, stream, P, parity);
            hipLaunchKernelGGL(gs_big_extent_kernel, chunks, dim3(GS_BIG_BLOCK), 0, stream, P, parity);
            hipLaunchKernelGGL(gs_big_bins_kernel, chunks, dim3(GS_BIG_BLOCK), 0, stream, P, parity);
            hipLaunchKernelGGL(gs_big_cut_kernel, tasks, dim3(64), 0, stream, P, parity);
            hipLaunchKernelGGL(gs_big_count_kernel, chunks, dim3(GS_BIG_BLOCK), 0, stream, P, parity);
            hipLaunchKernelGGL(gs_big_finish_kernel, tasks, dim3(64), 0, stream, P, parity, level);
            hipLaunchKernelGGL(gs_big_scatter_kernel, chunks, dim3(GS_BIG_BLOCK), 0, stream, P, parity);
        }
        hipLaunchKernelGGL(gs_wave_level_kernel, dim3(g_wave), dim3(64), 0, stream, P, parity, level);
        GS_TRY(hipGetLastError());
        if (level % 8 == 7 || level < 1) {  // are there tasks left?  (asked every eighth level: a million primitives take ~25)
            GS_TRY(hipMemcpyAsync(&ctl, P.ctl, sizeof(ctl), hipMemcpyDeviceToHost, stream));
            GS_TRY(hipStreamSynchronize(stream));
            if (ctl.error) GS_TRY(hipErrorUnknown);
            if (ctl.n_big[parity ^ 1] == 0) large_left = false;
            if (ctl.n_big[parity ^ 1] == 0 && ctl.n_wave[parity ^ 1] == 0) break;
        }
    }
    if (ctl.n_sweep > 0) {
        hipLaunchKernelGGL(gs_sweep_kernel, dim3((ctl.n_sweep + 63) / 64), dim3(64), 0, stream, P);
        GS_TRY(hipGetLastError());
    }
    for (int l = ctl.max_level; l >= 0 && ctl.n_top > 0; --l) {
        hipLaunchKernelGGL(gs_mark_level_kernel, dim3((ctl.n_top + 255) / 256), dim3(256), 0, stream, P, l);
        GS_TRY(hipGetLastError());
    }
    GS_TRY(hipStreamSynchronize(stream));
    *levels = ctl.n_top > 0 ? ctl.max_level + 1 : 0;
#undef GS_TRY
    cleanup();
    return hipSuccess;
}

}  // namespace rrt
