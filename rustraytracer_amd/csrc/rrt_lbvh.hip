// GPU construction of a linear BVH (LBVH) over the bounded primitives of a scene.
//
//   1. host: f64 bounds of every primitive, rounded outward to f32 and padded (rrt_api.cpp)
//   2. morton_kernel   63-bit Morton code of the box centre (21 bits per axis)
//   3. rocPRIM radix sort of (code, primitive) pairs; equal codes are ordered by sorted position
//   4. hierarchy_kernel Karras' parallel binary radix tree: one thread per internal node
//   5. refit_kernel    bottom-up AABB fit, one thread per leaf, second arrival at a node continues
//
//   6. collapse_kernel subtrees of a few primitives become single leaves (slot ranges)
//   7. widen_kernel    every binary node with its children expanded: the 4-wide nodes rays traverse
//   8. reachable/compact the wide nodes rays can reach, packed to the front of the node array
// Steps 7 and 8 also serve binary trees built on the host (bvh_widen_host_tree; rrt_sah.hpp).
//
// A binary node holds the boxes of BOTH children; child < 0 encodes a leaf, see BvhNode;
// prim_of_slot[slot] = index into the bounds arrays.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>  // rocPRIM uses ::memset in host code without including it

#include <rocprim/rocprim.hpp>

#include "rrt_lbvh.hpp"

namespace rrt {

// 21 bits -> every third bit of 63
__device__ __forceinline__ unsigned long long expand_bits21(unsigned long long x) {
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void __launch_bounds__(256)
morton_kernel(const float* __restrict__ lo, const float* __restrict__ hi, int n, float3 scene_lo, float3 inv_extent,
              unsigned long long* __restrict__ keys, unsigned int* __restrict__ index) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float cx = 0.5f * (lo[3 * i] + hi[3 * i]), cy = 0.5f * (lo[3 * i + 1] + hi[3 * i + 1]),
          cz = 0.5f * (lo[3 * i + 2] + hi[3 * i + 2]);
    const float cells = 2097152.f;  // 2^21 per axis
    float nx = fminf(fmaxf((cx - scene_lo.x) * inv_extent.x * cells, 0.f), cells - 1.f);
    float ny = fminf(fmaxf((cy - scene_lo.y) * inv_extent.y * cells, 0.f), cells - 1.f);
    float nz = fminf(fmaxf((cz - scene_lo.z) * inv_extent.z * cells, 0.f), cells - 1.f);
    keys[i] = (expand_bits21((unsigned long long)nx) << 2) | (expand_bits21((unsigned long long)ny) << 1) |
              expand_bits21((unsigned long long)nz);
    index[i] = (unsigned int)i;
}

// Length of the common prefix of sorted keys i and j; equal codes are told apart by the position
// itself (Karras 2012, section 4), so the tree is well defined for any input.
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    unsigned long long x = keys[i] ^ keys[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (2012), fig. 4.
// Also records the range of sorted slots under every internal node.
__global__ void __launch_bounds__(256)
hierarchy_kernel(const unsigned long long* __restrict__ keys, int n, BvhNode* __restrict__ nodes,
                 int* __restrict__ node_parent, int* __restrict__ leaf_parent, int* __restrict__ range_first,
                 int* __restrict__ range_count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + (d < 0 ? d : 0);
    int lo_i = i < j ? i : j, hi_i = i < j ? j : i;
    int left = lo_i == gamma ? ~gamma : gamma;
    int right = hi_i == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    nodes[i].c0 = left;
    nodes[i].c1 = right;
    range_first[i] = lo_i;
    range_count[i] = hi_i - lo_i + 1;
    if (left < 0) leaf_parent[~left] = i; else node_parent[left] = i;
    if (right < 0) leaf_parent[~right] = i; else node_parent[right] = i;
    if (i == 0) node_parent[0] = -1;
}

// Folds every subtree of at most leaf_max primitives into one leaf of its parent: the child link
// becomes the subtree's slot range, the child box (already the union of the subtree) stays.
__global__ void __launch_bounds__(256)
collapse_kernel(BvhNode* __restrict__ nodes, int n, const int* __restrict__ range_first,
                const int* __restrict__ range_count, int leaf_max) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int c0 = nodes[i].c0, c1 = nodes[i].c1;
    if (c0 >= 0 && range_count[c0] <= leaf_max) nodes[i].c0 = ~(range_first[c0] | ((range_count[c0] - 1) << BVH_SLOT_BITS));
    if (c1 >= 0 && range_count[c1] <= leaf_max) nodes[i].c1 = ~(range_first[c1] | ((range_count[c1] - 1) << BVH_SLOT_BITS));
}

// Quantises up to four child boxes onto the node's own grid (see WideNode) and writes the node.
__device__ void write_wide_node(const float (*lo)[3], const float (*hi)[3], const int* child, int used, WideNode* out) {
    WideNode W;
    W.pad0 = 0;
    W.pad1[0] = W.pad1[1] = 0;
    for (int a = 0; a < 3; ++a) {
        float org = lo[0][a], top = hi[0][a];
        for (int c = 1; c < used; ++c) { org = fminf(org, lo[c][a]); top = fmaxf(top, hi[c][a]); }
        // smallest power of two s with org + 255 s >= top, in the traversal's own arithmetic
        int ex;
        (void)frexpf(fmaxf(top - org, 1e-30f) / 255.f, &ex);  // value = m 2^ex, m in [0.5, 1)
        int e = ex + 127;                                      // biased exponent of 2^ex >= value
        e = e < 1 ? 1 : (e > 254 ? 254 : e);
        while (e < 254 && fmaf(255.f, __uint_as_float((uint32_t)e << 23), org) < top) ++e;
        const float s = __uint_as_float((uint32_t)e << 23);
        uint32_t wlo = 0, whi = 0;
        for (int c = 0; c < 4; ++c) {
            int ql = 0, qh = 0;  // an empty slot decodes to a point box; its NO_CHILD link is what rejects it
            if (c < used) {
                ql = (int)floorf((lo[c][a] - org) / s);
                ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
                while (ql > 0 && fmaf((float)ql, s, org) > lo[c][a]) --ql;
                qh = (int)ceilf((hi[c][a] - org) / s);
                qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
                while (qh < 255 && fmaf((float)qh, s, org) < hi[c][a]) ++qh;
            }
            wlo |= (uint32_t)ql << (8 * c);
            whi |= (uint32_t)qh << (8 * c);
        }
        W.org[a] = org;
        W.e[a] = (uint8_t)e;
        W.qlo[a] = wlo;
        W.qhi[a] = whi;
    }
    for (int c = 0; c < 4; ++c) W.child[c] = c < used ? child[c] : BVH_NO_CHILD;
    *out = W;
}

// Wide node i = binary node i with both children expanded, boxes quantised (see WideNode).
__global__ void __launch_bounds__(256) widen_kernel(const BvhNode* __restrict__ nodes, int n, WideNode* __restrict__ wide) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const BvhNode N = nodes[i];
    float lo[4][3], hi[4][3];
    int child[4];
    int k = 0;
    auto put = [&](const float* l, const float* h, int c) {
        for (int a = 0; a < 3; ++a) { lo[k][a] = l[a]; hi[k][a] = h[a]; }
        child[k] = c;
        ++k;
    };
    if (N.c0 >= 0) {
        const BvhNode C = nodes[N.c0];
        put(C.lo0, C.hi0, C.c0);
        put(C.lo1, C.hi1, C.c1);
    } else {
        put(N.lo0, N.hi0, N.c0);
    }
    if (N.c1 >= 0) {
        const BvhNode C = nodes[N.c1];
        put(C.lo0, C.hi0, C.c0);
        put(C.lo1, C.hi1, C.c1);
    } else {
        put(N.lo1, N.hi1, N.c1);
    }
    write_wide_node(lo, hi, child, k, &wide[i]);
}

// Greedy widening: the children of wide node i are found by opening, again and again, the internal child with the
// largest surface area until there are four (or no internal child is left) -- a large child is the one rays are most
// likely to enter, so opening it saves the most steps; opening both children of the binary node one level, as
// widen_kernel does, spends slots on small children of lopsided subtrees.  Which binary nodes become wide nodes is
// then known only top-down: state[i] = 1 marks a node whose turn it is, 2 a finished one; every launch handles the
// marked nodes and marks their internal children, the host repeats it until a launch marks nothing.
// use_marks: the tree's builder has chosen the children itself (BvhNode::pad0, SAH_OPEN_* bits; rrt_sah.hpp).
__global__ void __launch_bounds__(256)
widen_greedy_kernel(const BvhNode* __restrict__ nodes, int n, WideNode* __restrict__ wide, int* __restrict__ state,
                    int* __restrict__ marked, int use_marks) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    if (__hip_atomic_load(&state[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1) return;
    float lo[4][3], hi[4][3];
    int child[4];
    int k = 0;
    auto put = [&](int at, const float* l, const float* h, int c) {
        for (int a = 0; a < 3; ++a) { lo[at][a] = l[a]; hi[at][a] = h[a]; }
        child[at] = c;
    };
    {
        const BvhNode N = nodes[i];
        put(0, N.lo0, N.hi0, N.c0);
        put(1, N.lo1, N.hi1, N.c1);
        k = 2;
    }
    if (use_marks) {
        const int32_t shape = nodes[i].pad0;
        auto open = [&](int at) {  // the child in slot `at` gives way to its two children; returns the slot of the second one
            const BvhNode C = nodes[child[at]];
            put(at, C.lo0, C.hi0, C.c0);
            put(k, C.lo1, C.hi1, C.c1);
            return k++;
        };
        int left1 = -1, right1 = -1;
        if ((shape & SAH_OPEN_LEFT) && child[0] >= 0) left1 = open(0);
        if ((shape & SAH_OPEN_RIGHT) && child[1] >= 0 && k < 4) right1 = open(1);
        if ((shape & SAH_OPEN_LEFT_0) && left1 >= 0 && child[0] >= 0 && k < 4) open(0);
        if ((shape & SAH_OPEN_LEFT_1) && left1 >= 0 && child[left1] >= 0 && k < 4) open(left1);
        if ((shape & SAH_OPEN_RIGHT_0) && right1 >= 0 && child[1] >= 0 && k < 4) open(1);
        if ((shape & SAH_OPEN_RIGHT_1) && right1 >= 0 && child[right1] >= 0 && k < 4) open(right1);
    }
    while (!use_marks && k < 4) {
        int best = -1;
        float best_area = -1.f;
        for (int c = 0; c < k; ++c) {
            if (child[c] < 0) continue;
            const float dx = hi[c][0] - lo[c][0], dy = hi[c][1] - lo[c][1], dz = hi[c][2] - lo[c][2];
            const float area = dx * dy + dy * dz + dz * dx;
            if (area > best_area) { best_area = area; best = c; }
        }
        if (best < 0) break;
        const BvhNode C = nodes[child[best]];
        put(best, C.lo0, C.hi0, C.c0);
        put(k, C.lo1, C.hi1, C.c1);
        ++k;
    }
    write_wide_node(lo, hi, child, k, &wide[i]);
    int any = 0;
    for (int c = 0; c < k; ++c)
        if (child[c] >= 0) {
            __hip_atomic_store(&state[child[c]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            any = 1;
        }
    __hip_atomic_store(&state[i], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (any) __hip_atomic_store(marked, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256) finished_flag_kernel(const int* __restrict__ state, int n, int* __restrict__ flag) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    flag[i] = state[i] == 2 ? 1 : 0;
}

// Wide node i is reached by rays iff binary node i sits at an even depth (odd levels are absorbed
// into their parents) and is not inside a folded subtree.
__global__ void __launch_bounds__(256)
reachable_kernel(const int* __restrict__ node_parent, const int* __restrict__ range_count, int n, int leaf_max,
                 int* __restrict__ flag) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int depth = 0;
    for (int p = node_parent[i]; p >= 0; p = node_parent[p]) ++depth;
    flag[i] = ((depth & 1) == 0 && (i == 0 || range_count[i] > leaf_max)) ? 1 : 0;
}

// Packs the reachable wide nodes to the front of `out`, in index order (which is the sorted order of
// the primitives, so a subtree stays contiguous), and renames the child links.
__global__ void __launch_bounds__(256)
compact_kernel(const WideNode* __restrict__ wide, const int* __restrict__ flag, const int* __restrict__ new_index, int n,
               WideNode* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || !flag[i]) return;
    WideNode W = wide[i];
    for (int c = 0; c < 4; ++c)
        if (W.child[c] >= 0) W.child[c] = new_index[W.child[c]];
    out[new_index[i]] = W;
}

template <typename R>
__global__ void __launch_bounds__(256)
pack_leaves_kernel(const PrimRec<R>* __restrict__ prims, const int* __restrict__ prim_of_slot, int n,
                   LeafRec<R>* __restrict__ out) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int g = prim_of_slot[s];
    LeafRec<R> r;
    for (int k = 0; k < 12; ++k) r.f[k] = prims[g].f[k];
    r.type = prims[g].type;
    r.g = g;
    r.pad[0] = r.pad[1] = 0;
    out[s] = r;
}

__global__ void __launch_bounds__(256)
refit_kernel(const float* __restrict__ lo, const float* __restrict__ hi, const int* __restrict__ prim_slot_src, int n,
             BvhNode* nodes, const int* __restrict__ node_parent, const int* __restrict__ leaf_parent,
             unsigned int* flags) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int prim = prim_slot_src[s];
    float bl[3] = {lo[3 * prim], lo[3 * prim + 1], lo[3 * prim + 2]};
    float bh[3] = {hi[3 * prim], hi[3 * prim + 1], hi[3 * prim + 2]};
    int cur = ~s;
    int p = leaf_parent[s];
    while (p >= 0) {
        BvhNode* N = &nodes[p];
        if (N->c0 == cur) {
            for (int k = 0; k < 3; ++k) { N->lo0[k] = bl[k]; N->hi0[k] = bh[k]; }
        } else {
            for (int k = 0; k < 3; ++k) { N->lo1[k] = bl[k]; N->hi1[k] = bh[k]; }
        }
        __threadfence();  // release: the box is visible before the arrival is
        unsigned int old = atomicAdd(&flags[p], 1u);
        if (old == 0u) return;  // first arrival: the sibling subtree finishes this node
        __threadfence();  // acquire: the sibling's box, written by another workgroup
        for (int k = 0; k < 3; ++k) {
            float l0 = __hip_atomic_load(&N->lo0[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float l1 = __hip_atomic_load(&N->lo1[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float h0 = __hip_atomic_load(&N->hi0[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float h1 = __hip_atomic_load(&N->hi1[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bl[k] = fminf(l0, l1);
            bh[k] = fmaxf(h0, h1);
        }
        cur = p;
        p = node_parent[p];
    }
}

// Runs widen_greedy_kernel level by level from the root until no node is marked any more.  state: n ints.
static hipError_t widen_greedy(const BvhNode* d_nodes, int n, WideNode* wide_tmp, int* state, int max_rounds, bool use_marks, hipStream_t stream) {
    int* marked = nullptr;
    hipError_t e = hipMalloc(&marked, sizeof(int));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(state, 0, sizeof(int) * n, stream);
    const int one = 1;
    if (e == hipSuccess) e = hipMemcpyAsync(state, &one, sizeof(int), hipMemcpyHostToDevice, stream);
    const dim3 block(256), grid((n + 255) / 256);
    for (int round = 0; e == hipSuccess; ++round) {
        int again = 0;
        e = hipMemsetAsync(marked, 0, sizeof(int), stream);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(widen_greedy_kernel, grid, block, 0, stream, d_nodes, n, wide_tmp, state, marked, use_marks ? 1 : 0);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&again, marked, sizeof(int), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess || !again) break;
        if (round > max_rounds) { e = hipErrorUnknown; break; }
    }
    (void)hipFree(marked);
    return e;
}

hipError_t lbvh_build(const float* d_lo, const float* d_hi, int n, const float scene_lo[3], const float scene_hi[3],
                      int leaf_max, WideNode* d_wide, int* d_prim_of_slot, hipStream_t stream) {
    if (n < 2 || n > BVH_MAX_PRIMS || leaf_max < 1 || leaf_max > 8) return hipErrorInvalidValue;
    hipError_t e;
    BvhNode* d_nodes = nullptr;
    WideNode* wide_tmp = nullptr;
    unsigned long long *keys_in = nullptr, *keys_out = nullptr;
    unsigned int* index_in = nullptr;
    int *node_parent = nullptr, *leaf_parent = nullptr, *range_first = nullptr, *range_count = nullptr;
    unsigned int* flags = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    auto cleanup = [&]() {
        (void)hipFree(d_nodes);
        (void)hipFree(wide_tmp);
        (void)hipFree(keys_in); (void)hipFree(keys_out); (void)hipFree(index_in); (void)hipFree(node_parent);
        (void)hipFree(leaf_parent); (void)hipFree(range_first); (void)hipFree(range_count); (void)hipFree(flags);
        (void)hipFree(temp);
    };
#define LBVH_TRY(x) do { e = (x); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    LBVH_TRY(hipMalloc(&d_nodes, sizeof(BvhNode) * (size_t)(n - 1)));
    LBVH_TRY(hipMalloc(&keys_in, sizeof(unsigned long long) * n));
    LBVH_TRY(hipMalloc(&keys_out, sizeof(unsigned long long) * n));
    LBVH_TRY(hipMalloc(&index_in, sizeof(unsigned int) * n));
    LBVH_TRY(hipMalloc(&node_parent, sizeof(int) * n));
    LBVH_TRY(hipMalloc(&leaf_parent, sizeof(int) * n));
    LBVH_TRY(hipMalloc(&range_first, sizeof(int) * n));
    LBVH_TRY(hipMalloc(&range_count, sizeof(int) * n));
    LBVH_TRY(hipMalloc(&flags, sizeof(unsigned int) * n));
    LBVH_TRY(hipMemsetAsync(flags, 0, sizeof(unsigned int) * n, stream));
    float3 slo = make_float3(scene_lo[0], scene_lo[1], scene_lo[2]);
    auto inv = [](float a, float b) { return b > a ? 1.0f / (b - a) : 0.0f; };
    float3 iext = make_float3(inv(scene_lo[0], scene_hi[0]), inv(scene_lo[1], scene_hi[1]), inv(scene_lo[2], scene_hi[2]));
    dim3 block(256), grid((n + 255) / 256);
    hipLaunchKernelGGL(morton_kernel, grid, block, 0, stream, d_lo, d_hi, n, slo, iext, keys_in, index_in);
    LBVH_TRY(hipGetLastError());
    unsigned int* index_out = reinterpret_cast<unsigned int*>(d_prim_of_slot);
    LBVH_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, index_in, index_out, (size_t)n, 0, 63,
                                       stream));
    LBVH_TRY(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    LBVH_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, index_in, index_out, (size_t)n, 0, 63,
                                       stream));
    hipLaunchKernelGGL(hierarchy_kernel, grid, block, 0, stream, keys_out, n, d_nodes, node_parent, leaf_parent,
                       range_first, range_count);
    LBVH_TRY(hipGetLastError());
    hipLaunchKernelGGL(refit_kernel, grid, block, 0, stream, d_lo, d_hi, d_prim_of_slot, n, d_nodes, node_parent,
                       leaf_parent, flags);
    LBVH_TRY(hipGetLastError());
    if (leaf_max > 1) {
        hipLaunchKernelGGL(collapse_kernel, grid, block, 0, stream, d_nodes, n, range_first, range_count, leaf_max);
        LBVH_TRY(hipGetLastError());
    }
    LBVH_TRY(hipMalloc(&wide_tmp, sizeof(WideNode) * (size_t)(n - 1)));
    // only some of the binary nodes become wide nodes that rays visit: pack those (more useful nodes per cache line)
    int* flag = reinterpret_cast<int*>(flags);   // the refit flags are free again
    int* new_index = leaf_parent;                // and so is this array
    static const bool greedy = [] { const char* e = getenv("RRT_DEBUG_WIDEN_GREEDY"); return e ? atoi(e) != 0 : true; }();  // experiments
    if (greedy) {
        int* state = range_first;  // free after the collapse
        LBVH_TRY(widen_greedy(d_nodes, n, wide_tmp, state, 4 * 96, false, stream));  // no radix tree over 63 + 32 bits is deeper
        hipLaunchKernelGGL(finished_flag_kernel, grid, block, 0, stream, state, n, flag);
        LBVH_TRY(hipGetLastError());
    } else {
        hipLaunchKernelGGL(widen_kernel, grid, block, 0, stream, d_nodes, n, wide_tmp);
        LBVH_TRY(hipGetLastError());
        hipLaunchKernelGGL(reachable_kernel, grid, block, 0, stream, node_parent, range_count, n, leaf_max, flag);
        LBVH_TRY(hipGetLastError());
    }
    {
        size_t scan_bytes = 0;
        LBVH_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flag, new_index, 0, (size_t)(n - 1), rocprim::plus<int>(), stream));
        if (scan_bytes > temp_bytes) {
            LBVH_TRY(hipStreamSynchronize(stream));
            (void)hipFree(temp);
            temp = nullptr;
            LBVH_TRY(hipMalloc(&temp, scan_bytes));
            temp_bytes = scan_bytes;
        }
        LBVH_TRY(rocprim::exclusive_scan(temp, scan_bytes, flag, new_index, 0, (size_t)(n - 1), rocprim::plus<int>(), stream));
    }
    hipLaunchKernelGGL(compact_kernel, grid, block, 0, stream, wide_tmp, flag, new_index, n, d_wide);
    LBVH_TRY(hipGetLastError());
    LBVH_TRY(hipStreamSynchronize(stream));
#undef LBVH_TRY
    cleanup();
    return hipSuccess;
}

// The wide nodes of a binary tree built on the host (rrt_sah.hpp): m binary nodes, node 0 the root.
hipError_t bvh_widen_host_tree(const BvhNode* h_nodes, int m, bool use_marks, WideNode* d_wide, hipStream_t stream) {
    if (m < 1) return hipErrorInvalidValue;
    const int n = m + 1;  // the kernels take the primitive count of a full binary tree
    hipError_t e;
    BvhNode* d_nodes = nullptr;
    WideNode* wide_tmp = nullptr;
    int *state = nullptr, *flag = nullptr, *new_index = nullptr;
    void* temp = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_nodes); (void)hipFree(wide_tmp); (void)hipFree(state); (void)hipFree(flag); (void)hipFree(new_index);
        (void)hipFree(temp);
    };
#define LBVH_TRY(x) do { e = (x); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    LBVH_TRY(hipMalloc(&d_nodes, sizeof(BvhNode) * (size_t)m));
    LBVH_TRY(hipMalloc(&wide_tmp, sizeof(WideNode) * (size_t)m));
    LBVH_TRY(hipMalloc(&state, sizeof(int) * n));
    LBVH_TRY(hipMalloc(&flag, sizeof(int) * n));
    LBVH_TRY(hipMalloc(&new_index, sizeof(int) * n));
    LBVH_TRY(hipMemcpyAsync(d_nodes, h_nodes, sizeof(BvhNode) * (size_t)m, hipMemcpyHostToDevice, stream));
    dim3 block(256), grid((n + 255) / 256);
    LBVH_TRY(widen_greedy(d_nodes, n, wide_tmp, state, m, use_marks, stream));
    hipLaunchKernelGGL(finished_flag_kernel, grid, block, 0, stream, state, n, flag);
    LBVH_TRY(hipGetLastError());
    size_t scan_bytes = 0;
    LBVH_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flag, new_index, 0, (size_t)m, rocprim::plus<int>(), stream));
    LBVH_TRY(hipMalloc(&temp, scan_bytes ? scan_bytes : 16));
    LBVH_TRY(rocprim::exclusive_scan(temp, scan_bytes, flag, new_index, 0, (size_t)m, rocprim::plus<int>(), stream));
    hipLaunchKernelGGL(compact_kernel, grid, block, 0, stream, wide_tmp, flag, new_index, n, d_wide);
    LBVH_TRY(hipGetLastError());
    LBVH_TRY(hipStreamSynchronize(stream));
#undef LBVH_TRY
    cleanup();
    return hipSuccess;
}

template <typename R>
static hipError_t pack_leaves_t(const PrimRec<R>* prims, const int* prim_of_slot, int n, LeafRec<R>* out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_leaves_kernel<R>, dim3((n + 255) / 256), dim3(256), 0, stream, prims, prim_of_slot, n, out);
    return hipGetLastError();
}
hipError_t lbvh_pack_leaves(const PrimRec<float>* prims, const int* prim_of_slot, int n, LeafRec<float>* out, hipStream_t s) {
    return pack_leaves_t<float>(prims, prim_of_slot, n, out, s);
}
hipError_t lbvh_pack_leaves(const PrimRec<double>* prims, const int* prim_of_slot, int n, LeafRec<double>* out, hipStream_t s) {
    return pack_leaves_t<double>(prims, prim_of_slot, n, out, s);
}

}  // namespace rrt
