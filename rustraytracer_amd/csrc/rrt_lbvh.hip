// The tree on the GPU: Morton order, clustering, 4-wide nodes.
//
//   1. host: f64 bounds of every primitive, rounded outward to f32 and padded (rrt_api.cpp)
//   2. morton_kernel   63-bit Morton code of the box centre (21 bits per axis)
//   3. rocPRIM radix sort of (code, primitive) pairs
//   4. ploc_build      the binary tree by clustering along that order (rrt_ploc.hip), one primitive per leaf,
//                      with the marks that say which descendants every 4-wide node absorbs
//   5. widen_kernel    the 4-wide nodes rays traverse, boxes quantised to 8 bits; top-down, one launch per level
//   6. compact_kernel  packs them to the front of the node array
// Steps 5 and 6 also serve binary trees built on the host (bvh_widen_host_tree; rrt_sah.hpp).
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

// Wide node i = binary node i with the descendants its builder marked absorbed (BvhNode::pad0, SAH_OPEN_* bits;
// sah_mark_node in rrt_sah.hpp): up to four children, boxes quantised (see WideNode).  Which binary nodes become
// wide nodes is known only top-down: state[i] = 1 marks a node whose turn it is, 2 a finished one; every launch
// handles the marked nodes and marks their internal children, the host repeats it until a launch marks nothing.
__global__ void __launch_bounds__(256)
widen_kernel(const BvhNode* __restrict__ nodes, int n, WideNode* __restrict__ wide, int* __restrict__ state,
             int* __restrict__ marked, unsigned int* __restrict__ level_of, unsigned int level) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    if (__hip_atomic_load(&state[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1) return;
    float lo[4][3], hi[4][3];
    int child[4];
    int k = 2;
    auto put = [&](int at, const float* l, const float* h, int c) {
        for (int a = 0; a < 3; ++a) { lo[at][a] = l[a]; hi[at][a] = h[a]; }
        child[at] = c;
    };
    const BvhNode N = nodes[i];
    put(0, N.lo0, N.hi0, N.c0);
    put(1, N.lo1, N.hi1, N.c1);
    auto open = [&](int at) {  // the child in slot `at` gives way to its two children; returns the slot of the second one
        const BvhNode C = nodes[child[at]];
        put(at, C.lo0, C.hi0, C.c0);
        put(k, C.lo1, C.hi1, C.c1);
        return k++;
    };
    const int32_t shape = N.pad0;
    int left1 = -1, right1 = -1;
    if ((shape & SAH_OPEN_LEFT) && child[0] >= 0) left1 = open(0);
    if ((shape & SAH_OPEN_RIGHT) && child[1] >= 0 && k < 4) right1 = open(1);
    if ((shape & SAH_OPEN_LEFT_0) && left1 >= 0 && child[0] >= 0 && k < 4) open(0);
    if ((shape & SAH_OPEN_LEFT_1) && left1 >= 0 && child[left1] >= 0 && k < 4) open(left1);
    if ((shape & SAH_OPEN_RIGHT_0) && right1 >= 0 && child[1] >= 0 && k < 4) open(1);
    if ((shape & SAH_OPEN_RIGHT_1) && right1 >= 0 && child[right1] >= 0 && k < 4) open(right1);
    write_wide_node(lo, hi, child, k, &wide[i]);
    level_of[i] = level;
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

// Level-major packing order: key = the node's level in the wide tree (unreachable binary nodes sort behind everything).
__global__ void __launch_bounds__(256)
level_key_kernel(const int* __restrict__ state, unsigned int* __restrict__ level_of, unsigned int* __restrict__ index, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    level_of[i] = state[i] == 2 ? (level_of[i] < 0xfffeu ? level_of[i] : 0xfffeu) : 0xffffu;
    index[i] = (unsigned int)i;
}
__global__ void __launch_bounds__(256)
rank_to_index_kernel(const unsigned int* __restrict__ order, int n, int* __restrict__ new_index) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n - 1) return;
    new_index[order[j]] = j;  // (unreachable nodes get ranks behind the reachable ones; nobody links to them)
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

// d_wide = the wide nodes of the m binary nodes at d_nodes (marked, node 0 the root), packed; *depth = the levels of
// wide nodes (what bounds a traversal stack: a walk holds at most three entries per level).  Synchronises the stream.
// level_major = false: packed in the order of the binary nodes' numbers (the builders that number the top of the tree
// first); true: level by level from the root, by number inside a level (for a builder that numbers depth-first: the
// traversal kernels keep the first wide nodes of the array in LDS, and those should be the top of the tree).
static hipError_t widen_and_pack(const BvhNode* d_nodes, int m, WideNode* d_wide, hipStream_t stream, int* depth, bool level_major) {
    const int n = m + 1;  // the kernels take the primitive count of a full binary tree
    hipError_t e;
    WideNode* wide_tmp = nullptr;
    int *state = nullptr, *flag = nullptr, *new_index = nullptr, *marked = nullptr;
    unsigned int *level_of = nullptr, *level_sorted = nullptr, *index_in = nullptr, *index_sorted = nullptr;
    void* temp = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(wide_tmp); (void)hipFree(state); (void)hipFree(flag); (void)hipFree(new_index); (void)hipFree(marked);
        (void)hipFree(level_of); (void)hipFree(level_sorted); (void)hipFree(index_in); (void)hipFree(index_sorted);
        (void)hipFree(temp);
    };
#define LBVH_TRY(x) do { e = (x); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    LBVH_TRY(hipMalloc(&wide_tmp, sizeof(WideNode) * (size_t)m));
    LBVH_TRY(hipMalloc(&state, sizeof(int) * n));
    LBVH_TRY(hipMalloc(&flag, sizeof(int) * n));
    LBVH_TRY(hipMalloc(&new_index, sizeof(int) * n));
    LBVH_TRY(hipMalloc(&marked, sizeof(int)));
    LBVH_TRY(hipMalloc(&level_of, sizeof(unsigned int) * n));
    LBVH_TRY(hipMemsetAsync(state, 0, sizeof(int) * n, stream));
    const int one = 1;
    LBVH_TRY(hipMemcpyAsync(state, &one, sizeof(int), hipMemcpyHostToDevice, stream));
    const dim3 block(256), grid((n + 255) / 256);
    for (int round = 0;; ++round) {  // one launch per level of the wide tree
        int again = 0;
        LBVH_TRY(hipMemsetAsync(marked, 0, sizeof(int), stream));
        hipLaunchKernelGGL(widen_kernel, grid, block, 0, stream, d_nodes, n, wide_tmp, state, marked, level_of, (unsigned int)round);
        LBVH_TRY(hipGetLastError());
        LBVH_TRY(hipMemcpyAsync(&again, marked, sizeof(int), hipMemcpyDeviceToHost, stream));
        LBVH_TRY(hipStreamSynchronize(stream));
        *depth = round + 1;
        if (!again) break;
        if (round > m) LBVH_TRY(hipErrorUnknown);
    }
    // only some of the binary nodes are the root of a wide node: pack those (more useful nodes per cache line)
    hipLaunchKernelGGL(finished_flag_kernel, grid, block, 0, stream, state, n, flag);
    LBVH_TRY(hipGetLastError());
    size_t scan_bytes = 0;
    if (level_major) {
        LBVH_TRY(hipMalloc(&level_sorted, sizeof(unsigned int) * n));
        LBVH_TRY(hipMalloc(&index_in, sizeof(unsigned int) * n));
        LBVH_TRY(hipMalloc(&index_sorted, sizeof(unsigned int) * n));
        hipLaunchKernelGGL(level_key_kernel, grid, block, 0, stream, state, level_of, index_in, n);
        LBVH_TRY(hipGetLastError());
        LBVH_TRY(rocprim::radix_sort_pairs(nullptr, scan_bytes, level_of, level_sorted, index_in, index_sorted, (size_t)m, 0, 16, stream));
        LBVH_TRY(hipMalloc(&temp, scan_bytes ? scan_bytes : 16));
        LBVH_TRY(rocprim::radix_sort_pairs(temp, scan_bytes, level_of, level_sorted, index_in, index_sorted, (size_t)m, 0, 16, stream));
        hipLaunchKernelGGL(rank_to_index_kernel, grid, block, 0, stream, index_sorted, n, new_index);
        LBVH_TRY(hipGetLastError());
    } else {
        LBVH_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flag, new_index, 0, (size_t)m, rocprim::plus<int>(), stream));
        LBVH_TRY(hipMalloc(&temp, scan_bytes ? scan_bytes : 16));
        LBVH_TRY(rocprim::exclusive_scan(temp, scan_bytes, flag, new_index, 0, (size_t)m, rocprim::plus<int>(), stream));
    }
    hipLaunchKernelGGL(compact_kernel, grid, block, 0, stream, wide_tmp, flag, new_index, n, d_wide);
    LBVH_TRY(hipGetLastError());
    LBVH_TRY(hipStreamSynchronize(stream));
#undef LBVH_TRY
    cleanup();
    return hipSuccess;
}

hipError_t lbvh_build(const float* d_lo, const float* d_hi, int n, const float scene_lo[3], const float scene_hi[3],
                      WideNode* d_wide, int* d_prim_of_slot, hipStream_t stream, int* depth) {
    if (n < 2 || n > BVH_MAX_PRIMS) return hipErrorInvalidValue;
    hipError_t e;
    BvhNode* d_nodes = nullptr;
    unsigned long long *keys_in = nullptr, *keys_out = nullptr;
    unsigned int* index_in = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    auto cleanup = [&]() {
        (void)hipFree(d_nodes); (void)hipFree(keys_in); (void)hipFree(keys_out); (void)hipFree(index_in); (void)hipFree(temp);
    };
#define LBVH_TRY(x) do { e = (x); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    LBVH_TRY(hipMalloc(&d_nodes, sizeof(BvhNode) * (size_t)(n - 1)));
    LBVH_TRY(hipMalloc(&keys_in, sizeof(unsigned long long) * n));
    LBVH_TRY(hipMalloc(&keys_out, sizeof(unsigned long long) * n));
    LBVH_TRY(hipMalloc(&index_in, sizeof(unsigned int) * n));
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
    LBVH_TRY(ploc_build(d_lo, d_hi, d_prim_of_slot, n, d_nodes, stream));
    LBVH_TRY(widen_and_pack(d_nodes, n - 1, d_wide, stream, depth, false));
#undef LBVH_TRY
    cleanup();
    return hipSuccess;
}

// The wide nodes of a binary tree built and marked on the host (rrt_sah.hpp): m binary nodes, node 0 the root.
hipError_t bvh_widen_host_tree(const BvhNode* h_nodes, int m, WideNode* d_wide, hipStream_t stream, int* depth) {
    if (m < 1) return hipErrorInvalidValue;
    BvhNode* d_nodes = nullptr;
    hipError_t e = hipMalloc(&d_nodes, sizeof(BvhNode) * (size_t)m);
    if (e == hipSuccess) e = hipMemcpyAsync(d_nodes, h_nodes, sizeof(BvhNode) * (size_t)m, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = widen_and_pack(d_nodes, m, d_wide, stream, depth, false);
    (void)hipFree(d_nodes);
    return e;
}

// The whole surface-area tree on the GPU (rrt_gsah.hip): binary tree, wide nodes, packed level by level.
hipError_t gsah_tree(const float* d_lo, const float* d_hi, const float* d_size, int n, WideNode* d_wide, int* d_prim_of_slot,
                     hipStream_t stream, int* depth) {
    if (n < 2 || n > BVH_MAX_PRIMS) return hipErrorInvalidValue;
    BvhNode* d_nodes = nullptr;
    hipError_t e = hipMalloc(&d_nodes, sizeof(BvhNode) * (size_t)(n - 1));
    int levels = 0;
    if (e == hipSuccess) e = gsah_build(d_lo, d_hi, d_size, n, d_nodes, d_prim_of_slot, stream, &levels);
    if (e == hipSuccess) e = widen_and_pack(d_nodes, n - 1, d_wide, stream, depth, true);
    (void)hipFree(d_nodes);
    return e;
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
