// GPU construction of a linear BVH (LBVH) over the bounded primitives of a scene.
//
//   1. host: f64 bounds of every primitive, rounded outward to f32 and padded (rrt_api.cpp)
//   2. morton_kernel   30-bit Morton code of the box centre | sorted position tie-break -> unique 64-bit keys
//   3. rocPRIM radix sort of the keys (the primitive slot rides in the low 32 bits)
//   4. hierarchy_kernel Karras' parallel binary radix tree: one thread per internal node
//   5. refit_kernel    bottom-up AABB fit, one thread per leaf, second arrival at a node continues
//
// Node layout: an internal node holds the boxes of BOTH children, so one 64-byte fetch decides both
// descents.  child < 0 encodes a leaf: ~child = sorted slot; prim_of_slot[slot] = scan-order primitive.
#include <hip/hip_runtime.h>

#include <cstring>  // rocPRIM uses ::memset in host code without including it

#include <rocprim/rocprim.hpp>

#include "rrt_lbvh.hpp"

namespace rrt {

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void __launch_bounds__(256)
morton_kernel(const float* __restrict__ lo, const float* __restrict__ hi, int n, float3 scene_lo, float3 inv_extent,
              unsigned long long* __restrict__ keys) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float cx = 0.5f * (lo[3 * i] + hi[3 * i]), cy = 0.5f * (lo[3 * i + 1] + hi[3 * i + 1]),
          cz = 0.5f * (lo[3 * i + 2] + hi[3 * i + 2]);
    float nx = fminf(fmaxf((cx - scene_lo.x) * inv_extent.x * 1024.f, 0.f), 1023.f);
    float ny = fminf(fmaxf((cy - scene_lo.y) * inv_extent.y * 1024.f, 0.f), 1023.f);
    float nz = fminf(fmaxf((cz - scene_lo.z) * inv_extent.z * 1024.f, 0.f), 1023.f);
    uint32_t code = (expand_bits10((uint32_t)nx) << 2) | (expand_bits10((uint32_t)ny) << 1) | expand_bits10((uint32_t)nz);
    keys[i] = ((unsigned long long)code << 32) | (uint32_t)i;  // unique: equal codes are split by the index
}

__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (2012), fig. 4.
__global__ void __launch_bounds__(256)
hierarchy_kernel(const unsigned long long* __restrict__ keys, int n, BvhNode* __restrict__ nodes,
                 int* __restrict__ node_parent, int* __restrict__ leaf_parent, int* __restrict__ prim_of_slot) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) prim_of_slot[i] = (int)(uint32_t)keys[i];
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
    if (left < 0) leaf_parent[~left] = i; else node_parent[left] = i;
    if (right < 0) leaf_parent[~right] = i; else node_parent[right] = i;
    if (i == 0) node_parent[0] = -1;
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

hipError_t lbvh_build(const float* d_lo, const float* d_hi, int n, const float scene_lo[3], const float scene_hi[3],
                      BvhNode* d_nodes, int* d_prim_of_slot, hipStream_t stream) {
    if (n < 2) return hipErrorInvalidValue;
    hipError_t e;
    unsigned long long *keys_in = nullptr, *keys_out = nullptr;
    int *node_parent = nullptr, *leaf_parent = nullptr;
    unsigned int* flags = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    auto cleanup = [&]() {
        (void)hipFree(keys_in); (void)hipFree(keys_out); (void)hipFree(node_parent); (void)hipFree(leaf_parent);
        (void)hipFree(flags); (void)hipFree(temp);
    };
#define LBVH_TRY(x) do { e = (x); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    LBVH_TRY(hipMalloc(&keys_in, sizeof(unsigned long long) * n));
    LBVH_TRY(hipMalloc(&keys_out, sizeof(unsigned long long) * n));
    LBVH_TRY(hipMalloc(&node_parent, sizeof(int) * n));
    LBVH_TRY(hipMalloc(&leaf_parent, sizeof(int) * n));
    LBVH_TRY(hipMalloc(&flags, sizeof(unsigned int) * n));
    LBVH_TRY(hipMemsetAsync(flags, 0, sizeof(unsigned int) * n, stream));
    float3 slo = make_float3(scene_lo[0], scene_lo[1], scene_lo[2]);
    auto inv = [](float a, float b) { return b > a ? 1.0f / (b - a) : 0.0f; };
    float3 iext = make_float3(inv(scene_lo[0], scene_hi[0]), inv(scene_lo[1], scene_hi[1]), inv(scene_lo[2], scene_hi[2]));
    dim3 block(256), grid((n + 255) / 256);
    hipLaunchKernelGGL(morton_kernel, grid, block, 0, stream, d_lo, d_hi, n, slo, iext, keys_in);
    LBVH_TRY(hipGetLastError());
    LBVH_TRY(rocprim::radix_sort_keys(nullptr, temp_bytes, keys_in, keys_out, (size_t)n, 0, 64, stream));
    LBVH_TRY(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    LBVH_TRY(rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, (size_t)n, 0, 64, stream));
    hipLaunchKernelGGL(hierarchy_kernel, grid, block, 0, stream, keys_out, n, d_nodes, node_parent, leaf_parent,
                       d_prim_of_slot);
    LBVH_TRY(hipGetLastError());
    hipLaunchKernelGGL(refit_kernel, grid, block, 0, stream, d_lo, d_hi, d_prim_of_slot, n, d_nodes, node_parent,
                       leaf_parent, flags);
    LBVH_TRY(hipGetLastError());
    LBVH_TRY(hipStreamSynchronize(stream));
#undef LBVH_TRY
    cleanup();
    return hipSuccess;
}

}  // namespace rrt
