// GPU construction of the binary tree by agglomerative clustering along the Morton order
// (Meister & Bittner, "Parallel Locally-Ordered Clustering for Bounding Volume Hierarchy Construction", 2018).
//
//   1. morton codes of the box centres, radix sort (as for the radix tree, rrt_lbvh.hip)
//   2. every primitive starts as a cluster; the clusters stay in Morton order
//   3. per round: every cluster looks at its PLOC_RADIUS neighbours on either side and picks the one whose union
//      with it has the smallest box area; two clusters that pick each other merge into a new binary node; the
//      survivors are compacted, order kept
//   4. the last merge makes the root
//   5. after every round the new nodes get their wide-node marks (sah_mark_node: which descendants the 4-wide
//      node rooted there absorbs), so the widening kernel has nothing to decide
// A large box (a wall of the room around a mesh) is nobody's cheapest partner until the clusters around it have
// grown to its size, so it ends near the root by itself -- what the radix tree, which only looks at box centres,
// cannot do.  Node numbers are handed out from n - 2 downwards in merge order (by a scan, so the tree does not
// depend on scheduling): the root is node 0 and children have larger numbers than their parents.
#include <hip/hip_runtime.h>

#include <cstring>  // rocPRIM uses ::memset in host code without including it

#include <rocprim/rocprim.hpp>

#include "rrt_lbvh.hpp"
#include "rrt_sah.hpp"

namespace rrt {

namespace {

constexpr int PLOC_RADIUS = 16;
constexpr int PLOC_BLOCK = 256;

struct Cluster {
    float lo[3];
    int32_t link;  // node number, or ~slot of a single primitive
    float hi[3];
    int32_t pad;
};

__device__ __forceinline__ float union_area(const Cluster& a, const Cluster& b) {
    const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
    const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
    const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    const float u = dx * dy + dy * dz + dz * dx;
#ifdef RRT_EXP_PLOC_RATIO  // experiment: union area squared over the sum of the two areas
    auto area = [](const Cluster& c) { const float x = c.hi[0] - c.lo[0], y = c.hi[1] - c.lo[1], z = c.hi[2] - c.lo[2]; return x * y + y * z + z * x; };
    return u * u / fmaxf(area(a) + area(b), 1e-30f);
#else
    return u;
#endif
}

__global__ void __launch_bounds__(PLOC_BLOCK)
ploc_init_kernel(const float* __restrict__ lo, const float* __restrict__ hi, const int* __restrict__ prim_of_slot, int n,
                 Cluster* __restrict__ out) {
    const int i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int p = prim_of_slot[i];
    Cluster c;
    for (int a = 0; a < 3; ++a) { c.lo[a] = lo[3 * p + a]; c.hi[a] = hi[3 * p + a]; }
    c.link = ~i;
    c.pad = 0;
    out[i] = c;
}

// nearest[i] = the cluster within PLOC_RADIUS positions of i whose union with i has the smallest area
// (ties: the lower position, so that the choice is the same from both sides)
__global__ void __launch_bounds__(PLOC_BLOCK)
ploc_nearest_kernel(const Cluster* __restrict__ c, int m, int* __restrict__ nearest) {
    __shared__ Cluster tile[PLOC_BLOCK + 2 * PLOC_RADIUS];
    const int base = blockIdx.x * PLOC_BLOCK - PLOC_RADIUS;
    for (int t = threadIdx.x; t < PLOC_BLOCK + 2 * PLOC_RADIUS; t += PLOC_BLOCK) {
        const int j = base + t;
        if (j >= 0 && j < m) tile[t] = c[j];
    }
    __syncthreads();
    const int i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
    if (i >= m) return;
    const Cluster me = tile[threadIdx.x + PLOC_RADIUS];
    float best = 3.0e38f;
    int best_j = -1;
    for (int d = -PLOC_RADIUS; d <= PLOC_RADIUS; ++d) {
        const int j = i + d;
        if (d == 0 || j < 0 || j >= m) continue;
        const float a = union_area(me, tile[threadIdx.x + PLOC_RADIUS + d]);
        if (a < best) { best = a; best_j = j; }
    }
    nearest[i] = best_j;
}

// A round that pairs neighbours by position alone (2k with 2k + 1): see ploc_build.
__global__ void __launch_bounds__(PLOC_BLOCK) ploc_pair_kernel(int m, int* __restrict__ nearest) {
    const int i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
    if (i >= m) return;
    nearest[i] = (i ^ 1) < m ? (i ^ 1) : -1;
}

// flags[i] = (cluster i survives the round) | (cluster i is the lower one of a merging pair) << 32
__global__ void __launch_bounds__(PLOC_BLOCK)
ploc_flag_kernel(const int* __restrict__ nearest, int m, unsigned long long* __restrict__ flags) {
    const int i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
    if (i >= m) return;
    const int j = nearest[i];
    const bool mutual = j >= 0 && nearest[j] == i;
    const unsigned long long keep = (mutual && j < i) ? 0ull : 1ull;
    const unsigned long long merge = (mutual && i < j) ? 1ull : 0ull;
    flags[i] = keep | (merge << 32);
}

// Writes the new binary nodes and the next round's clusters.  scan[i] = exclusive prefix sums of flags.
__global__ void __launch_bounds__(PLOC_BLOCK)
ploc_merge_kernel(const Cluster* __restrict__ c, const int* __restrict__ nearest, const unsigned long long* __restrict__ flags,
                  const unsigned long long* __restrict__ scan, int m, int first_node, BvhNode* __restrict__ nodes,
                  Cluster* __restrict__ out) {
    const int i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
    if (i >= m) return;
    const unsigned long long f = flags[i];
    if (!(f & 1ull)) return;  // absorbed by its partner
    Cluster me = c[i];
    if (f >> 32) {
        const Cluster other = c[nearest[i]];
        const int id = first_node - (int)(scan[i] >> 32);
        BvhNode N;
        for (int a = 0; a < 3; ++a) { N.lo0[a] = me.lo[a]; N.hi0[a] = me.hi[a]; N.lo1[a] = other.lo[a]; N.hi1[a] = other.hi[a]; }
        N.c0 = me.link;
        N.c1 = other.link;
        N.pad0 = N.pad1 = 0;
        nodes[id] = N;
        for (int a = 0; a < 3; ++a) { me.lo[a] = fminf(me.lo[a], other.lo[a]); me.hi[a] = fmaxf(me.hi[a], other.hi[a]); }
        me.link = id;
    }
    out[(int)(scan[i] & 0xffffffffull)] = me;
}

// Which descendants the wide node rooted at each of this round's new nodes absorbs (sah_mark_node): their children
// and grandchildren were made in earlier rounds.
__global__ void __launch_bounds__(PLOC_BLOCK) ploc_mark_kernel(BvhNode* nodes, float* f, int first, int count) {
    const int i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
    if (i < count) sah_mark_node(nodes, f, (size_t)(first + i));
}

}  // namespace

// d_nodes: n - 1 binary nodes (node 0 the root, one primitive per leaf: link ~slot; BvhNode::pad0 = the wide-node
// marks of sah_mark_node); d_prim_of_slot as
// produced by the Morton sort (lbvh_sort).  Needs n >= 2.  Synchronises the stream.
hipError_t ploc_build(const float* d_lo, const float* d_hi, const int* d_prim_of_slot, int n, BvhNode* d_nodes, hipStream_t stream) {
    if (n < 2) return hipErrorInvalidValue;
    hipError_t e;
    Cluster *a = nullptr, *b = nullptr;
    int* nearest = nullptr;
    float* f = nullptr;
    unsigned long long *flags = nullptr, *scan = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    auto cleanup = [&]() {
        (void)hipFree(a); (void)hipFree(b); (void)hipFree(nearest); (void)hipFree(f); (void)hipFree(flags); (void)hipFree(scan); (void)hipFree(temp);
    };
#define PLOC_TRY(x) do { e = (x); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    PLOC_TRY(hipMalloc(&a, sizeof(Cluster) * (size_t)n));
    PLOC_TRY(hipMalloc(&b, sizeof(Cluster) * (size_t)n));
    PLOC_TRY(hipMalloc(&nearest, sizeof(int) * (size_t)n));
    PLOC_TRY(hipMalloc(&f, sizeof(float) * (size_t)n));
    PLOC_TRY(hipMalloc(&flags, sizeof(unsigned long long) * ((size_t)n + 1)));
    PLOC_TRY(hipMalloc(&scan, sizeof(unsigned long long) * ((size_t)n + 1)));
    PLOC_TRY(rocprim::exclusive_scan(nullptr, temp_bytes, flags, scan, 0ull, (size_t)n + 1, rocprim::plus<unsigned long long>(), stream));
    PLOC_TRY(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    auto grid = [](int m) { return dim3((unsigned)((m + PLOC_BLOCK - 1) / PLOC_BLOCK)); };
    hipLaunchKernelGGL(ploc_init_kernel, grid(n), dim3(PLOC_BLOCK), 0, stream, d_lo, d_hi, d_prim_of_slot, n, a);
    PLOC_TRY(hipGetLastError());
    int m = n, first_node = n - 2;
    // Every round merges at least the globally cheapest pair, and on meshes a third of the clusters; an input on which
    // every cluster prefers the same side (boxes growing along the order) would merge one pair per round.  A round that
    // merges fewer than 1/32 of the clusters is therefore followed by one that pairs neighbours by position, which halves
    // the count whatever the boxes are: at most ~450 rounds for 2^28 primitives.
    bool pair_by_position = false;
    for (int round = 0; m > 1; ++round) {
        if (round > 4096) PLOC_TRY(hipErrorUnknown);
        if (pair_by_position) hipLaunchKernelGGL(ploc_pair_kernel, grid(m), dim3(PLOC_BLOCK), 0, stream, m, nearest);
        else hipLaunchKernelGGL(ploc_nearest_kernel, grid(m), dim3(PLOC_BLOCK), 0, stream, a, m, nearest);
        PLOC_TRY(hipGetLastError());
        hipLaunchKernelGGL(ploc_flag_kernel, grid(m), dim3(PLOC_BLOCK), 0, stream, nearest, m, flags);
        PLOC_TRY(hipGetLastError());
        PLOC_TRY(hipMemsetAsync(flags + m, 0, sizeof(unsigned long long), stream));  // scan[m] = the totals
        PLOC_TRY(rocprim::exclusive_scan(temp, temp_bytes, flags, scan, 0ull, (size_t)m + 1, rocprim::plus<unsigned long long>(), stream));
        hipLaunchKernelGGL(ploc_merge_kernel, grid(m), dim3(PLOC_BLOCK), 0, stream, a, nearest, flags, scan, m, first_node, d_nodes, b);
        PLOC_TRY(hipGetLastError());
        unsigned long long total = 0;
        PLOC_TRY(hipMemcpyAsync(&total, scan + m, sizeof(total), hipMemcpyDeviceToHost, stream));
        PLOC_TRY(hipStreamSynchronize(stream));
        const int kept = (int)(total & 0xffffffffull), merged = (int)(total >> 32);
        if (merged < 1 || kept != m - merged) PLOC_TRY(hipErrorUnknown);
        hipLaunchKernelGGL(ploc_mark_kernel, grid(merged), dim3(PLOC_BLOCK), 0, stream, d_nodes, f, first_node - merged + 1, merged);
        PLOC_TRY(hipGetLastError());
        pair_by_position = !pair_by_position && m > 64 && merged < m / 32;
        first_node -= merged;
        m = kept;
        Cluster* t = a; a = b; b = t;
    }
    if (first_node != -1) PLOC_TRY(hipErrorUnknown);
#undef PLOC_TRY
    cleanup();
    return hipSuccess;
}

}  // namespace rrt
