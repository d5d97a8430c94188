// Divergent dependent gathers: every lane walks its own chain of random records in a table.
// Measures lane-steps/s for record sizes 16..128 B and table sizes, at full occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VEC>  // record = VEC x 16 bytes
__global__ void __launch_bounds__(256) chase(const uint4* __restrict__ table, uint32_t n_rec, int steps, uint32_t* out) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_rec;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        const uint4* r = table + (size_t)idx * VEC;
        uint4 v[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = r[k];
        uint32_t h = 0;
#pragma unroll
        for (int k = 0; k < VEC; ++k) h += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        acc += h;
        idx = (h * 2654435761u + (uint32_t)s) % n_rec;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

// The same chase with the four lanes of a quad fetching each record together: load j brings, in lane L, the 16-byte
// piece (L & 3) of the record that quad lane j wants, so the four lanes of a quad read ONE 64-byte record per load
// instead of four different ones (does the address unit charge per lane or per line?).  The hash of the benchmark
// only needs sums over a record, so the pieces are folded where they land and the sums exchanged inside the quad.
template <int DPP_CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_CTRL, 0xf, 0xf, false);
}
__global__ void __launch_bounds__(256) chase_quad(const uint4* __restrict__ table, uint32_t n_rec, int steps, uint32_t* out) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_rec;
    const uint32_t piece = threadIdx.x & 3u;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        const uint32_t i0 = quad_perm<0x00>(idx), i1 = quad_perm<0x55>(idx), i2 = quad_perm<0xaa>(idx), i3 = quad_perm<0xff>(idx);
        const uint4 v0 = table[(size_t)i0 * 4 + piece], v1 = table[(size_t)i1 * 4 + piece];
        const uint4 v2 = table[(size_t)i2 * 4 + piece], v3 = table[(size_t)i3 * 4 + piece];
        uint32_t p0 = v0.x ^ v0.y ^ v0.z ^ v0.w, p1 = v1.x ^ v1.y ^ v1.z ^ v1.w;
        uint32_t p2 = v2.x ^ v2.y ^ v2.z ^ v2.w, p3 = v3.x ^ v3.y ^ v3.z ^ v3.w;
        // sum over the quad of p_j = the hash of quad lane j's record
        p0 += quad_perm<0xb1>(p0); p1 += quad_perm<0xb1>(p1); p2 += quad_perm<0xb1>(p2); p3 += quad_perm<0xb1>(p3);
        p0 += quad_perm<0x4e>(p0); p1 += quad_perm<0x4e>(p1); p2 += quad_perm<0x4e>(p2); p3 += quad_perm<0x4e>(p3);
        const uint32_t h = piece == 0 ? p0 : piece == 1 ? p1 : piece == 2 ? p2 : p3;
        acc += h;
        idx = (h * 2654435761u + (uint32_t)s) % n_rec;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

void run_quad(const uint4* d_table, size_t table_bytes, uint32_t* d_out, int blocks, int steps) {
    uint32_t n_rec = (uint32_t)(table_bytes / 64);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(chase_quad, dim3(blocks), dim3(256), 0, 0, d_table, n_rec, 8, d_out);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(chase_quad, dim3(blocks), dim3(256), 0, 0, d_table, n_rec, steps, d_out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double lane_steps = (double)blocks * 256 * steps;
    printf("record  64 B  table %5zu MB  %7.2f G lane-steps/s  %7.1f GB/s  (%.2f ms)  quad-cooperative loads\n", table_bytes >> 20,
           lane_steps / ms / 1e6, lane_steps * 64 / ms / 1e6, ms);
}

template <int VEC>
void run(const uint4* d_table, size_t table_bytes, uint32_t* d_out, int blocks, int steps) {
    uint32_t n_rec = (uint32_t)(table_bytes / (16 * VEC));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(chase<VEC>, dim3(blocks), dim3(256), 0, 0, d_table, n_rec, 8, d_out);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(chase<VEC>, dim3(blocks), dim3(256), 0, 0, d_table, n_rec, steps, d_out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double lane_steps = (double)blocks * 256 * steps;
    printf("record %3d B  table %5zu MB  %7.2f G lane-steps/s  %7.1f GB/s  (%.2f ms)\n", 16 * VEC, table_bytes >> 20,
           lane_steps / ms / 1e6, lane_steps * 16 * VEC / ms / 1e6, ms);
}

int main(int argc, char** argv) {
    int blocks = 256 * 8, steps = 400;
    size_t max_bytes = (size_t)1 << 30;
    std::vector<uint32_t> h(max_bytes / 4);
    uint32_t x = 12345;
    for (auto& w : h) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; w = x; }
    uint4* d_table; uint32_t* d_out;
    CHECK(hipMalloc(&d_table, max_bytes));
    CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(d_table, h.data(), max_bytes, hipMemcpyHostToDevice));
    for (size_t mb : {2, 16, 128, 1024}) {
        size_t bytes = mb << 20;
        run<1>(d_table, bytes, d_out, blocks, steps);
        run<2>(d_table, bytes, d_out, blocks, steps);
        run<4>(d_table, bytes, d_out, blocks, steps);
        run_quad(d_table, bytes, d_out, blocks, steps);
        run<8>(d_table, bytes, d_out, blocks, steps);
    }
    return 0;
}
