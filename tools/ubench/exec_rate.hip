// How long a wave64 VALU instruction occupies its SIMD as a function of the EXEC mask (gfx950):
// all 64 lanes, the low 32, the low 16, one lane in four (every quarter of the wave still has live lanes).
// Measured on MI355X: 5.8 / 5.6 / 5.6 / 5.6 cycles per v_fma_f32 with three VGPR sources -- no pass of a partly
// masked wave is skipped; lanes switched off by divergence cost what live lanes cost.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k(float* out, int iters, int mode) {
    const unsigned lane = threadIdx.x & 63u;
    const bool live = mode == 0 ? true : mode == 1 ? lane < 32u : mode == 2 ? lane < 16u : (lane & 3u) == 0u;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    if (live) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                a0 = __builtin_fmaf(a0, a1, a2); a1 = __builtin_fmaf(a1, a2, a3); a2 = __builtin_fmaf(a2, a3, a4); a3 = __builtin_fmaf(a3, a4, a5);
                a4 = __builtin_fmaf(a4, a5, a6); a5 = __builtin_fmaf(a5, a6, a7); a6 = __builtin_fmaf(a6, a7, a0); a7 = __builtin_fmaf(a7, a0, a1);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks_per_cu = 8, iters = 20000;
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * blocks_per_cu));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[4] = {"all 64 lanes", "lanes 0-31", "lanes 0-15", "one lane in four"};
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, 100, mode);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, iters, mode);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double insts = (double)iters * 64 * blocks_per_cu;  // wave instructions per SIMD
        printf("%-18s %.3f ms, %.2f cycles(@2.4GHz) per wave64 v_fma_f32 per SIMD (8 waves per SIMD)\n", names[mode], ms,
               ms * 1e-3 * 2.4e9 / insts);
    }
    CHECK(hipFree(out));
    return 0;
}
