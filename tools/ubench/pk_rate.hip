// Issue rate of packed FP32 instructions on gfx950 (cycles per wave64 instruction per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, f2 s0, f2 s1) {
    f2 a0 = {threadIdx.x * 1e-3f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (KIND == 0) {  // v_pk_fma_f32, VGPR operands
                a0 = __builtin_elementwise_fma(a0, a1, a2); a1 = __builtin_elementwise_fma(a1, a2, a3);
                a2 = __builtin_elementwise_fma(a2, a3, a4); a3 = __builtin_elementwise_fma(a3, a4, a5);
                a4 = __builtin_elementwise_fma(a4, a5, a6); a5 = __builtin_elementwise_fma(a5, a6, a7);
                a6 = __builtin_elementwise_fma(a6, a7, a0); a7 = __builtin_elementwise_fma(a7, a0, a1);
            } else if (KIND == 1) {  // v_pk_fma_f32 with an SGPR-pair operand
                a0 = __builtin_elementwise_fma(a0, s0, a1); a1 = __builtin_elementwise_fma(a1, s1, a2);
                a2 = __builtin_elementwise_fma(a2, s0, a3); a3 = __builtin_elementwise_fma(a3, s1, a4);
                a4 = __builtin_elementwise_fma(a4, s0, a5); a5 = __builtin_elementwise_fma(a5, s1, a6);
                a6 = __builtin_elementwise_fma(a6, s0, a7); a7 = __builtin_elementwise_fma(a7, s1, a0);
            } else if (KIND == 2) {  // v_pk_mul_f32
                a0 = a0 * a1; a1 = a1 * a2; a2 = a2 * a3; a3 = a3 * a4; a4 = a4 * a5; a5 = a5 * a6; a6 = a6 * a7; a7 = a7 * a0;
            } else if (KIND == 3) {  // v_pk_add_f32
                a0 = a0 + a1; a1 = a1 + a2; a2 = a2 + a3; a3 = a3 + a4; a4 = a4 + a5; a5 = a5 + a6; a6 = a6 + a7; a7 = a7 + a0;
            } else if (KIND == 4) {  // scalar v_fma_f32 on the halves (same flops as KIND 0, twice the instructions)
                a0.x = __builtin_fmaf(a0.x, a1.x, a2.x); a0.y = __builtin_fmaf(a0.y, a1.y, a2.y);
                a1.x = __builtin_fmaf(a1.x, a2.x, a3.x); a1.y = __builtin_fmaf(a1.y, a2.y, a3.y);
                a2.x = __builtin_fmaf(a2.x, a3.x, a4.x); a2.y = __builtin_fmaf(a2.y, a3.y, a4.y);
                a3.x = __builtin_fmaf(a3.x, a4.x, a0.x); a3.y = __builtin_fmaf(a3.y, a4.y, a0.y);
            } else if (KIND == 5) {  // v_min3_f32
                a0.x = fminf(fminf(a0.x, a1.x), a2.x); a1.x = fminf(fminf(a1.x, a2.x), a3.x); a2.x = fminf(fminf(a2.x, a3.x), a4.x); a3.x = fminf(fminf(a3.x, a4.x), a5.x);
                a4.x = fminf(fminf(a4.x, a5.x), a6.x); a5.x = fminf(fminf(a5.x, a6.x), a7.x); a6.x = fminf(fminf(a6.x, a7.x), a0.x); a7.x = fminf(fminf(a7.x, a0.x), a1.x);
            }
        }
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <int KIND>
int run(const char* name) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    f2 s0 = {1.0001f, 0.9999f}, s1 = {0.9998f, 1.0002f};
    for (int blocks_per_cu : {1, 2, 4, 8}) {
        int grid = cus * blocks_per_cu;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, 100, s0, s1);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, s0, s1);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double n = (double)iters * 8 * 8 * blocks_per_cu;
        printf("%-34s waves/SIMD %d: %.3f ms, %.2f cycles(@2.4GHz) per source-level op per SIMD\n", name, blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / n);
    }
    CHECK(hipFree(out));
    return 0;
}
int main() {
    run<0>("v_pk_fma_f32 vgpr");
    run<1>("v_pk_fma_f32 sgpr-pair operand");
    run<2>("v_pk_mul_f32");
    run<3>("v_pk_add_f32");
    run<4>("2 x v_fma_f32 (same flops)");
    run<5>("v_min3_f32");
    return 0;
}
