// Measures sustained VALU issue rate on gfx950 for a few instruction kinds:
// cycles per wave64 instruction per SIMD, as a function of waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s0, float s1) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (KIND == 0) {  // 8 independent v_fma_f32 (VGPR operands)
                a0 = __builtin_fmaf(a0, a1, a2); a1 = __builtin_fmaf(a1, a2, a3); a2 = __builtin_fmaf(a2, a3, a4); a3 = __builtin_fmaf(a3, a4, a5);
                a4 = __builtin_fmaf(a4, a5, a6); a5 = __builtin_fmaf(a5, a6, a7); a6 = __builtin_fmaf(a6, a7, a0); a7 = __builtin_fmaf(a7, a0, a1);
            } else if (KIND == 1) {  // fma with one SGPR operand
                a0 = __builtin_fmaf(a0, s0, a1); a1 = __builtin_fmaf(a1, s1, a2); a2 = __builtin_fmaf(a2, s0, a3); a3 = __builtin_fmaf(a3, s1, a4);
                a4 = __builtin_fmaf(a4, s0, a5); a5 = __builtin_fmaf(a5, s1, a6); a6 = __builtin_fmaf(a6, s0, a7); a7 = __builtin_fmaf(a7, s1, a0);
            } else if (KIND == 2) {  // v_mul_lo_u32
                u0 *= 0x9E3779B1u; u1 *= 0x85EBCA6Bu; u2 *= 0xC2B2AE35u; u3 *= 0x27D4EB2Fu;
                u0 ^= u1; u1 ^= u2; u2 ^= u3; u3 ^= u0;
            } else if (KIND == 3) {  // v_rcp_f32
                a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
                a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
            } else if (KIND == 4) {  // dependent chain of fma
                a0 = __builtin_fmaf(a0, a0, a1); a0 = __builtin_fmaf(a0, a0, a1); a0 = __builtin_fmaf(a0, a0, a1); a0 = __builtin_fmaf(a0, a0, a1);
                a0 = __builtin_fmaf(a0, a0, a1); a0 = __builtin_fmaf(a0, a0, a1); a0 = __builtin_fmaf(a0, a0, a1); a0 = __builtin_fmaf(a0, a0, a1);
            } else if (KIND == 5) {  // xor/shift integer ops
                u0 ^= u0 >> 7; u1 ^= u1 << 9; u2 ^= u2 >> 5; u3 ^= u3 << 3;
                u0 += u1; u1 += u2; u2 += u3; u3 += u0;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3);
}

template <int KIND>
int run(const char* name, int insts_per_iter) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int blocks_per_cu : {1, 2, 4, 8}) {
        int grid = cus * blocks_per_cu;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0001f, 0.9999f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.9999f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double wave_insts_per_simd = (double)iters * 8 * insts_per_iter * blocks_per_cu;  // one wave of each block per SIMD
        double cycles = ms * 1e-3 * 2.4e9;
        printf("%-28s waves/SIMD %d: %.3f ms, %.2f cycles(@2.4GHz) per wave-instruction per SIMD\n", name, blocks_per_cu, ms,
               cycles / wave_insts_per_simd);
    }
    CHECK(hipFree(out));
    return 0;
}

int main() {
    run<0>("v_fma_f32 vgpr", 8);
    run<1>("v_fma_f32 sgpr operand", 8);
    run<2>("v_mul_lo_u32 (+xor)", 8);
    run<3>("v_rcp_f32", 8);
    run<4>("v_fma_f32 dependent", 8);
    run<5>("shift/xor/add int", 12);
    return 0;
}
