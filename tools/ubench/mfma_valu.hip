// Can the matrix pipe take dot products off the VALU of a ray-per-lane kernel?
//
// v_mfma_f32_4x4x1_16B_f32 does, per lane, acc[i] += A[row i] * b for four rows i, where b is the lane's
// own value (its ray component) and A[row i] is the value lane (4 * block + i) holds: sixteen independent
// 4x4 outer products, lane L = 4 * block + column.  With the same four coefficients in every block (lane L
// holds coefficient L % 4) this is "four rows of a table times this lane's number" = the inner step of a
// wave-uniform primitive scan.
//
//  1. layout and rounding: a chain of four such instructions against a chain of fmaf, bit for bit, on random
//     and on tiny (subnormal) operands;
//  2. issue rates at 1..8 waves per SIMD: packed FMAs alone, MFMAs alone, and both interleaved in one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(64) exact_kernel(const float* coef /*[4 rows][4 k]*/, const float* x /*[64][4]*/, float* out_mfma, float* out_fma) {
    const int lane = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(coef[(lane & 3) * 4 + k], x[lane * 4 + k], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) {
        out_mfma[lane * 4 + i] = acc[i];
        float r = 0.f;
        for (int k = 0; k < 4; ++k) r = __builtin_fmaf(coef[i * 4 + k], x[lane * 4 + k], r);
        out_fma[lane * 4 + i] = r;
    }
}

// P packed FMAs and M matrix instructions per inner step, all independent
template <int P, int M>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters, f2 s0) {
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{threadIdx.x * 1e-3f + i, 1.f};
    f4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = f4{(float)i, 0.f, 1.f, 2.f};
    float ma = threadIdx.x * 1e-4f, mb = 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < P) a[j] = __builtin_elementwise_fma(a[j], s0, a[(j + 1) & 7]);
                if ((j & 1) == 0 && (j >> 1) < M) c[j >> 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ma, mb, c[j >> 1], 0, 0, 0);
            }
        }
    }
    f2 s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
    f4 t = c[0] + c[1] + c[2] + c[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + t[0] + t[1] + t[2] + t[3];
}

template <int P, int M>
int run(const char* name) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 10000;
    f2 s0 = {1.0001f, 0.9999f};
    for (int blocks_per_cu : {1, 2, 4, 8}) {
        int grid = cus * blocks_per_cu;
        hipLaunchKernelGGL((rate_kernel<P, M>), dim3(grid), dim3(256), 0, 0, out, 100, s0);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((rate_kernel<P, M>), dim3(grid), dim3(256), 0, 0, out, iters, s0);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double steps = (double)iters * 4 * blocks_per_cu;  // inner steps per SIMD (one wave of each block per SIMD)
        double cycles = ms * 1e-3 * 2.4e9 / steps;
        printf("%-34s waves/SIMD %d: %8.3f ms, %6.2f cycles(@2.4GHz) per step of %d v_pk_fma_f32 + %d v_mfma_f32_4x4x1\n", name, blocks_per_cu,
               ms, cycles, P, M);
    }
    CHECK(hipFree(out));
    return 0;
}

int main() {
    // ---- 1. layout and rounding
    std::vector<float> coef(16), x(256), om(256), of(256);
    float *d_coef, *d_x, *d_om, *d_of;
    CHECK(hipMalloc(&d_coef, 64)); CHECK(hipMalloc(&d_x, 1024)); CHECK(hipMalloc(&d_om, 1024)); CHECK(hipMalloc(&d_of, 1024));
    srand(7);
    for (int round = 0; round < 3; ++round) {
        const float scale = round == 0 ? 1.f : (round == 1 ? 1e-20f : 1e3f);
        for (auto& v : coef) v = (rand() / (float)RAND_MAX - 0.5f) * scale;
        for (auto& v : x) v = (rand() / (float)RAND_MAX - 0.5f) * scale;
        CHECK(hipMemcpy(d_coef, coef.data(), 64, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_x, x.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(exact_kernel, dim3(1), dim3(64), 0, 0, d_coef, d_x, d_om, d_of);
        CHECK(hipMemcpy(om.data(), d_om, 1024, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(of.data(), d_of, 1024, hipMemcpyDeviceToHost));
        int diff = 0, nonzero = 0, subnormal = 0;
        for (int i = 0; i < 256; ++i) {
            diff += memcmp(&om[i], &of[i], 4) != 0;
            nonzero += of[i] != 0.f;
            subnormal += of[i] != 0.f && fabsf(of[i]) < 1.17549435e-38f;
        }
        printf("exactness, operand scale %g: %d of 256 results differ in bits from the fmaf chain (%d nonzero, %d subnormal); lane 5: mfma %a fma %a\n",
               scale, diff, nonzero, subnormal, om[20], of[20]);
    }
    // ---- 2. rates
    run<8, 0>("8 pk_fma");
    run<0, 4>("4 mfma");
    run<0, 2>("2 mfma");
    run<8, 4>("8 pk_fma + 4 mfma");
    run<8, 2>("8 pk_fma + 2 mfma");
    run<4, 4>("4 pk_fma + 4 mfma");
    return 0;
}
