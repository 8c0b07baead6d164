// Microbenchmark: sustained v_mfma_i32_32x32x32_i8 / v_mfma_i32_16x16x64_i8 rate (TOP/s, 2 ops per MAC).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_i8_peak.hip -o tools/mfma_i8_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void __launch_bounds__(256) spin(int *out, int iters)
{
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
    int r = 0;
    if (SHAPE == 32) {
        i32x16 acc[4] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[c], 0, 0, 0);
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 16; ++e)
                r += acc[c][e];
    } else {
        i32x4 acc[4] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[c], 0, 0, 0);
        for (int c = 0; c < 4; ++c)
            for (int e = 0; e < 4; ++e)
                r += acc[c][e];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int SHAPE>
int run(const char *name, int *out, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, threads = 256, iters = 2048;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((spin<SHAPE>), dim3(blocks), dim3(threads), 0, 0, out, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best)
            best = ms;
    }
    const double mfmas = (double)blocks * 4 * iters * 16.0;
    const double ops = mfmas * (SHAPE == 32 ? 2.0 * 32 * 32 * 32 : 2.0 * 16 * 16 * 64);
    printf("%-28s %d wave(s)/SIMD  %.3f ms  %.0f TOP/s  (%.1f ns per MFMA per SIMD)\n", name, waves_per_simd, best,
           ops / (best * 1e-3) / 1e12, best * 1e6 / (iters * 16.0 * waves_per_simd));
    return 0;
}

int main()
{
    int *out;
    CHECK(hipMalloc(&out, sizeof(int) * 256 * 8 * 256));
    run<32>("v_mfma_i32_32x32x32_i8", out, 1);
    run<32>("v_mfma_i32_32x32x32_i8", out, 2);
    run<16>("v_mfma_i32_16x16x64_i8", out, 1);
    run<16>("v_mfma_i32_16x16x64_i8", out, 2);
    return 0;
}
