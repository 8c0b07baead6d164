// Microbenchmark: how do v_mfma_f32_32x32x2_f32 and fp32 VALU instructions share a SIMD?
// Per loop trip a wavefront issues one MFMA followed by N independent v_fma_f32 (8 chains), N = 0..64;
// if the two pipes ran side by side a trip would cost max(MFMA time, N x VALU issue time).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_coexec.hip -o tools/mfma_f32_coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));

template <int N, bool MFMA>
__global__ void __launch_bounds__(256) spin(float *out, float seed, int iters)
{
    extern __shared__ float lds[];
    float acc[8], b[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { acc[c] = seed * (threadIdx.x + 1) + c; b[c] = 1.0f + 1e-6f * (threadIdx.x + c); }
    f16v d0 = {}, d1 = {};
    const float a = seed * threadIdx.x, bb = 0.5f + seed;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (MFMA) {
                if (u == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(d0) : "v"(a), "v"(bb));
                else        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(d1) : "v"(a), "v"(bb));
            }
#pragma unroll
            for (int k = 0; k < N; ++k)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k & 7]) : "v"(b[k & 7]), "v"(b[(k + 1) & 7]));
        }
    }
    float r = lds[0] * 0.0f + d0[3] + d1[5];
#pragma unroll
    for (int c = 0; c < 8; ++c) r += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int N, bool MFMA> int run(float *out, int waves_per_simd)
{
    const int lds = waves_per_simd >= 8 ? 1024 : (160 * 1024 / waves_per_simd) - 512;
    CHECK(hipFuncSetAttribute((const void *)spin<N, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = 256 * 16, iters = 400;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((spin<N, MFMA>), dim3(blocks), dim3(256), lds, 0, out, 1.000001f, iters);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    // trips per SIMD: blocks*4 waves / 1024 SIMDs, iters*2 trips each
    const double trips_per_simd = (double)blocks * 4 / 1024 * iters * 2;
    printf("waves/SIMD %d  %s + %2d v_fma_f32: %.3f ms  %.1f ns per trip per SIMD\n", waves_per_simd, MFMA ? "mfma" : "    ", N, best,
           best * 1e6 / trips_per_simd);
    return 0;
}

int main()
{
    float *out; CHECK(hipMalloc(&out, sizeof(float) * 256 * 16 * 256));
    for (int w : {1, 2, 4, 8}) {
        run<0, true>(out, w); run<16, false>(out, w); run<16, true>(out, w); run<32, false>(out, w); run<32, true>(out, w);
        run<48, false>(out, w); run<48, true>(out, w); run<64, false>(out, w); run<64, true>(out, w);
    }
    return 0;
}
