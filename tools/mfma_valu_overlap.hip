// Microbenchmark: do bf16 MFMAs and fp32 VALU work of different wavefronts overlap on a SIMD?
// Each wavefront alternates a group of 7 v_mfma_f32_32x32x16_bf16 with a group of 270 VALU
// instructions (the filter's mix); times for MFMA only, VALU only and both, against waves/SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

template <bool MFMA, bool VALU, bool DEP>
__global__ void __launch_bounds__(256) spin(float *out, float seed, int iters)
{
    extern __shared__ float lds[];
    float acc[8], b[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        acc[c] = seed * (threadIdx.x + 1) + c;
        b[c] = 1.0f + 1e-6f * (threadIdx.x + c);
    }
    f32x16 d[4] = {};
    uint4 av = make_uint4(threadIdx.x, 2, 3, 4), bv = make_uint4(5, 6, threadIdx.x, 8);
    for (int i = 0; i < iters; ++i) {
        if (MFMA) {
#pragma unroll
            for (int o = 0; o < 7; ++o)
                d[o & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                                   d[o & 3], 0, 0, 0);
        }
        if (VALU) {
            if (DEP) {      // the VALU group consumes the MFMA results, as the filter does
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    acc[c] += d[c & 3][c];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    asm volatile(
                        "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %0, %1, %2, %0\n"
                        "v_fma_f32 %0, %1, %2, %0\n v_mul_f32 %0, %0, %1\n v_fma_f32 %0, %1, %2, %0\n"
                        "v_mul_f32 %0, %0, %1\n v_fma_f32 %0, %1, %2, %0\n v_max_f32 %0, %0, %1\n"
                        "v_mul_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                        "v_fma_f32 %0, %1, %2, %0\n v_add_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n"
                        "v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %0, %0, %1, 31\n"
                        : "+v"(acc[c]) : "v"(b[c]), "v"(b[(c + 1) & 7]));
                }
            }
        }
    }
    float r = lds[0] * 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        r += acc[c];
#pragma unroll
    for (int o = 0; o < 4; ++o)
        r += d[o][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <bool MFMA, bool VALU, bool DEP>
int run(const char *name, float *out, int waves_per_simd)
{
    const int lds = waves_per_simd >= 8 ? 1024 : (160 * 1024 / waves_per_simd) - 512;
    CHECK(hipFuncSetAttribute((const void *)spin<MFMA, VALU, DEP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = 256 * 24, threads = 256, iters = 200;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((spin<MFMA, VALU, DEP>), dim3(blocks), dim3(threads), lds, 0, out, 1.000001f, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best)
            best = ms;
    }
    // per SIMD: blocks * 4 waves / 1024 SIMDs wave-iterations
    const double groups_per_simd = (double)blocks * 4 / 1024 * iters;
    printf("%-12s waves/SIMD %d  %.3f ms  %.1f ns per group per SIMD\n", name, waves_per_simd, best,
           best * 1e6 / groups_per_simd);
    return 0;
}

int main()
{
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 24 * 256));
    const int occ[] = {1, 2, 3, 4, 8};
    for (int w : occ) {
        run<true, false, false>("mfma", out, w);
        run<false, true, false>("valu", out, w);
        run<true, true, false>("both", out, w);
        run<true, true, true>("both dep", out, w);
    }
    return 0;
}
