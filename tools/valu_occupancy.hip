// Microbenchmark: VALU issue rate against waves per SIMD (occupancy limited through the
// dynamic LDS size of a 256-thread block), for the instruction kinds of the scoring filter.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_occupancy.hip -o tools/valu_occupancy
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum { FMA32, MUL32, ADD32, MAX32, ALIGNBIT, MIX };

template <int MODE, int CHAINS>
__global__ void __launch_bounds__(256) spin(float *out, float seed, int iters)
{
    extern __shared__ float lds[];
    float acc[CHAINS], b[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        acc[c] = seed * (threadIdx.x + 1) + c;
        b[c] = 1.0f + 1e-6f * (threadIdx.x + c);
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (MODE == FMA32) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(b[c]), "v"(b[(c + 1) % CHAINS]));
                } else if (MODE == MUL32) {
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                } else if (MODE == ADD32) {
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                } else if (MODE == MAX32) {
                    asm volatile("v_max_f32 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                } else if (MODE == ALIGNBIT) {
                    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(acc[c]) : "v"(b[c]));
                } else {    // 12 fma/mul/add : 1 max : 2 alignbit, as in one filter element (counted as 15)
                    asm volatile(
                        "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %0, %1, %2, %0\n"
                        "v_fma_f32 %0, %1, %2, %0\n v_mul_f32 %0, %0, %1\n v_fma_f32 %0, %1, %2, %0\n"
                        "v_mul_f32 %0, %0, %1\n v_fma_f32 %0, %1, %2, %0\n v_max_f32 %0, %0, %1\n"
                        "v_mul_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                        "v_fma_f32 %0, %1, %2, %0\n v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %0, %0, %1, 31\n"
                        : "+v"(acc[c]) : "v"(b[c]), "v"(b[(c + 1) % CHAINS]));
                }
            }
        }
    }
    float r = lds[0] * 0.0f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
        r += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE, int CHAINS>
int run(const char *name, float *out, int waves_per_simd)
{
    // one 256-thread block = one wave per SIMD; LDS sized so that exactly waves_per_simd blocks fit a CU
    const int lds = waves_per_simd >= 8 ? 1024 : (160 * 1024 / waves_per_simd) - 512;
    CHECK(hipFuncSetAttribute((const void *)spin<MODE, CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = 256 * 24, threads = 256, iters = MODE == MIX ? 64 : 512;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((spin<MODE, CHAINS>), dim3(blocks), dim3(threads), lds, 0, out, 1.000001f, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best)
            best = ms;
    }
    const double ops = (double)blocks * threads * iters * 8.0 * CHAINS * (MODE == MIX ? 15 : 1);
    printf("%-10s chains %d waves/SIMD %d  %.3f ms  %.2f T lane-instr/s\n", name, CHAINS, waves_per_simd, best,
           ops / (best * 1e-3) / 1e12);
    return 0;
}

int main()
{
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 24 * 256));
    const int occ[] = {1, 2, 3, 4, 8};
    for (int w : occ) {
        run<FMA32, 8>("v_fma_f32", out, w);
        run<FMA32, 1>("v_fma_f32", out, w);
        run<MUL32, 8>("v_mul_f32", out, w);
        run<ADD32, 8>("v_add_f32", out, w);
        run<MAX32, 8>("v_max_f32", out, w);
        run<ALIGNBIT, 8>("alignbit", out, w);
        run<MIX, 8>("mix", out, w);
        run<MIX, 1>("mix", out, w);
    }
    return 0;
}
