// Microbenchmark: sustained int32 VALU rates on one GPU for the instructions the
// matcher is made of.  Prints T lane-op/s per variant.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/valu_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum { XOR_ONLY, BCNT_ONLY, XOR_BCNT, ADD_ONLY, MED3_ONLY };

template <int MODE, int CHAINS>
__global__ void __launch_bounds__(256) spin(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[CHAINS], acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        a[c] = seed * (threadIdx.x + 1) + c;
        acc[c] = c;
    }
    uint32_t m = seed ^ threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (MODE == XOR_ONLY) {
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(acc[c]) : "v"(acc[c]), "v"(m));
                } else if (MODE == BCNT_ONLY) {
                    asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(acc[c]) : "v"(a[c]), "v"(acc[c]));
                } else if (MODE == XOR_BCNT) {
                    uint32_t x;
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(a[c]), "v"(m));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(acc[c]) : "v"(x), "v"(acc[c]));
                } else if (MODE == ADD_ONLY) {
                    asm volatile("v_add_u32 %0, %1, %2" : "=v"(acc[c]) : "v"(acc[c]), "v"(m));
                } else {
                    asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(acc[c]) : "v"(acc[c]), "v"(a[c]), "v"(m));
                }
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
        r += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
int run(const char *name, uint32_t *out, int ops_per_step)
{
    const int blocks = 256 * 8, threads = 256, iters = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((spin<MODE, 4>), dim3(blocks), dim3(threads), 0, 0, out, 12345u, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best)
            best = ms;
    }
    const double ops = (double)blocks * threads * iters * 16.0 * 4 * ops_per_step;
    printf("%-10s %.3f ms  %.2f T lane-op/s\n", name, best, ops / (best * 1e-3) / 1e12);
    return 0;
}

int main()
{
    uint32_t *out;
    CHECK(hipMalloc(&out, sizeof(uint32_t) * 256 * 8 * 256));
    run<ADD_ONLY>("v_add_u32", out, 1);
    run<XOR_ONLY>("v_xor_b32", out, 1);
    run<BCNT_ONLY>("v_bcnt", out, 1);
    run<XOR_BCNT>("xor+bcnt", out, 2);
    run<MED3_ONLY>("v_med3_u32", out, 1);
    return 0;
}
