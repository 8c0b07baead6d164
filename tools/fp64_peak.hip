// Microbenchmark: sustained fp64 VALU instruction rates on one GPU (T lane-op/s).
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_peak.hip -o tools/fp64_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum { FMA, MUL, ADD, CMP, FMA_SGPR };

template <int MODE, int CHAINS>
__global__ void __launch_bounds__(256) spin(double *out, double seed, int iters)
{
    double acc[CHAINS], b[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        acc[c] = seed * (threadIdx.x + 1) + c;
        b[c] = 1.0 + 1e-9 * (threadIdx.x + c);
    }
    unsigned long long m = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (MODE == FMA) {
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(b[c]), "v"(b[(c + 1) % CHAINS]));
                } else if (MODE == FMA_SGPR) {
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[c]) : "s"(seed), "v"(b[c]));
                } else if (MODE == MUL) {
                    asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                } else if (MODE == ADD) {
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                } else {
                    unsigned long long r;
                    asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(r) : "v"(acc[c]), "v"(b[c]));
                    m ^= r;
                }
            }
        }
    }
    double r = (double)(m & 1);
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
        r += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
int run(const char *name, double *out)
{
    const int blocks = 256 * 8, threads = 256, iters = 1024;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((spin<MODE, 4>), dim3(blocks), dim3(threads), 0, 0, out, 1.000001, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best)
            best = ms;
    }
    const double ops = (double)blocks * threads * iters * 16.0 * 4;
    printf("%-12s %.3f ms  %.2f T lane-op/s\n", name, best, ops / (best * 1e-3) / 1e12);
    return 0;
}

int main()
{
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * 8 * 256));
    run<FMA>("v_fma_f64", out);
    run<FMA_SGPR>("v_fma_f64 s", out);
    run<MUL>("v_mul_f64", out);
    run<ADD>("v_add_f64", out);
    run<CMP>("v_cmp_lt_f64", out);
    return 0;
}
