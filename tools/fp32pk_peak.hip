// Microbenchmark: sustained packed-fp32 VALU instruction rates on one GPU
// (T lane-instructions/s; one v_pk_* lane-instruction works on two floats).
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp32pk_peak.hip -o tools/fp32pk_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum { PK_FMA, PK_FMA_SGPR, PK_MUL, PK_ADD, FMA32, MAXU32, CMP32, ALIGNBIT, FMA64 };

using f32x2 = float __attribute__((ext_vector_type(2)));

template <int MODE, int CHAINS>
__global__ void __launch_bounds__(256) spin(double *out, double seed, int iters)
{
    f32x2 acc[CHAINS], b[CHAINS];
    double dacc[CHAINS], db[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        acc[c] = f32x2{(float)seed * (threadIdx.x + 1) + c, (float)seed + c};
        b[c] = f32x2{1.0f + 1e-6f * (threadIdx.x + c), 1.0f - 1e-6f * c};
        dacc[c] = seed * (threadIdx.x + 1) + c;
        db[c] = 1.0 + 1e-9 * (threadIdx.x + c);
    }
    unsigned long long m = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (MODE == PK_FMA) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(b[c]), "v"(b[(c + 1) % CHAINS]));
                } else if (MODE == PK_FMA_SGPR) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "s"(seed), "v"(b[c]));
                } else if (MODE == PK_MUL) {
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                } else if (MODE == PK_ADD) {
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                } else if (MODE == FMA32) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[c].x) : "v"(b[c].x), "v"(b[c].y));
                } else if (MODE == MAXU32) {
                    asm volatile("v_max_u32 %0, %0, %1" : "+v"(acc[c].x) : "v"(b[c].x));
                } else if (MODE == ALIGNBIT) {
                    asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(acc[c].x) : "v"(b[c].x));
                } else if (MODE == FMA64) {
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dacc[c]) : "v"(db[c]), "v"(db[(c + 1) % CHAINS]));
                } else {
                    unsigned long long r;
                    asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(r) : "v"(acc[c].x), "v"(b[c].x));
                    m ^= r;
                }
            }
        }
    }
    double r = (double)(m & 1);
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
        r += (double)acc[c].x + (double)acc[c].y + dacc[c];
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
    printf("%-16s %.3f ms  %.2f T lane-instr/s\n", name, best, ops / (best * 1e-3) / 1e12);
    return 0;
}

int main()
{
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * 8 * 256));
    run<FMA64>("v_fma_f64", out);
    run<PK_FMA>("v_pk_fma_f32", out);
    run<PK_FMA_SGPR>("v_pk_fma_f32 s", out);
    run<PK_MUL>("v_pk_mul_f32", out);
    run<PK_ADD>("v_pk_add_f32", out);
    run<FMA32>("v_fma_f32", out);
    run<MAXU32>("v_max_u32", out);
    run<CMP32>("v_cmp_lt_f32", out);
    run<ALIGNBIT>("v_alignbit_b32", out);
    return 0;
}
