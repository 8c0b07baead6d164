// Microbenchmark: sustained rates of the packed 16-bit vector instructions on one GPU
// (T lane-instr/s; one lane-instruction works on two 16-bit halves), with v_fma_f32 beside them.
// Build: hipcc --offload-arch=gfx950 -O3 tools/pk_f16_peak.hip -o tools/pk_f16_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum { FMA32, PKFMA16, PKMUL16, PKADD16, PKMAX16, PKADDU16, PKLSHR16, FMA16, PKFMA16_S, MIX };

template <int MODE, int CHAINS>
__global__ void __launch_bounds__(256) spin(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t acc[CHAINS], b[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        acc[c] = 0x3c003c00u + ((threadIdx.x + c) & 7);   // (1.0, 1.0) as halves, a few low bits set
        b[c] = 0x3c003c00u;
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (MODE == FMA32)
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(b[c]), "v"(b[(c + 1) % CHAINS]));
                else if (MODE == PKFMA16)
                    asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(b[c]), "v"(b[(c + 1) % CHAINS]));
                else if (MODE == PKFMA16_S)
                    asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "s"(seed), "v"(b[c]));
                else if (MODE == PKMUL16)
                    asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                else if (MODE == PKADD16)
                    asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                else if (MODE == PKMAX16)
                    asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                else if (MODE == PKADDU16)
                    asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                else if (MODE == PKLSHR16)
                    asm volatile("v_pk_lshrrev_b16 %0, 15, %1" : "=v"(acc[c]) : "v"(b[c]));
                else if (MODE == FMA16)
                    asm volatile("v_fma_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(b[c]), "v"(b[(c + 1) % CHAINS]));
                else if (MODE == MIX) {
                    // the mix a packed first-level supporter test would issue: mostly fma, some mul/max/add, integer tail
                    if (u % 8 < 5)
                        asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(b[c]), "v"(b[(c + 1) % CHAINS]));
                    else if (u % 8 == 5)
                        asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                    else if (u % 8 == 6)
                        asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
                    else
                        asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(acc[c]) : "v"(b[c]));
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
int run(const char *name, uint32_t *out)
{
    const int blocks = 256 * 8, threads = 256, iters = 1024;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((spin<MODE, 4>), dim3(blocks), dim3(threads), 0, 0, out, 0x3c003c00u, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best)
            best = ms;
    }
    const double ops = (double)blocks * threads * iters * 16.0 * 4;
    printf("%-18s %.3f ms  %.2f T lane-instr/s\n", name, best, ops / (best * 1e-3) / 1e12);
    return 0;
}

int main()
{
    uint32_t *out;
    CHECK(hipMalloc(&out, sizeof(uint32_t) * 256 * 8 * 256));
    run<FMA32>("v_fma_f32", out);
    run<PKFMA16>("v_pk_fma_f16", out);
    run<PKFMA16_S>("v_pk_fma_f16 sgpr", out);
    run<PKMUL16>("v_pk_mul_f16", out);
    run<PKADD16>("v_pk_add_f16", out);
    run<PKMAX16>("v_pk_max_f16", out);
    run<PKADDU16>("v_pk_add_u16", out);
    run<PKLSHR16>("v_pk_lshrrev_b16", out);
    run<FMA16>("v_fma_f16", out);
    run<MIX>("packed mix", out);
    CHECK(hipFree(out));
    return 0;
}
