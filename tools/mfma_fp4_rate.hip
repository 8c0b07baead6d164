// Microbenchmark: what the fp4 matrix-core instruction of the matcher sustains, and at what clock.
// A bare loop of v_mfma_f32_32x32x64_f8f6f4 (or its v_mfma_scale form), four independent
// accumulators per wavefront, timed with
//   * operands all zero, operands as the matcher makes them (train bits expanded to {0, v} and
//     -+0.5, query bits to +-1/v: about half of the train slots zero), and dense operands (raw
//     random words: every slot a non-zero fp4 value);
//   * one and two wavefronts per SIMD.
// Each run also reads the shader clock counter (s_memtime) against the constant 100 MHz counter
// (s_memrealtime), which gives the frequency the chip held while the loop ran.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_fp4_rate.hip -o tools/mfma_fp4_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using i32x8 = int __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// DATA: 0 zeros, 1 as the matcher expands descriptor bits, 2 dense random
template <bool SCALED, int DATA>
__global__ void __launch_bounds__(256) spin(float *out, unsigned long long *clk, int iters)
{
    i32x8 a[2] = {}, b[2] = {};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t wt = mix(threadIdx.x * 2 + u + 1), wq = mix(threadIdx.x * 2 + u + 977);
        if (DATA == 1) {
            a[u][0] = (int)(wt & 0x11111111u);
            a[u][1] = (int)(wt & 0x22222222u);
            a[u][2] = (int)(wt & 0x44444444u);
            a[u][3] = (int)((wt & 0x88888888u) | 0x11111111u);
            const uint32_t n = ~wq;
            b[u][0] = (int)(0x44444444u | ((n & 0x11111111u) << 3));
            b[u][1] = (int)(0x22222222u | ((n & 0x22222222u) << 2));
            b[u][2] = (int)(0x11111111u | ((n & 0x44444444u) << 1));
            b[u][3] = (int)(0x22222222u | (wq & 0x88888888u));
        } else if (DATA == 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[u][c] = (int)mix(wt + c);
                b[u][c] = (int)mix(wq + c);
            }
        }
    }
    f32x16 c[4] = {};
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (SCALED)
                c[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[u & 1], b[u & 1], c[u], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            else
                c[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[u & 1], b[u & 1], c[u], 4, 4, 0, 0, 0, 0);
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = w1 - w0;
    }
}

template <bool SCALED, int DATA>
int run(const char *name, float *out, unsigned long long *clk, int waves_per_simd)
{
    // one block = 4 waves = one wave per SIMD of a CU; LDS request limits blocks per CU
    const int lds = waves_per_simd >= 2 ? 64 * 1024 : 120 * 1024;
    CHECK(hipFuncSetAttribute((const void *)spin<SCALED, DATA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = 256 * waves_per_simd * 4, iters = 4000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((spin<SCALED, DATA>), dim3(blocks), dim3(256), lds, 0, out, clk, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) {
            best = ms;
            CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
        }
    }
    const double mfma_per_simd = (double)blocks * 4 / 1024 * iters * 4;
    const double ns = best * 1e6 / mfma_per_simd;
    const double ghz = h[1] ? (double)h[0] / ((double)h[1] / 0.1) : 0.0;   // shader ticks per ns (100 MHz wall counter)
    printf("%-9s %-8s %d wave/SIMD: %6.2f ns per MFMA per SIMD = %5.0f TOP/s chip-wide, counter ratio %.2f GHz -> %.1f cycles per MFMA\n",
           SCALED ? "scaled" : "unscaled", name, waves_per_simd, ns, 2.0 * 32 * 32 * 64 * 1024 / ns / 1e3, ghz, ns * ghz);
    return 0;
}

int main()
{
    float *out;
    unsigned long long *clk;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 4096));
    CHECK(hipMalloc(&clk, 16));
    for (int w = 1; w <= 2; ++w) {
        run<true, 0>("zeros", out, clk, w);
        run<false, 0>("zeros", out, clk, w);
        run<true, 1>("matcher", out, clk, w);
        run<false, 1>("matcher", out, clk, w);
        run<true, 2>("dense", out, clk, w);
        run<false, 2>("dense", out, clk, w);
    }
    return 0;
}
