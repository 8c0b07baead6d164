// Probe: Hamming distance of 64-bit rows on v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (e2m1)
// operands - train bits as {0, v} with v in {0.5, 1, 2}, query bits as +-1/v, block scale 2^10 -
// checked against popcounts; then the back-to-back issue time of the instruction.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_fp4_probe.hip -o tools/mfma_fp4_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using i32x8 = int __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void expand_train(uint32_t w, i32x8 &v)
{
    v[0] = (int)(w & 0x11111111u);            // bit 4n   -> 0.5
    v[1] = (int)(w & 0x22222222u);            // bit 4n+1 -> 1.0
    v[2] = (int)(w & 0x44444444u);            // bit 4n+2 -> 2.0
    v[3] = (int)((w >> 3) & 0x11111111u);     // bit 4n+3 -> 0.5
    v[4] = v[5] = v[6] = v[7] = 0;
}
__device__ __forceinline__ void expand_query(uint32_t w, i32x8 &v)
{
    const uint32_t n = ~w;
    v[0] = (int)(0x44444444u | ((n & 0x11111111u) << 3));   // +-2.0
    v[1] = (int)(0x22222222u | ((n & 0x22222222u) << 2));   // +-1.0
    v[2] = (int)(0x11111111u | ((n & 0x44444444u) << 1));   // +-0.5
    v[3] = (int)(0x44444444u | (n & 0x88888888u));          // +-2.0
    v[4] = v[5] = v[6] = v[7] = 0;
}

// train[32][2], query[32][2] words; out[32 train][32 query] = 1024 * (pop(q) - ham)
__global__ void probe(const uint32_t *train, const uint32_t *query, float *out)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    i32x8 a, b;
    expand_train(train[2 * r + h], a);
    expand_query(query[2 * r + h], b);
    f32x16 c = {};
    const int scale_one = 0x7F7F7F7F, scale_1024 = 0x89898989;   // E8M0: 2^(e - 127)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, scale_one, 0, scale_1024);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        out[row * 32 + r] = c[i];
    }
}

__global__ void __launch_bounds__(256) spin(float *out, int iters)
{
    i32x8 a = {}, b = {};
    a[0] = threadIdx.x; b[1] = threadIdx.x * 3;
    f32x16 c[4] = {};
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            c[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[u], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    out[blockIdx.x * 256 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}

int main()
{
    uint32_t ht[64], hq[64];
    srand(7);
    for (int i = 0; i < 64; ++i) {
        ht[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
        hq[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    }
    ht[0] = ht[1] = 0xFFFFFFFFu; hq[0] = hq[1] = 0u; hq[2] = hq[3] = 0xFFFFFFFFu; ht[4] = ht[5] = 0u;
    uint32_t *dt, *dq;
    float *dout, hout[1024];
    CHECK(hipMalloc(&dt, sizeof(ht)));
    CHECK(hipMalloc(&dq, sizeof(hq)));
    CHECK(hipMalloc(&dout, sizeof(float) * 256 * 2048));
    CHECK(hipMemcpy(dt, ht, sizeof(ht), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dq, hq, sizeof(hq), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dt, dq, dout);
    CHECK(hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int tr = 0; tr < 32; ++tr)
        for (int q = 0; q < 32; ++q) {
            int ham = 0, popq = 0;
            for (int h = 0; h < 2; ++h) {
                ham += __builtin_popcount(ht[2 * tr + h] ^ hq[2 * q + h]);
                popq += __builtin_popcount(hq[2 * q + h]);
            }
            const float want = 1024.0f * (float)(popq - ham);
            if (hout[tr * 32 + q] != want && bad++ < 5)
                printf("train %d query %d: got %g want %g\n", tr, q, hout[tr * 32 + q], want);
        }
    printf("fp4 Hamming probe: %d of 1024 wrong\n", bad);

    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = 2048, iters = 2000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, dout, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best)
            best = ms;
    }
    const double per_simd = (double)blocks * 4 / 1024 * iters * 4;
    printf("v_mfma_scale_f32_32x32x64_f8f6f4 (fp4): %.2f ns per MFMA per SIMD, %.0f T bit-pairs/s chip-wide\n",
           best * 1e6 / per_simd, (double)blocks * 4 * iters * 4 * 32 * 32 * 64 / (best * 1e-3) / 1e12);
    return bad != 0;
}
