// Microbenchmark for a scoring loop whose rigid transform runs on the fp32 matrix cores.
//
//  1. What v_mfma_f32_32x32x2_f32 computes, bit for bit: D = C + a0*b0 + a1*b1 against fused and unfused
//     evaluation orders on random operands (the guard band of the scoring filter is derived from it).
//  2. The rate of the loop: per 32 x 32 tile (32 hypotheses x 32 points) 16 evaluations per lane of the 16
//     VALU instructions that remain (4 fma residuals, 2 mul + 2 fma squares, add, max, mul, sub, add, fma,
//     cmp, alignbit) with and without the 7 MFMAs that produce their inputs, two accumulator sets.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/score_mfma_proxy.hip -o tools/score_mfma_proxy
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f16v __attribute__((ext_vector_type(16)));

// ---- 1. arithmetic of the instruction
__global__ void mfma_once(const float *A, const float *B, const float *C, float *D)
{
    const int lane = threadIdx.x;                 // one wavefront
    const float a = A[(lane & 31) * 2 + (lane >> 5)];      // A[i][k], i = lane % 32, k = lane / 32
    const float b = B[(lane >> 5) * 32 + (lane & 31)];     // B[k][j], k = lane / 32, j = lane % 32
    f16v c;
    for (int v = 0; v < 16; ++v)
        c[v] = C[(8 * (v / 4) + 4 * (lane >> 5) + (v % 4)) * 32 + (lane & 31)];
    const f16v d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int v = 0; v < 16; ++v)
        D[(8 * (v / 4) + 4 * (lane >> 5) + (v % 4)) * 32 + (lane & 31)] = d[v];
}

static int arithmetic()
{
    const int trials = 400;
    float *A, *B, *C, *D;
    CHECK(hipMallocManaged(&A, 64 * 4)); CHECK(hipMallocManaged(&B, 64 * 4));
    CHECK(hipMallocManaged(&C, 1024 * 4)); CHECK(hipMallocManaged(&D, 1024 * 4));
    long n = 0, bad[5] = {0, 0, 0, 0, 0};
    srand(7);
    auto rnd = [](int spread) { return (float)((rand() / (double)RAND_MAX - 0.5) * exp2((double)(rand() % (2 * spread + 1) - spread))); };
    for (int t = 0; t < trials; ++t) {
        const int spread = (t % 4 == 0) ? 0 : (t % 4 == 1) ? 6 : (t % 4 == 2) ? 20 : 40;
        for (int i = 0; i < 64; ++i) { A[i] = rnd(spread); B[i] = rnd(spread); }
        for (int i = 0; i < 1024; ++i) C[i] = (t % 8 == 7) ? 0.0f : rnd(spread);
        hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, A, B, C, D);
        CHECK(hipDeviceSynchronize());
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                const float a0 = A[i * 2], a1 = A[i * 2 + 1], b0 = B[j], b1 = B[32 + j], c = C[i * 32 + j], d = D[i * 32 + j];
                const float cand[5] = {
                    fmaf(a1, b1, fmaf(a0, b0, c)),                    // fused chain, k = 0 first
                    fmaf(a0, b0, fmaf(a1, b1, c)),                    // fused chain, k = 1 first
                    (float)((double)c + (double)a0 * b0 + (double)a1 * b1),   // one rounding at the end (exact for fp32 products)
                    (c + a0 * b0) + a1 * b1,                          // every product and sum rounded
                    fmaf(a1, b1, a0 * b0) + c};
                for (int k = 0; k < 5; ++k)
                    bad[k] += memcmp(&cand[k], &d, 4) != 0;
                ++n;
            }
    }
    printf("v_mfma_f32_32x32x2_f32 vs host models over %ld results: mismatches fma(a1,b1,fma(a0,b0,c)) %ld | "
           "fma(a0,b0,fma(a1,b1,c)) %ld | single rounding %ld | unfused %ld | fma(a1,b1,a0*b0)+c %ld\n",
           n, bad[0], bad[1], bad[2], bad[3], bad[4]);
    return 0;
}

// ---- 2. rate of the loop
template <bool MFMA, bool EVAL>
__global__ void __launch_bounds__(256, 2) tile_loop(const float4 *__restrict__ recs, int n_tiles, float *out, float crx, float c1)
{
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    // hypothesis-side operands of this wavefront (any finite numbers)
    const float s = 1.0f + 0.001f * (blockIdx.x % 7) + 0.01f * (threadIdx.x >> 6);
    const float az0 = half ? 0.01f * s : 0.02f, az1 = half ? 0.3f : 0.99f * s;
    const float ax0 = half ? 3.0f : 350.0f * s, ax1 = half ? -2.0f * s : 10.0f;
    const float ay0 = half ? 350.0f * s : 1.0f, ay1 = half ? 5.0f : -3.0f * s;
    const float ar = half ? 1e-9f : 1e-12f;
    f16v Z[2], X[2], Y[2], R[2];
    unsigned bits[16];
    for (int v = 0; v < 16; ++v)
        bits[v] = 0;
    unsigned long long any = 0;
    auto load = [&](int t, float4 &r0, float4 &r1) {
        const float4 *p = recs + 2 * ((size_t)(t % 64) * 32 + col);
        r0 = p[0];
        r1 = p[1];
    };
    auto mfma = [&](int buf, const float4 &r0, const float4 &r1) __attribute__((always_inline)) {
        const float b0 = half ? r0.y : r0.x, b1 = half ? 1.0f : r0.z, b2 = half ? 1.0f : r1.w;
        f16v zero;
        for (int v = 0; v < 16; ++v)
            zero[v] = 0.0f;
        if (MFMA) {
            Z[buf] = __builtin_amdgcn_mfma_f32_32x32x2f32(az0, b0, zero, 0, 0, 0);
            X[buf] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax0, b0, zero, 0, 0, 0);
            Y[buf] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay0, b0, zero, 0, 0, 0);
            R[buf] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, b2, zero, 0, 0, 0);
            Z[buf] = __builtin_amdgcn_mfma_f32_32x32x2f32(az1, b1, Z[buf], 0, 0, 0);
            X[buf] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax1, b1, X[buf], 0, 0, 0);
            Y[buf] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay1, b1, Y[buf], 0, 0, 0);
        } else {
            // stand-in that costs nothing: the compiler must take the accumulators as changed from tile to tile
            Z[buf][0] = b1 * az1; X[buf][0] = b0 * ax0; Y[buf][0] = b0 * ay0; R[buf][0] = b2 * ar;
            for (int v = 0; v < 16; ++v)
                asm volatile("" : "+v"(Z[buf][v]), "+v"(X[buf][v]), "+v"(Y[buf][v]), "+v"(R[buf][v]));
        }
    };
    auto evals = [&](int buf, const float4 &r0, const float4 &r1) __attribute__((always_inline)) {
        const float ul = r0.w, vl = r1.x, ur = r1.y, vr = r1.z;
        if (!EVAL) {
            for (int v = 0; v < 16; ++v)
                asm volatile("" :: "v"(Z[buf][v]), "v"(X[buf][v]), "v"(Y[buf][v]), "v"(R[buf][v]));
            return;
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float hz = Z[buf][v], gx = X[buf][v], gy = Y[buf][v];
            const float gxr = gx + crx;
            const float dxl = __builtin_fmaf(ul, hz, -gx), dyl = __builtin_fmaf(vl, hz, -gy);
            const float dxr = __builtin_fmaf(ur, hz, -gxr), dyr = __builtin_fmaf(vr, hz, -gy);
            const float sl = __builtin_fmaf(dxl, dxl, dyl * dyl), sr = __builtin_fmaf(dxr, dxr, dyr * dyr);
            const float m = fmaxf(sl, sr), w = hz * hz, d = m - w, e = m + w;
            const float r = __builtin_fmaf(c1, e, R[buf][v]);
            bits[v] = __builtin_amdgcn_alignbit(bits[v], __float_as_uint(d), 31);
            any |= __ballot(!(fabsf(d) > r));
        }
    };
    // ring of four record buffers, loads three tiles ahead (pinned where they are written: the scheduler
    // otherwise sinks them to their first use); unrolled by four so that every index is static
    float4 pa[4], pb[4];
    load(0, pa[0], pb[0]);
    load(1, pa[1], pb[1]);
    load(2, pa[2], pb[2]);
    mfma(0, pa[0], pb[0]);
    for (int t = 0; t < n_tiles; t += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            __builtin_amdgcn_sched_barrier(0);
            load(t + u + 3, pa[(u + 3) & 3], pb[(u + 3) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            mfma((u + 1) & 1, pa[(u + 1) & 3], pb[(u + 1) & 3]);
            evals(u & 1, pa[u], pb[u]);
        }
    }
    unsigned acc = (unsigned)(any & 1);
    for (int v = 0; v < 16; ++v)
        acc += __popc(bits[v]);
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)acc;
}

template <bool MFMA, bool EVAL> static int rate(const char *name, const float4 *recs, float *out)
{
    const int blocks = 256 * 8, n_tiles = 512;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((tile_loop<MFMA, EVAL>), dim3(blocks), dim3(256), 0, 0, recs, n_tiles, out, -190.0f, 3e-6f);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double evals = (double)blocks * 4 * n_tiles * 1024.0;
    printf("%-28s %.3f ms  %.3f T evaluations/s  (%.1f T lane-instr/s at 16 per evaluation)\n", name, best,
           evals / (best * 1e-3) / 1e12, 16.0 * evals / (best * 1e-3) / 1e12);
    return 0;
}

int main()
{
    if (arithmetic()) return 1;
    float4 *recs; float *out;
    const int n_pts = 64 * 32;
    CHECK(hipMallocManaged(&recs, sizeof(float4) * 2 * n_pts));
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 8 * 256));
    srand(3);
    for (int i = 0; i < n_pts; ++i) {
        const float X = (rand() % 4000) / 100.0f - 20, Y = (rand() % 500) / 100.0f - 3, Zz = 4 + (rand() % 1000) / 10.0f;
        recs[2 * i] = make_float4(X, Y, Zz, (rand() % 600) - 300.0f);
        recs[2 * i + 1] = make_float4((rand() % 180) - 90.0f, (rand() % 600) - 300.0f, (rand() % 180) - 90.0f, (fabsf(X) + fabsf(Y) + Zz) * (fabsf(X) + fabsf(Y) + Zz));
    }
    CHECK(hipDeviceSynchronize());
    if (rate<false, true>("16 VALU per evaluation alone", recs, out)) return 1;
    if (rate<true, false>("7 fp32 MFMAs per tile alone", recs, out)) return 1;
    if (rate<true, true>("both", recs, out)) return 1;
    return 0;
}
