// Microbenchmark: the instruction stream of one scoring evaluation (17 fma, 3 mul, 3 add/sub, max, cmp,
// alignbit = 26 VALU instructions), with all operands in VGPRs and with 13 of the fma reading an SGPR
// operand as score_mixed_kernel's do (point data arrives through scalar loads); 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/score_mix_peak.hip -o tools/score_mix_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
// one "evaluation": 17 fma (13 of them with an SGPR operand when SG), 3 mul, 3 add/sub, max, cmp, alignbit
template <bool SG>
__global__ void __launch_bounds__(256) spin(float *out, float seed, float s0, float s1, int iters)
{
    float a[4], b[4];
    for (int c = 0; c < 4; ++c) { a[c] = seed * (threadIdx.x + 1) + c; b[c] = 1.0f + 1e-6f * (threadIdx.x + c); }
    unsigned long long m = ~0ull;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned long long r;
                if (SG) {
                    asm volatile(
                        "v_fma_f32 %0, %1, %3, %0\n v_fma_f32 %0, %1, %4, %0\n v_fma_f32 %0, %1, %3, %0\n v_fma_f32 %0, %1, %4, %0\n"
                        "v_fma_f32 %0, %1, %3, %0\n v_fma_f32 %0, %1, %4, %0\n v_fma_f32 %0, %1, %3, %0\n v_fma_f32 %0, %1, %4, %0\n"
                        "v_fma_f32 %0, %1, %3, %0\n v_fma_f32 %0, %1, %4, %0\n v_fma_f32 %0, %1, %3, %0\n v_fma_f32 %0, %1, %4, %0\n"
                        "v_fma_f32 %0, %1, %3, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n"
                        "v_fma_f32 %0, %1, %0, %0\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                        "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_max_f32 %0, %0, %1\n"
                        "v_cmp_gt_f32_e64 %2, |%0|, %1\n v_alignbit_b32 %0, %0, %1, 31\n"
                        : "+v"(a[c]), "+v"(b[c]), "=s"(r) : "s"(s0), "s"(s1));
                } else {
                    asm volatile(
                        "v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n"
                        "v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n"
                        "v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n"
                        "v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n v_fma_f32 %0, %1, %0, %0\n"
                        "v_fma_f32 %0, %1, %0, %0\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                        "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_sub_f32 %0, %0, %1\n v_max_f32 %0, %0, %1\n"
                        "v_cmp_gt_f32_e64 %2, |%0|, %1\n v_alignbit_b32 %0, %0, %1, 31\n"
                        : "+v"(a[c]), "+v"(b[c]), "=s"(r) : "s"(s0), "s"(s1));
                }
                m &= r;
            }
        }
    }
    float r = (float)(m & 1);
    for (int c = 0; c < 4; ++c) r += a[c] + b[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <bool SG> int run(const char *name, float *out)
{
    const int blocks = 256 * 16, iters = 256;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(spin<SG>, dim3(blocks), dim3(256), 0, 0, out, 1.000001f, 0.999f, 1.001f, iters);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double evals_per_simd = (double)blocks * 4 / 1024 * iters * 16;   // wave-evaluations per SIMD
    printf("%-24s %.3f ms  %.1f ns per 26-instruction evaluation per SIMD = %.2f T lane-instr/s\n", name, best,
           best * 1e6 / evals_per_simd, (double)blocks * 256 * iters * 16 * 26.0 / (best * 1e-3) / 1e12);
    return 0;
}
int main()
{
    float *out; CHECK(hipMalloc(&out, sizeof(float) * 256 * 16 * 256));
    run<false>("vgpr operands", out);
    run<true>("13 fma with sgpr operand", out);
    return 0;
}
