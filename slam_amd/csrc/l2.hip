// Batched float L2 2-NN (cv2.BFMatcher(NORM_L2).knnMatch, the matcher the reference constructs
// by default for its SIFT descriptors: models/Matcher.py:15) on gfx950.
//
// lane = query, whose descriptor stays in VGPRs; the train rows of the item are walked in index
// order with wave-uniform addresses, so they arrive through scalar loads and cost no VALU; per
// dimension one v_sub_f32 and one v_fma_f32.  Squared distances are accumulated in fp32 in
// dimension order; SIFT descriptors hold integers 0..255, for which every partial sum is an
// exactly representable integer, so the result does not depend on the summation order and
// equals cv2's bit for bit.  The reported distance is sqrtf of it, as cv2 reports for NORM_L2.
// Tie rule as cv2's scan: a train row is admitted only if strictly closer than the current
// k-th best, so equal distances keep the lower train index first.
#include <float.h>

#include "common.h"

namespace slam {
namespace {

struct L2Args {
    const float *q;
    const float *t;
    const int32_t *seg;
    int tiles_per_item;
    int k;
    int mode;
    double thr;
    const float *yq;
    const float *yt;
    int32_t *out_idx;
    float *out_dist;
    uint8_t *keep;
};

template <int D>
__global__ void __launch_bounds__(256) l2_knn_kernel(const L2Args a)
{
    const int item = blockIdx.x / a.tiles_per_item;
    const int tile = blockIdx.x - item * a.tiles_per_item;
    const int32_t *seg = a.seg + SLAM_SEG_STRIDE * item;
    const int qs = seg[0], qn = seg[1], ts = seg[2], tn = seg[3], os = seg[4];
    if (tile * 256 >= qn)
        return;
    const int row = tile * 256 + threadIdx.x;
    const bool ok = row < qn;
    float qv[D];
    const float4 *src = reinterpret_cast<const float4 *>(a.q + (size_t)(qs + (ok ? row : 0)) * D);
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
        const float4 v = src[c];
        qv[4 * c] = v.x;
        qv[4 * c + 1] = v.y;
        qv[4 * c + 2] = v.z;
        qv[4 * c + 3] = v.w;
    }
    float b1 = FLT_MAX, b2 = FLT_MAX;
    int i1 = -1, i2 = -1;
    const float *__restrict__ tbase = a.t + (size_t)ts * D;
    for (int j = 0; j < tn; ++j) {
        const float *__restrict__ tr = tbase + (size_t)j * D;   // wave-uniform: scalar loads
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const float d = qv[c] - tr[c];
            acc = __builtin_fmaf(d, d, acc);
        }
        const bool lt1 = acc < b1, lt2 = acc < b2;
        b2 = lt1 ? b1 : (lt2 ? acc : b2);
        i2 = lt1 ? i1 : (lt2 ? j : i2);
        b1 = lt1 ? acc : b1;
        i1 = lt1 ? j : i1;
    }
    if (!ok)
        return;
    if (a.k < 2) {
        i2 = -1;
        b2 = FLT_MAX;
    }
    const float d0 = i1 >= 0 ? sqrtf(b1) : FLT_MAX;
    const float d1 = i2 >= 0 ? sqrtf(b2) : FLT_MAX;
    const size_t g = (size_t)os + row;
    reinterpret_cast<int2 *>(a.out_idx)[g] = make_int2(i1, i2);
    reinterpret_cast<float2 *>(a.out_dist)[g] = make_float2(d0, d1);
    if (a.mode == 1) {
        a.keep[g] = (i2 >= 0) && ((double)d0 < a.thr * (double)d1);
    } else if (a.mode == 2) {
        bool kp = false;
        if (i1 >= 0) {
            const double dev = fabs((double)a.yq[qs + row] - (double)a.yt[ts + i1]);
            kp = dev <= a.thr;
        }
        a.keep[g] = kp;
    }
}

}  // namespace
}  // namespace slam

using namespace slam;

extern "C" int slam_l2_match(const float *q, const float *t, const int32_t *seg, int batch, int max_nq, int max_nt,
                             int dim, int k, int mode, double thr, const float *yq, const float *yt,
                             int32_t *out_idx, float *out_dist, uint8_t *keep, void *stream)
{
    if (batch < 0 || max_nq < 0 || max_nt < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match: negative size");
    if (dim != 128 && dim != 64)
        return fail(SLAM_ERR_UNSUPPORTED, "slam_l2_match: dim must be 128 or 64 (got %d)", dim);
    if (k != 1 && k != 2)
        return fail(SLAM_ERR_UNSUPPORTED, "slam_l2_match: k must be 1 or 2 (got %d)", k);
    if (mode < 0 || mode > 2)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match: mode must be 0, 1 or 2 (got %d)", mode);
    if (mode == 1 && k != 2)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match: the ratio test (mode 1) needs k = 2");
    if (batch == 0 || max_nq == 0)
        return SLAM_OK;
    if (!q || !seg || !out_idx || !out_dist || (max_nt > 0 && !t))
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match: null buffer");
    if (mode != 0 && !keep)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match: mode %d needs a keep buffer", mode);
    if (mode == 2 && (!yq || !yt))
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match: the row filter (mode 2) needs yq and yt");
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(t)) & 15)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match: descriptor tables must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(out_idx) | reinterpret_cast<uintptr_t>(out_dist)) & 7)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match: outputs must be 8-byte aligned");
    L2Args a;
    a.q = q;
    a.t = t;
    a.seg = seg;
    a.tiles_per_item = (max_nq + 255) / 256;
    a.k = k;
    a.mode = mode;
    a.thr = thr;
    a.yq = yq;
    a.yt = yt;
    a.out_idx = out_idx;
    a.out_dist = out_dist;
    a.keep = keep;
    const long long blocks = (long long)batch * a.tiles_per_item;
    if (blocks > 0x7FFFFFFFLL)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match: grid too large (%lld blocks)", blocks);
    if (dim == 128)
        hipLaunchKernelGGL(l2_knn_kernel<128>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(l2_knn_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("slam_l2_match");
}
