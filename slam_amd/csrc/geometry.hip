// DLT triangulation and four-image consensus join (gfx950).
//
// All fp64 arithmetic here is compiled with -ffp-contract=off and written in the
// reference's operation order, one IEEE rounding per operation, so supporter
// decisions are reproducible bit for bit on any IEEE machine.
#include "common.h"
#include "score_records.h"

namespace slam {

namespace {

struct Mat34 {
    double m[12];
};

// ---------------------------------------------------------------------------
// DLT: null vector of A = [x1*P3-P1; y1*P3-P2; x2*Q3-Q1; y2*Q3-Q2] by one-sided
// Jacobi on the columns of A (utils/utils.py:527-560 asks LAPACK for vh[-1]).
// One lane per point; A and V (32 doubles) stay in registers because every
// index below is a compile-time constant after unrolling.
constexpr int DLT_MAX_SWEEPS = 30;
constexpr double DLT_TOL2 = 1e-30;

__device__ __forceinline__ void dlt_point(const Mat34 &P, const Mat34 &Q, double x1, double y1, double x2, double y2,
                                          double &X, double &Y, double &Z)
{
    double a[4][4], v[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        a[0][c] = x1 * P.m[8 + c] - P.m[c];
        a[1][c] = y1 * P.m[8 + c] - P.m[4 + c];
        a[2][c] = x2 * Q.m[8 + c] - Q.m[c];
        a[3][c] = y2 * Q.m[8 + c] - Q.m[4 + c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            v[r][c] = (r == c) ? 1.0 : 0.0;

    // squared norm of column c / inner product of columns p and q, as fma chains
    auto dot = [&](int p, int q) __attribute__((always_inline)) {
        return __builtin_fma(a[3][p], a[3][q], __builtin_fma(a[2][p], a[2][q], __builtin_fma(a[1][p], a[1][q], a[0][p] * a[0][q])));
    };
    for (int sweep = 0; sweep < DLT_MAX_SWEEPS; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double alpha = dot(p, p), beta = dot(q, q), gamma = dot(p, q);
                if (gamma * gamma <= DLT_TOL2 * (alpha * beta))
                    continue;
                rotated = true;
                // tan = g / D with d = beta - alpha, g = 2 gamma, D = d + sign(d) hypot(d, g) (the smaller
                // root); (cs, sn) = (|D|, sign(D) g) / hypot(D, g): two square roots and one division
                const double d = beta - alpha, g = 2.0 * gamma;
                const double gg = g * g;
                const double r = sqrt(__builtin_fma(d, d, gg));
                const double D = d + __builtin_copysign(r, d);
                const double inv = 1.0 / sqrt(__builtin_fma(D, D, gg));
                const double cs = __builtin_fabs(D) * inv;
                const double sn = (D < 0.0 ? -g : g) * inv;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const double ap = a[r4][p], aq = a[r4][q];
                    a[r4][p] = __builtin_fma(cs, ap, -(sn * aq));
                    a[r4][q] = __builtin_fma(sn, ap, cs * aq);
                    const double vp = v[r4][p], vq = v[r4][q];
                    v[r4][p] = __builtin_fma(cs, vp, -(sn * vq));
                    v[r4][q] = __builtin_fma(sn, vp, cs * vq);
                }
            }
        }
        if (!rotated)
            break;
    }
    double nmin = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double n2 = dot(c, c);
        if (c == 0 || n2 < nmin) {
            nmin = n2;
            s0 = v[0][c];
            s1 = v[1][c];
            s2 = v[2][c];
            s3 = v[3][c];
        }
    }
    X = s0 / s3;
    Y = s1 / s3;
    Z = s2 / s3;
}

__global__ void __launch_bounds__(256) dlt_kernel(const Mat34 P, const Mat34 Q, const float *__restrict__ x1,
                                                  const float *__restrict__ y1, const float *__restrict__ x2,
                                                  const float *__restrict__ y2, int n, double *__restrict__ X,
                                                  double *__restrict__ Y, double *__restrict__ Z,
                                                  uint8_t *__restrict__ valid)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float fx1 = x1[i], fx2 = x2[i];
    double x, y, z;
    dlt_point(P, Q, (double)fx1, (double)y1[i], (double)fx2, (double)y2[i], x, y, z);
    X[i] = x;
    Y[i] = y;
    Z[i] = z;
    valid[i] = !(z <= 0.0 || fx1 < fx2);
}

struct MatchedArgs {
    const float *xq, *yq, *xt, *yt;
    const int32_t *seg;
    int tiles_per_item;
    const int32_t *idx;
    const uint8_t *keep;
    double *X, *Y, *Z;
    uint8_t *valid;
    int32_t *stereo_map;
};

// A block takes DLT_TILE keypoints of one frame.  Only the kept matches are triangulated (about
// 60 % of the keypoints): their local indices are compacted into LDS first, so that the
// wavefronts running the Jacobi iteration are full.  The results go back through LDS as well: the
// compacted lanes would otherwise store 8 bytes each into scattered rows (partial memory lines:
// the write traffic measured twice its size); instead the tile's X, Y, Z and flags are written
// once, row after row, NaN rows of unmatched keypoints included.
constexpr int DLT_TILE = 512;

__global__ void __launch_bounds__(256) dlt_matched_kernel(const Mat34 P, const Mat34 Q, const MatchedArgs a)
{
    __shared__ int list[DLT_TILE];
    __shared__ int wave_base[DLT_TILE / 64 + 1];
    __shared__ double out[3][DLT_TILE];
    __shared__ uint8_t flag[DLT_TILE];
    const int item = blockIdx.x / a.tiles_per_item;
    const int tile = blockIdx.x - item * a.tiles_per_item;
    const int32_t *seg = a.seg + SLAM_SEG_STRIDE * item;
    const int qs = seg[0], qn = seg[1], ts = seg[2], os = seg[4];
    const int i0 = tile * DLT_TILE;
    if (i0 >= qn)
        return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pass 1: which keypoints have a kept match (per-wavefront counts), NaN / 0 for every row of the tile
    int jj[DLT_TILE / 256];
    unsigned long long mask[DLT_TILE / 256];
    const double nan = __builtin_nan("");
#pragma unroll
    for (int e = 0; e < DLT_TILE / 256; ++e) {
        const int l = e * 256 + tid;
        const int i = i0 + l;
        int j = -1;
        if (i < qn) {
            const size_t g = (size_t)os + i;
            if (a.keep[g])
                j = a.idx[2 * g];
            if (a.stereo_map)
                a.stereo_map[g] = j;
        }
        out[0][l] = nan;
        out[1][l] = nan;
        out[2][l] = nan;
        flag[l] = 0;
        jj[e] = j;
        mask[e] = __ballot(j >= 0);
        if (lane == 0)
            wave_base[e * 4 + wave] = __popcll(mask[e]);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
#pragma unroll
        for (int w = 0; w < DLT_TILE / 64; ++w) {
            const int c = wave_base[w];
            wave_base[w] = run;
            run += c;
        }
        wave_base[DLT_TILE / 64] = run;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < DLT_TILE / 256; ++e) {
        if (jj[e] >= 0) {
            const int pos = wave_base[e * 4 + wave] + __popcll(mask[e] & ((1ull << lane) - 1ull));
            list[pos] = (e * 256 + tid) | (jj[e] << 9);       // local keypoint (9 bits) and its right keypoint
        }
    }
    __syncthreads();
    // pass 2: the kept matches, densely
    const int count = wave_base[DLT_TILE / 64];
    for (int k = tid; k < count; k += 256) {
        const int l = list[k] & (DLT_TILE - 1), j = list[k] >> 9;
        const int i = i0 + l;
        const float fx1 = a.xq[qs + i], fx2 = a.xt[ts + j];
        double x, y, z;
        dlt_point(P, Q, (double)fx1, (double)a.yq[qs + i], (double)fx2, (double)a.yt[ts + j], x, y, z);
        out[0][l] = x;
        out[1][l] = y;
        out[2][l] = z;
        flag[l] = !(z <= 0.0 || fx1 < fx2);
    }
    __syncthreads();
    // pass 3: the tile's rows, contiguously
    const int n = min(DLT_TILE, qn - i0);
    const size_t g0 = (size_t)os + i0;
    for (int l = tid; l < n; l += 256) {
        a.X[g0 + l] = out[0][l];
        a.Y[g0 + l] = out[1][l];
        a.Z[g0 + l] = out[2][l];
        a.valid[g0 + l] = flag[l];
    }
}

// ---------------------------------------------------------------------------
// Four-image join: one block per frame pair, order-preserving compaction.
struct JoinArgs {
    const int32_t *pair;
    const int32_t *t_idx;
    const uint8_t *t_keep;
    const int32_t *stereo_map;
    const uint8_t *valid3d;
    const double *X, *Y, *Z;
    const float *xL, *yL, *xR, *yR;
    int cap;
    int32_t *rows;
    double *cX, *cY, *cZ;
    float *cuL, *cvL, *cuR, *cvR;
    int32_t *count;
    // optional: the scoring workspace of the same point sets, filled on the way
    double *score_work;
    size_t n_rec;
    double fx, fy, cx, cy, thr;
};

constexpr int JOIN_BLOCK = 1024;   // one block per frame pair: wide, so that a pair is a couple of trips
constexpr int JOIN_WAVES = JOIN_BLOCK / 64;

__global__ void __launch_bounds__(JOIN_BLOCK) join_kernel(const JoinArgs a)
{
    __shared__ int wave_tot[JOIN_WAVES];
    __shared__ unsigned long long red[3][JOIN_WAVES];
    unsigned long long am = 0ull, um = 0ull, wm = 0ull;
    const int32_t *pr = a.pair + SLAM_PAIR_STRIDE * blockIdx.x;
    const int p0 = pr[0], pn = pr[1], c0 = pr[2], cn = pr[3], t0 = pr[4], r0 = pr[5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int base = 0;
    for (int q0 = 0; q0 < pn; q0 += JOIN_BLOCK) {
        const int q = q0 + tid;
        bool take = false;
        int t = -1, pr_right = -1, cr_right = -1;
        if (q < pn && a.t_keep[t0 + q]) {
            t = a.t_idx[2 * (size_t)(t0 + q)];
            if (t >= 0 && t < cn) {
                pr_right = a.stereo_map[p0 + q];
                cr_right = a.stereo_map[c0 + t];
                take = pr_right >= 0 && cr_right >= 0 && a.valid3d[p0 + q];
            }
        }
        const unsigned long long m = __ballot(take);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0)
            wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = base, total = 0;
#pragma unroll
        for (int w = 0; w < JOIN_WAVES; ++w) {
            off += w < wave ? wave_tot[w] : 0;
            total += wave_tot[w];
        }
        const int slot = off + before;
        if (take && slot < a.cap) {
            const size_t o = (size_t)blockIdx.x * a.cap + slot;
            reinterpret_cast<int4 *>(a.rows)[o] = make_int4(q, pr_right, t, cr_right);
            const double X = a.X[p0 + q], Y = a.Y[p0 + q], Z = a.Z[p0 + q];
            const float uL = a.xL[c0 + t], vL = a.yL[c0 + t], uR = a.xR[r0 + cr_right], vR = a.yR[r0 + cr_right];
            a.cX[o] = X;
            a.cY[o] = Y;
            a.cZ[o] = Z;
            a.cuL[o] = uL;
            a.cvL[o] = vL;
            a.cuR[o] = uR;
            a.cvR[o] = vR;
            if (a.score_work)
                score_write_record(a.score_work, a.n_rec, o, X, Y, Z, uL, vL, uR, vR, a.fx, a.fy, a.cx, a.cy, a.thr, am, um,
                                   wm);
        }
        base += total;
        __syncthreads();
    }
    if (tid == 0)
        a.count[blockIdx.x] = base;
    if (a.score_work) {
        // the set's magnitude bounds: one block owns the set, so they are stored, not merged
        am = wave_max_u64(am);
        um = wave_max_u64(um);
        wm = wave_max_u64(wm);
        if (lane == 0) {
            red[0][wave] = am;
            red[1][wave] = um;
            red[2][wave] = wm;
        }
        __syncthreads();
        if (tid < 4) {
            unsigned long long m = 0ull;
            if (tid < 3) {
                m = red[tid][0];
#pragma unroll
                for (int w = 1; w < JOIN_WAVES; ++w)
                    m = red[tid][w] > m ? red[tid][w] : m;
            }
            reinterpret_cast<unsigned long long *>(score_work_hdr(a.score_work, a.n_rec))[SCORE_HDR * blockIdx.x + tid] = m;
        }
    }
}

Mat34 load34(const double *p)
{
    Mat34 m;
    for (int i = 0; i < 12; ++i)
        m.m[i] = p[i];
    return m;
}

}  // namespace

}  // namespace slam

using namespace slam;

extern "C" int slam_triangulate_dlt(const double *P, const double *Q, const float *x1, const float *y1,
                                    const float *x2, const float *y2, int n, double *X, double *Y, double *Z,
                                    uint8_t *valid, void *stream)
{
    if (n < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_triangulate_dlt: n = %d", n);
    if (!P || !Q)
        return fail(SLAM_ERR_BAD_ARG, "slam_triangulate_dlt: null camera matrix");
    if (n == 0)
        return SLAM_OK;
    if (!x1 || !y1 || !x2 || !y2 || !X || !Y || !Z || !valid)
        return fail(SLAM_ERR_BAD_ARG, "slam_triangulate_dlt: null buffer");
    hipLaunchKernelGGL(dlt_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, load34(P), load34(Q), x1,
                       y1, x2, y2, n, X, Y, Z, valid);
    return check_launch("slam_triangulate_dlt");
}

extern "C" int slam_triangulate_matched(const double *P, const double *Q, const float *xq, const float *yq,
                                        const float *xt, const float *yt, const int32_t *seg, int batch, int max_nq,
                                        const int32_t *idx, const uint8_t *keep, double *X, double *Y, double *Z,
                                        uint8_t *valid, int32_t *stereo_map, void *stream)
{
    if (batch < 0 || max_nq < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_triangulate_matched: negative size");
    if (!P || !Q)
        return fail(SLAM_ERR_BAD_ARG, "slam_triangulate_matched: null camera matrix");
    if (batch == 0 || max_nq == 0)
        return SLAM_OK;
    if (!xq || !yq || !xt || !yt || !seg || !idx || !keep || !X || !Y || !Z || !valid)
        return fail(SLAM_ERR_BAD_ARG, "slam_triangulate_matched: null buffer");
    MatchedArgs a;
    a.xq = xq;
    a.yq = yq;
    a.xt = xt;
    a.yt = yt;
    a.seg = seg;
    a.tiles_per_item = (max_nq + DLT_TILE - 1) / DLT_TILE;
    a.idx = idx;
    a.keep = keep;
    a.X = X;
    a.Y = Y;
    a.Z = Z;
    a.valid = valid;
    a.stereo_map = stereo_map;
    const long long blocks = (long long)batch * a.tiles_per_item;
    if (blocks > 0x7FFFFFFFLL)
        return fail(SLAM_ERR_BAD_ARG, "slam_triangulate_matched: grid too large");
    hipLaunchKernelGGL(dlt_matched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, load34(P),
                       load34(Q), a);
    return check_launch("slam_triangulate_matched");
}

extern "C" int slam_join_consensus(const int32_t *pair, int n_pairs, int max_nq, const int32_t *t_idx,
                                   const uint8_t *t_keep, const int32_t *stereo_map, const uint8_t *valid3d,
                                   const double *X, const double *Y, const double *Z, const float *xL, const float *yL,
                                   const float *xR, const float *yR, int cap, int32_t *rows, double *cX, double *cY,
                                   double *cZ, float *cuL, float *cvL, float *cuR, float *cvR, int32_t *count,
                                   void *stream)
{
    return slam_join_consensus_prepared(pair, n_pairs, max_nq, t_idx, t_keep, stereo_map, valid3d, X, Y, Z, xL, yL, xR,
                                        yR, cap, rows, cX, cY, cZ, cuL, cvL, cuR, cvR, count, nullptr, 0.0, nullptr,
                                        stream);
}

extern "C" int slam_join_consensus_prepared(const int32_t *pair, int n_pairs, int max_nq, const int32_t *t_idx,
                                            const uint8_t *t_keep, const int32_t *stereo_map, const uint8_t *valid3d,
                                            const double *X, const double *Y, const double *Z, const float *xL,
                                            const float *yL, const float *xR, const float *yR, int cap, int32_t *rows,
                                            double *cX, double *cY, double *cZ, float *cuL, float *cvL, float *cuR,
                                            float *cvR, int32_t *count, const double *K, double thresh,
                                            double *score_work, void *stream)
{
    (void)max_nq;
    if (score_work && !K)
        return fail(SLAM_ERR_BAD_ARG, "slam_join_consensus_prepared: a scoring workspace needs the intrinsics");
    if (reinterpret_cast<uintptr_t>(score_work) & 15)
        return fail(SLAM_ERR_BAD_ARG, "slam_join_consensus_prepared: score_work must be 16-byte aligned");
    if (n_pairs < 0 || cap < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_join_consensus: negative size");
    if (n_pairs == 0)
        return SLAM_OK;
    if (!pair || !t_idx || !t_keep || !stereo_map || !valid3d || !X || !Y || !Z || !xL || !yL || !xR || !yR || !rows ||
        !cX || !cY || !cZ || !cuL || !cvL || !cuR || !cvR || !count)
        return fail(SLAM_ERR_BAD_ARG, "slam_join_consensus: null buffer");
    if (reinterpret_cast<uintptr_t>(rows) & 15)
        return fail(SLAM_ERR_BAD_ARG, "slam_join_consensus: rows must be 16-byte aligned");
    JoinArgs a;
    a.pair = pair;
    a.t_idx = t_idx;
    a.t_keep = t_keep;
    a.stereo_map = stereo_map;
    a.valid3d = valid3d;
    a.X = X;
    a.Y = Y;
    a.Z = Z;
    a.xL = xL;
    a.yL = yL;
    a.xR = xR;
    a.yR = yR;
    a.cap = cap;
    a.rows = rows;
    a.cX = cX;
    a.cY = cY;
    a.cZ = cZ;
    a.cuL = cuL;
    a.cvL = cvL;
    a.cuR = cuR;
    a.cvR = cvR;
    a.count = count;
    // with no slot per pair there are no records to lay out, and the scoring call does not use
    // the workspace either (cap == 0 takes its plain path)
    a.score_work = cap > 0 ? score_work : nullptr;
    a.n_rec = (size_t)n_pairs * cap;
    a.fx = K ? K[0] : 0.0;
    a.fy = K ? K[4] : 0.0;
    a.cx = K ? K[2] : 0.0;
    a.cy = K ? K[5] : 0.0;
    a.thr = thresh;
    hipLaunchKernelGGL(join_kernel, dim3(n_pairs), dim3(JOIN_BLOCK), 0, (hipStream_t)stream, a);
    return check_launch("slam_join_consensus");
}
