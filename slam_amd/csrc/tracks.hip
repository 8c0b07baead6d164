// Winner selection and inlier-table build after hypothesis scoring (gfx950).
#include "common.h"
#include "supports.h"

namespace slam {
namespace {

// First-best argmax over H supporter counts (ransac_for_pnp keeps a candidate only
// if it is strictly better than the best so far, utils/utils.py:438), one block
// per point set, and a copy of the winning pose.
__global__ void __launch_bounds__(256) select_best_kernel(const int32_t *__restrict__ counts,
                                                          const double *__restrict__ Rt, int H,
                                                          int32_t *__restrict__ best, int32_t *__restrict__ best_count,
                                                          double *__restrict__ best_Rt)
{
    __shared__ unsigned long long red[4];
    const int set = blockIdx.x;
    const int32_t *c = counts + (size_t)set * H;
    // key orders by count descending, then index ascending
    unsigned long long key = 0ull;
    for (int h = threadIdx.x; h < H; h += 256) {
        const unsigned long long k =
            ((unsigned long long)(unsigned)max(c[h], 0) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)h);
        key = k > key ? k : key;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off);
        key = o > key ? o : key;
    }
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = key;
    __syncthreads();
    unsigned long long m = red[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        m = red[w] > m ? red[w] : m;
    const int h = (int)(0xFFFFFFFFu - (unsigned)(m & 0xFFFFFFFFull));
    if (threadIdx.x == 0) {
        best[set] = h;
        best_count[set] = (int)(m >> 32);
    }
    if (threadIdx.x < 12 && best_Rt)
        best_Rt[(size_t)set * 12 + threadIdx.x] = Rt[((size_t)set * H + h) * 12 + threadIdx.x];
}

struct TableArgs {
    const int32_t *pair;
    const int32_t *rows;
    const int32_t *count;
    int cap;
    const unsigned long long *masks;
    int mask_words;
    const float *xL, *yL, *xR;
    const int32_t *out_off;
    int32_t *inl_rows;
    float *inl_loc;
    int32_t *n_inl;
};

// Order-preserving compaction of the consensus rows of one frame pair that `take(i)` keeps,
// with the (xl, xr, yl) feature locations TrackDatabase stores per track point
// (models/TrackDatabase.py:38-46).  Called by every thread of a 256-thread block; returns the
// number of rows kept.
template <typename Take>
__device__ __forceinline__ int compact_rows(const int32_t *pr, const int32_t *rows, size_t in_base, int n,
                                            const float *xL, const float *yL, const float *xR, size_t obase,
                                            int32_t *inl_rows, float *inl_loc, int *wave_tot, Take take)
{
    const int p0 = pr[0], c0 = pr[2], cr0 = pr[5], pr0 = pr[6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const bool keep = i < n && take(i);
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0)
            wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w)
            off += wave_tot[w];
        const int total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        if (keep) {
            const int4 r = reinterpret_cast<const int4 *>(rows)[in_base + i];
            const size_t o = obase + off + before;
            reinterpret_cast<int4 *>(inl_rows)[o] = r;
            float *loc = inl_loc + 6 * o;
            loc[0] = xL[p0 + r.x];
            loc[1] = xR[pr0 + r.y];
            loc[2] = yL[p0 + r.x];
            loc[3] = xL[c0 + r.z];
            loc[4] = xR[cr0 + r.w];
            loc[5] = yL[c0 + r.z];
        }
        base += total;
        __syncthreads();
    }
    return base;
}

// Rows whose supporter bit is set in a mask computed earlier.  One block per frame pair.
__global__ void __launch_bounds__(256) inlier_table_kernel(const TableArgs a)
{
    __shared__ int wave_tot[4];
    const int pair = blockIdx.x;
    const int n = min(max(a.count[pair], 0), a.cap);
    const size_t obase = a.out_off ? (size_t)a.out_off[pair] : (size_t)pair * a.cap;
    const unsigned long long *mask = a.masks + (size_t)pair * a.mask_words;
    const int kept = compact_rows(a.pair + SLAM_PAIR_STRIDE * pair, a.rows, (size_t)pair * a.cap, n, a.xL, a.yL, a.xR,
                                  obase, a.inl_rows, a.inl_loc, wave_tot,
                                  [&](int i) { return ((mask[i >> 6] >> (i & 63)) & 1ull) != 0ull; });
    if (threadIdx.x == 0)
        a.n_inl[pair] = kept;
}

// The same tables straight from the winning pose of every pair: the supporter test of the
// reference's final find_supporters (utils/utils.py:455) is evaluated here, and the pairs are
// packed back to back (pair p starts where the supporter counts of the pairs before it end), so
// no mask, no separate scoring pass and no prefix-sum launch are needed.
struct WinnerArgs {
    const int32_t *pair;
    const int32_t *rows;
    const int32_t *count;
    int cap;
    Mat33 K;
    Vec3 tr;
    double thresh;
    const double *pose;          // [P,12]
    const int32_t *best_count;   // [P] supporters of that pose (as counted by the scoring pass)
    const double *X, *Y, *Z;
    const float *uL, *vL, *uR, *vR;
    const float *xL, *yL, *xR;
    int32_t *inl_rows;
    float *inl_loc;
    int32_t *n_inl;
};

__global__ void __launch_bounds__(256) winner_table_kernel(const WinnerArgs a)
{
    __shared__ int wave_tot[4];
    __shared__ int red[4];
    const int pair = blockIdx.x;
    // exclusive prefix of the supporter counts of the pairs before this one
    int part = 0;
    for (int q = threadIdx.x; q < pair; q += 256)
        part += max(a.best_count[q], 0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        part += __shfl_xor(part, off);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = part;
    __syncthreads();
    const size_t obase = (size_t)red[0] + red[1] + red[2] + red[3];
    double rt[12];
#pragma unroll
    for (int c = 0; c < 12; ++c)
        rt[c] = a.pose[(size_t)pair * 12 + c];
    const int n = min(max(a.count[pair], 0), a.cap);
    const size_t in_base = (size_t)pair * a.cap;
    const int kept = compact_rows(a.pair + SLAM_PAIR_STRIDE * pair, a.rows, in_base, n, a.xL, a.yL, a.xR, obase,
                                  a.inl_rows, a.inl_loc, wave_tot, [&](int i) {
                                      const size_t o = in_base + i;
                                      return supports(rt, a.K, a.tr, a.X[o], a.Y[o], a.Z[o], (double)a.uL[o],
                                                      (double)a.vL[o], (double)a.uR[o], (double)a.vR[o], a.thresh);
                                  });
    if (threadIdx.x == 0)
        a.n_inl[pair] = kept;
}

}  // namespace
}  // namespace slam

using namespace slam;

extern "C" int slam_select_best(const int32_t *counts, const double *Rt, int n_sets, int H, int32_t *best,
                                int32_t *best_count, double *best_Rt, void *stream)
{
    if (n_sets < 0 || H < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_select_best: negative size");
    if (n_sets == 0)
        return SLAM_OK;
    if (H == 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_select_best: H = 0");
    if (!counts || !best || !best_count || (best_Rt && !Rt))
        return fail(SLAM_ERR_BAD_ARG, "slam_select_best: null buffer");
    hipLaunchKernelGGL(select_best_kernel, dim3(n_sets), dim3(256), 0, (hipStream_t)stream, counts, Rt, H, best,
                       best_count, best_Rt);
    return check_launch("slam_select_best");
}

extern "C" int slam_build_inlier_tables(const int32_t *pair, int n_pairs, const int32_t *rows, const int32_t *count,
                                        int cap, const uint64_t *masks, int mask_words, const float *xL,
                                        const float *yL, const float *xR, const int32_t *out_off, int32_t *inl_rows,
                                        float *inl_loc, int32_t *n_inl, void *stream)
{
    if (n_pairs < 0 || cap < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_build_inlier_tables: negative size");
    if (n_pairs == 0)
        return SLAM_OK;
    if (!pair || !rows || !count || !masks || !xL || !yL || !xR || !inl_rows || !inl_loc || !n_inl)
        return fail(SLAM_ERR_BAD_ARG, "slam_build_inlier_tables: null buffer");
    if (mask_words < (cap + 63) / 64)
        return fail(SLAM_ERR_BAD_ARG, "slam_build_inlier_tables: mask_words %d < ceil(cap/64)", mask_words);
    if ((reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(inl_rows)) & 15)
        return fail(SLAM_ERR_BAD_ARG, "slam_build_inlier_tables: row tables must be 16-byte aligned");
    TableArgs a;
    a.pair = pair;
    a.rows = rows;
    a.count = count;
    a.cap = cap;
    a.masks = reinterpret_cast<const unsigned long long *>(masks);
    a.mask_words = mask_words;
    a.xL = xL;
    a.yL = yL;
    a.xR = xR;
    a.out_off = out_off;
    a.inl_rows = inl_rows;
    a.inl_loc = inl_loc;
    a.n_inl = n_inl;
    hipLaunchKernelGGL(inlier_table_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("slam_build_inlier_tables");
}

extern "C" int slam_winner_tables(const int32_t *pair, int n_pairs, const int32_t *rows, const int32_t *count, int cap,
                                  const double *K, const double *t_right, double thresh, const double *pose,
                                  const int32_t *best_count, const double *X, const double *Y, const double *Z,
                                  const float *uL, const float *vL, const float *uR, const float *vR, const float *xL,
                                  const float *yL, const float *xR, int32_t *inl_rows, float *inl_loc, int32_t *n_inl,
                                  void *stream)
{
    if (n_pairs < 0 || cap < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_winner_tables: negative size");
    if (n_pairs == 0)
        return SLAM_OK;
    if (!pair || !rows || !count || !K || !t_right || !pose || !best_count || !xL || !yL || !xR || !inl_rows ||
        !inl_loc || !n_inl)
        return fail(SLAM_ERR_BAD_ARG, "slam_winner_tables: null buffer");
    if (cap > 0 && (!X || !Y || !Z || !uL || !vL || !uR || !vR))
        return fail(SLAM_ERR_BAD_ARG, "slam_winner_tables: null point buffer");
    if ((reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(inl_rows)) & 15)
        return fail(SLAM_ERR_BAD_ARG, "slam_winner_tables: row tables must be 16-byte aligned");
    WinnerArgs a;
    a.pair = pair;
    a.rows = rows;
    a.count = count;
    a.cap = cap;
    for (int i = 0; i < 9; ++i)
        a.K.m[i] = K[i];
    for (int i = 0; i < 3; ++i)
        a.tr.v[i] = t_right[i];
    a.thresh = thresh;
    a.pose = pose;
    a.best_count = best_count;
    a.X = X;
    a.Y = Y;
    a.Z = Z;
    a.uL = uL;
    a.vL = vL;
    a.uR = uR;
    a.vR = vR;
    a.xL = xL;
    a.yL = yL;
    a.xR = xR;
    a.inl_rows = inl_rows;
    a.inl_loc = inl_loc;
    a.n_inl = n_inl;
    hipLaunchKernelGGL(winner_table_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("slam_winner_tables");
}
