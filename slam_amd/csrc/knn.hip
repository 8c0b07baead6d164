// Batched brute-force Hamming 2-NN with fused ratio / row filters (gfx950).
//
// Work decomposition
//   block  = 4 wavefronts = one tile of 64*Q query rows of one batch item
//   lane   = Q query descriptors held in VGPRs for the whole scan
//   wave w = scans rows [64w, 64w+64) of every 256-row train tile; the train
//            tile sits in LDS and each row is read once per wave as a
//            wave-uniform (broadcast) ds_read_b128 and reused for the lane's Q
//            queries
//   inner  = per 32-bit word one v_xor_b32 + one v_bcnt_u32_b32 (popcount with
//            accumulate); per (query, train) pair three more VALU ops keep the
//            running top-2: key = dist << 22 | train_index, b2 = med3(b1,b2,key),
//            b1 = min(b1,key).  Keys are unique per train row, so integer order
//            on keys is exactly cv2's (distance, then lower index) order and the
//            merge across waves is order independent.
//   merge  = the four waves' (b1,b2) per query go through LDS once at the end.
#include <stdlib.h>

#include "knn_common.h"

namespace slam {

namespace {

template <int W, int Q>
__global__ void __launch_bounds__(BLOCK) hamming_knn_kernel(const KnnArgs a)
{
    constexpr int C = W / 4;  // 16-byte chunks per descriptor row
    __shared__ uint4 lds[2][TT * C];

    const int item = blockIdx.x / a.tiles_per_item;
    const int tile = blockIdx.x - item * a.tiles_per_item;
    const int32_t *seg = a.seg + SLAM_SEG_STRIDE * item;
    const int qs = seg[0], qn = seg[1], ts = seg[2], tn = seg[3], os = seg[4];
    const int q0 = tile * (WAVE * Q);
    if (q0 >= qn)
        return;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- queries into registers -------------------------------------------
    uint32_t qv[Q][W];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const int row = q0 + lane + WAVE * i;
        const bool ok = row < qn;
        const uint4 *src = a.q + (size_t)(qs + (ok ? row : 0)) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            uint4 v = ok ? src[c] : make_uint4(0, 0, 0, 0);
            qv[i][4 * c + 0] = v.x & word_mask(4 * c + 0, a.desc_bytes);
            qv[i][4 * c + 1] = v.y & word_mask(4 * c + 1, a.desc_bytes);
            qv[i][4 * c + 2] = v.z & word_mask(4 * c + 2, a.desc_bytes);
            qv[i][4 * c + 3] = v.w & word_mask(4 * c + 3, a.desc_bytes);
        }
    }
    uint32_t b1[Q], b2[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i)
        b1[i] = b2[i] = KEY_EMPTY;

    // ---- train tiles through LDS, double buffered ---------------------------
    // Thread `tid` stages chunks tid + 256*i (i < C): always the same 16-byte
    // part of a row, so its four word masks are fixed.
    const int part = tid % C;
    const uint4 pmask = make_uint4(word_mask(4 * part + 0, a.desc_bytes), word_mask(4 * part + 1, a.desc_bytes),
                                   word_mask(4 * part + 2, a.desc_bytes), word_mask(4 * part + 3, a.desc_bytes));
    const int ntiles = (tn + TT - 1) / TT;
    uint4 stage[C];
    auto fetch = [&](int tt) {
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int c = tid + BLOCK * i;
            const int row = tt * TT + c / C;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < tn)
                v = a.t[(size_t)(ts + row) * C + part];
            v.x &= pmask.x;
            v.y &= pmask.y;
            v.z &= pmask.z;
            v.w &= pmask.w;
            stage[i] = v;
        }
    };
    if (ntiles > 0)
        fetch(0);
    for (int tt = 0; tt < ntiles; ++tt) {
        uint4 *buf = lds[tt & 1];
#pragma unroll
        for (int i = 0; i < C; ++i)
            buf[tid + BLOCK * i] = stage[i];
        __syncthreads();
        if (tt + 1 < ntiles)
            fetch(tt + 1);

        const int base = tt * TT + wave * WAVE;
        const int cnt = min(WAVE, tn - base);
        const uint4 *rows = buf + wave * WAVE * C;
        for (int jj = 0; jj < cnt; ++jj) {
            uint32_t tw[W];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const uint4 v = rows[jj * C + c];
                tw[4 * c + 0] = v.x;
                tw[4 * c + 1] = v.y;
                tw[4 * c + 2] = v.z;
                tw[4 * c + 3] = v.w;
            }
            const uint32_t j = (uint32_t)(base + jj);
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                uint32_t d = 0;
#pragma unroll
                for (int w = 0; w < W; ++w)
                    d = __builtin_popcount(qv[i][w] ^ tw[w]) + d;
                insert_key(b1[i], b2[i], (d << KEY_SHIFT) | j);
            }
        }
    }

    // ---- merge the four waves' partial top-2 --------------------------------
    __syncthreads();
    uint2 *mk = reinterpret_cast<uint2 *>(&lds[0][0]);  // [4][Q][64]
#pragma unroll
    for (int i = 0; i < Q; ++i)
        mk[(wave * Q + i) * WAVE + lane] = make_uint2(b1[i], b2[i]);
    __syncthreads();
    if (tid >= WAVE * Q)
        return;
    const int qi = tid >> 6;
    const int row = q0 + lane + WAVE * qi;
    if (row >= qn)
        return;
    uint32_t k1 = KEY_EMPTY, k2 = KEY_EMPTY;
#pragma unroll
    for (int w = 0; w < BLOCK / WAVE; ++w) {
        const uint2 p = mk[(w * Q + qi) * WAVE + lane];
        insert_key(k1, k2, p.x);
        insert_key(k1, k2, p.y);
    }

    knn_store(a, qs, ts, os, row, k1, k2);
}

__global__ void __launch_bounds__(256) pack_kernel(const uint8_t *src, int n, int desc_bytes, int src_stride,
                                                   uint32_t *dst, int dst_words)
{
    const size_t total = (size_t)n * dst_words;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / dst_words;
        const int w = (int)(e - row * dst_words);
        const uint8_t *p = src + row * (size_t)src_stride + 4 * w;
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (4 * w + b < desc_bytes)
                v |= (uint32_t)p[b] << (8 * b);
        dst[e] = v;
    }
}

// Stand-alone forms of the two match filters, for callers that hold the match lists.
__global__ void __launch_bounds__(256) ratio_filter_kernel(const int32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ dist, int n, double thr,
                                                           uint8_t *__restrict__ keep)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        keep[i] = (idx[2 * i] >= 0) && (idx[2 * i + 1] >= 0) && ((double)dist[2 * i] < thr * (double)dist[2 * i + 1]);
}

__global__ void __launch_bounds__(256) row_filter_kernel(const float *__restrict__ y1, const float *__restrict__ y2,
                                                         int n, double thr, uint8_t *__restrict__ keep)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        keep[i] = fabs((double)y1[i] - (double)y2[i]) <= thr;
}

template <int W, int Q>
int launch_knn(const KnnArgs &a, int batch, int max_nq, hipStream_t s)
{
    KnnArgs args = a;
    args.tiles_per_item = (max_nq + WAVE * Q - 1) / (WAVE * Q);
    const long long blocks = (long long)batch * args.tiles_per_item;
    if (blocks > 0x7FFFFFFFLL)
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: grid too large (%lld blocks)", blocks);
    hipLaunchKernelGGL((hamming_knn_kernel<W, Q>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, args);
    return check_launch("slam_hamming_match");
}

template <int W>
int dispatch_q(const KnnArgs &a, int batch, int max_nq, hipStream_t s)
{
    // Fewer queries per lane when the batch is too small to fill 256 CUs.
    auto blocks = [&](int q) { return (long long)batch * ((max_nq + WAVE * q - 1) / (WAVE * q)); };
    if (blocks(4) >= 1024)
        return launch_knn<W, 4>(a, batch, max_nq, s);
    if (blocks(2) >= 1024)
        return launch_knn<W, 2>(a, batch, max_nq, s);
    return launch_knn<W, 1>(a, batch, max_nq, s);
}

}  // namespace

}  // namespace slam

using namespace slam;

extern "C" int slam_pack_descriptors(const uint8_t *src, int n, int desc_bytes, int src_stride, uint8_t *dst,
                                     int dst_stride, void *stream)
{
    if (n < 0 || desc_bytes <= 0 || src_stride < desc_bytes || (dst_stride != 32 && dst_stride != 64) ||
        desc_bytes > dst_stride)
        return fail(SLAM_ERR_BAD_ARG,
                    "slam_pack_descriptors: need 0 < desc_bytes <= src_stride, dst_stride in {32,64} >= desc_bytes "
                    "(got n=%d desc_bytes=%d src_stride=%d dst_stride=%d)",
                    n, desc_bytes, src_stride, dst_stride);
    if (n == 0)
        return SLAM_OK;
    if (!src || !dst)
        return fail(SLAM_ERR_BAD_ARG, "slam_pack_descriptors: null buffer");
    const int words = dst_stride / 4;
    const size_t total = (size_t)n * words;
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, n, desc_bytes, src_stride,
                       reinterpret_cast<uint32_t *>(dst), words);
    return check_launch("slam_pack_descriptors");
}

extern "C" int slam_hamming_match(const uint8_t *q, const uint8_t *t, const int32_t *seg, int batch, int max_nq,
                                  int max_nt, int desc_bytes, int stride_bytes, int k, int mode, double thr,
                                  const float *yq, const float *yt, int32_t *out_idx, int32_t *out_dist,
                                  uint8_t *keep, void *stream)
{
    if (batch < 0 || max_nq < 0 || max_nt < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: negative size");
    if (stride_bytes != 32 && stride_bytes != 64)
        return fail(SLAM_ERR_UNSUPPORTED,
                    "slam_hamming_match: stride_bytes must be 32 or 64 (got %d); repack with slam_pack_descriptors",
                    stride_bytes);
    if (desc_bytes <= 0 || desc_bytes > stride_bytes)
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: desc_bytes %d outside (0, %d]", desc_bytes, stride_bytes);
    if (k != 1 && k != 2)
        return fail(SLAM_ERR_UNSUPPORTED, "slam_hamming_match: k must be 1 or 2 (got %d)", k);
    if (max_nt > SLAM_MAX_TRAIN_ROWS)
        return fail(SLAM_ERR_UNSUPPORTED, "slam_hamming_match: max_nt %d exceeds %d", max_nt, SLAM_MAX_TRAIN_ROWS);
    if (mode < 0 || mode > 2)
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: mode must be 0, 1 or 2 (got %d)", mode);
    if (mode == 1 && k != 2)
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: the ratio test (mode 1) needs k = 2");
    if (batch == 0 || max_nq == 0)
        return SLAM_OK;
    if (!q || !seg || !out_idx || !out_dist || (max_nt > 0 && !t))
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: null buffer");
    if (mode != 0 && !keep)
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: mode %d needs a keep buffer", mode);
    if (mode == 2 && (!yq || !yt))
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: the row filter (mode 2) needs yq and yt");
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(t)) & 15)
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: descriptor tables must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(out_idx) | reinterpret_cast<uintptr_t>(out_dist)) & 7)
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: outputs must be 8-byte aligned");

    KnnArgs a;
    a.q = reinterpret_cast<const uint4 *>(q);
    a.t = reinterpret_cast<const uint4 *>(t);
    a.seg = seg;
    a.tiles_per_item = 0;
    a.desc_bytes = desc_bytes;
    a.k = k;
    a.mode = mode;
    a.thr = thr;
    a.yq = yq;
    a.yt = yt;
    a.out_idx = out_idx;
    a.out_dist = out_dist;
    a.keep = keep;
    hipStream_t s = (hipStream_t)stream;
    // Two implementations with identical results: the fp4 matrix-core kernel (default) and the VALU
    // popcount kernel (slam_select_kernels / SLAM_KNN_IMPL=valu), kept as the in-library cross-check.
    if (knn_impl() != SLAM_KNN_VALU)
        return launch_knn_fp4(a, batch, max_nq, stride_bytes, s);
    if (stride_bytes == 64)
        return dispatch_q<16>(a, batch, max_nq, s);
    return dispatch_q<8>(a, batch, max_nq, s);
}

extern "C" int slam_hamming_knn2(const uint8_t *q, const uint8_t *t, const int32_t *seg, int batch, int max_nq,
                                 int max_nt, int desc_bytes, int stride_bytes, int k, int32_t *out_idx,
                                 int32_t *out_dist, void *stream)
{
    return slam_hamming_match(q, t, seg, batch, max_nq, max_nt, desc_bytes, stride_bytes, k, 0, 0.0, nullptr, nullptr,
                              out_idx, out_dist, nullptr, stream);
}

extern "C" int slam_ratio_filter(const int32_t *idx, const int32_t *dist, int n, double thr, uint8_t *keep,
                                 void *stream)
{
    if (n < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_ratio_filter: n = %d", n);
    if (n == 0)
        return SLAM_OK;
    if (!idx || !dist || !keep)
        return fail(SLAM_ERR_BAD_ARG, "slam_ratio_filter: null buffer");
    hipLaunchKernelGGL(ratio_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, dist, n, thr,
                       keep);
    return check_launch("slam_ratio_filter");
}

extern "C" int slam_row_filter(const float *y1, const float *y2, int n, double thr, uint8_t *keep, void *stream)
{
    if (n < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_row_filter: n = %d", n);
    if (n == 0)
        return SLAM_OK;
    if (!y1 || !y2 || !keep)
        return fail(SLAM_ERR_BAD_ARG, "slam_row_filter: null buffer");
    hipLaunchKernelGGL(row_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, y1, y2, n, thr,
                       keep);
    return check_launch("slam_row_filter");
}
