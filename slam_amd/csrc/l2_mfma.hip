// Batched L2 2-NN for integer-valued descriptors on the int8 matrix cores of gfx950.
//
// SIFT descriptors as OpenCV emits them (the reference's default detector, models/Matcher.py:15)
// are float32 arrays holding integers 0..255.  For such data the squared distance is an exact
// integer, and with a' = a - 128 in [-128, 127]
//
//     |a - b|^2 = |a'|^2 + |b'|^2 - 2 a'.b'
//
// the cross term is an int8 dot product: D/32 v_mfma_i32_32x32x32_i8 per tile of 32 train rows x
// 32 queries.  Everything is integer arithmetic, so the distances equal cv2's float32
// accumulation bit for bit (all its partial sums are exactly representable integers) and the
// reported distance is sqrtf of it (cv2's NORM_L2).  Tie rule as cv2's scan: equal distances keep
// the lower train index first.
//
// Mapping (as the Hamming matcher's): block = 4 waves, a wave keeps QT tiles of 32 queries as B
// operands in VGPRs for the whole scan; train rows go through LDS 64 at a time (double
// buffered, row stride D + 16 so that the ds_read_b128 A fragments are conflict free), together
// with one int per row c_j = -256 |b'_j|^2 + 255 - (j mod 256).  The value that is maximised is
//
//     key = (a'.b' << 9) + c_j = 256 (2 a'.b' - |b'_j|^2) + (255 - j mod 256)
//
// i.e. smallest distance first, lowest index first inside a 256-row chunk: one
// v_lshl_add_u32 per accumulator, then 1.25 instructions per accumulator for the running top two
// (v_max3_i32 and one v_med3_i32 per two keys).  Every 256
// rows the two winners are folded into (distance, index) pairs.  |key| < 1.62e9 fits an int.
#include "common.h"

namespace slam {
namespace {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

constexpr int L2_TS = 64;          // train rows per super-tile
constexpr int L2_CHUNK = 256;      // train rows per key chunk
constexpr int L2_NONE = INT32_MIN + 1;                          // c_j of a row past the end (its A row is zero)
__device__ __forceinline__ bool l2_is_candidate(int key) { return key > -1700000000; }

struct L2iArgs {
    const int8_t *q;
    const int32_t *qn2;
    const int8_t *t;
    const int32_t *tn2;
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

__device__ __forceinline__ int med3i(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// (distance, index) pairs in lexicographic order; index -1 = empty
struct Best2 {
    int d1, i1, d2, i2;
};
__device__ __forceinline__ bool l2_less(int da, int ia, int db, int ib) { return da < db || (da == db && ia < ib); }
__device__ __forceinline__ void l2_insert(Best2 &b, int d, int i)
{
    if (i < 0)
        return;
    if (b.i1 < 0 || l2_less(d, i, b.d1, b.i1)) {
        b.d2 = b.d1;
        b.i2 = b.i1;
        b.d1 = d;
        b.i1 = i;
    } else if (b.i2 < 0 || l2_less(d, i, b.d2, b.i2)) {
        b.d2 = d;
        b.i2 = i;
    }
}

// float32 [n, D] -> int8 [n, D] (value - 128), |row|^2 of the shifted row, and a count of the
// entries that are not integers in 0..255 (such a table must use the float kernel).
template <int D>
__global__ void __launch_bounds__(256) l2_pack_kernel(const float *src, int n, int8_t *dst, int32_t *norm2, int32_t *bad)
{
    constexpr int CPR = D / 16;                   // 16-entry chunks per row (a power of two <= 8)
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = id / CPR;
    const int c = (int)(id - row * CPR);
    int sum = 0, nbad = 0;
    if (row < n) {
        const float4 *s = reinterpret_cast<const float4 *>(src + (size_t)row * D + 16 * c);
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float4 v = s[e];
            const float f[4] = {v.x, v.y, v.z, v.w};
            w[e] = 0u;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const bool ok = f[b] >= 0.0f && f[b] <= 255.0f && f[b] == truncf(f[b]);
                nbad += ok ? 0 : 1;
                const int iv = ok ? (int)f[b] - 128 : 0;
                sum += iv * iv;
                w[e] |= ((uint32_t)iv & 0xFFu) << (8 * b);
            }
        }
        *reinterpret_cast<uint4 *>(dst + (size_t)row * D + 16 * c) = make_uint4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int off = CPR / 2; off > 0; off >>= 1)
        sum += __shfl_xor(sum, off);
    if (row < n && c == 0)
        norm2[row] = sum;
    if (nbad)
        atomicAdd(bad, nbad);
}

// KS = k-steps of 32 dimensions (4 for 128, 2 for 64), QT = query tiles per wave
template <int KS, int QT>
__global__ void __launch_bounds__(256, 2) l2_mfma_kernel(const L2iArgs a)
{
    constexpr int D = KS * 32;
    constexpr int ROW = D + 16;        // bytes between staged rows in LDS
    constexpr int C = D / 16;          // 16-byte chunks per row
    constexpr int CPT = C / 4;         // chunks a thread stages per super-tile (256 threads, 64 rows)
    __shared__ __attribute__((aligned(16))) uint8_t lds[2][L2_TS * ROW];
    __shared__ __attribute__((aligned(16))) int cj[2][L2_TS];

    // blocks are dealt to the 8 XCDs round-robin; renumber so that the tiles of one item share an L2
    const int nb = gridDim.x, xq = nb >> 3, xr = nb & 7;
    const int xcd = blockIdx.x & 7, nth = blockIdx.x >> 3;
    const int bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + nth;
    const int item = bid / a.tiles_per_item;
    const int tile = bid - item * a.tiles_per_item;
    const int32_t *seg = a.seg + SLAM_SEG_STRIDE * item;
    const int qs = seg[0], qn = seg[1], ts = seg[2], tn = seg[3], os = seg[4];
    const int q0 = tile * (4 * QT * 32);
    if (q0 >= qn)
        return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;   // query within a tile / train row within an A fragment
    const int half = lane >> 5;  // which 16 of the 32 k-slots of a step this lane feeds

    // ---- B operands: this wave's queries ----------------------------------------------------
    i32x4 bq[QT][KS];
    int nq2[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int row = q0 + (wave * QT + t) * 32 + col;
        const int srow = qs + (row < qn ? row : 0);
        const int8_t *src = a.q + (size_t)srow * D + 16 * half;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            bq[t][ks] = *reinterpret_cast<const i32x4 *>(src + 32 * ks);
        nq2[t] = a.qn2[srow];
    }
    Best2 best[QT];
    int b1[QT], b2[QT];          // two largest keys of the current chunk
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        best[t].d1 = best[t].d2 = 0;
        best[t].i1 = best[t].i2 = -1;
        b1[t] = b2[t] = L2_NONE;
    }

    // staging: thread -> (row = tid % 64, chunks tid / 64 + 4 e)
    const int srow = tid & 63;
    const int spart = tid >> 6;
    const int n_st = (tn + L2_TS - 1) / L2_TS;
    auto load_raw = [&](int st, uint4 (&v)[CPT], int &c) __attribute__((always_inline)) {
        const int row = st * L2_TS + srow;
        const bool in = st < n_st && row < tn;
#pragma unroll
        for (int e = 0; e < CPT; ++e)
            v[e] = in ? *reinterpret_cast<const uint4 *>(a.t + (size_t)(ts + row) * D + 16 * (spart + 4 * e))
                      : make_uint4(0, 0, 0, 0);
        c = L2_NONE;
        if (in && spart == 0)
            c = -256 * a.tn2[ts + row] + 255 - (row & (L2_CHUNK - 1));
    };
    auto store_raw = [&](int buf, const uint4 (&v)[CPT], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < CPT; ++e)
            *reinterpret_cast<uint4 *>(lds[buf] + srow * ROW + 16 * (spart + 4 * e)) = v[e];
        if (spart == 0)
            cj[buf][srow] = c;
    };

    uint4 raw[CPT];
    int rawc;
    load_raw(0, raw, rawc);
    store_raw(0, raw, rawc);
    load_raw(1, raw, rawc);
    __syncthreads();

    const i32x16 zero = {};
    // one tile of 32 train rows against the wave's QT x 32 queries
    auto tile32 = [&](const uint8_t *abase, const int *cbase) __attribute__((always_inline)) {
        const uint8_t *arow = abase + col * ROW + 16 * half;
        i32x16 acc[QT];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const i32x4 af = *reinterpret_cast<const i32x4 *>(arow + 32 * ks);
#pragma unroll
            for (int t = 0; t < QT; ++t)
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bq[t][ks], ks == 0 ? zero : acc[t], 0, 0, 0);
        }
        // accumulator register r of this lane is train row (r & 3) + 8 (r >> 2) + 4 half of the tile
        i32x4 cc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            cc[g] = *reinterpret_cast<const i32x4 *>(cbase + 8 * g + 4 * half);
        // running top two, two keys x, y at a time: with b1 >= b2 the second largest of
        // {b1, b2, x, y} is max(b2, med3(b1, x, y)); the medians are folded into b2 pairwise
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            int m[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = 2 * i;
                const int x = (int)(((uint32_t)acc[t][r] << 9) + (uint32_t)cc[r >> 2][r & 3]);
                const int y = (int)(((uint32_t)acc[t][r + 1] << 9) + (uint32_t)cc[(r + 1) >> 2][(r + 1) & 3]);
                m[i] = med3i(b1[t], x, y);
                b1[t] = max(max(b1[t], x), y);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                b2[t] = max(max(b2[t], m[2 * i]), m[2 * i + 1]);
        }
    };

    for (int st = 0; st < n_st; ++st) {
        const int cur = st & 1;
        const int row0 = st * L2_TS;
        tile32(lds[cur], cj[cur]);
        if (st + 1 < n_st)
            store_raw(cur ^ 1, raw, rawc);          // the buffer's last readers passed the previous barrier
        if (st + 2 < n_st)
            load_raw(st + 2, raw, rawc);
        if (row0 + 32 < tn)
            tile32(lds[cur] + 32 * ROW, cj[cur] + 32);

        if (((st + 1) * L2_TS) % L2_CHUNK == 0 || st + 1 == n_st) {
            // fold the chunk's winners into (distance, index) pairs; b1 >= b2, so b1 goes first
            const int chunk0 = row0 & ~(L2_CHUNK - 1);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const int v[2] = {b1[t], b2[t]};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (l2_is_candidate(v[e])) {
                        const int s = v[e] >> 8;                     // 2 a'.b' - |b'|^2 (arithmetic shift floors)
                        const int j = 255 - (v[e] & 255);
                        l2_insert(best[t], nq2[t] - s, chunk0 + j);
                    }
                }
                b1[t] = b2[t] = L2_NONE;
            }
        }
        __syncthreads();
    }

    // ---- merge the two lane halves of each query, filter, store -------------------------
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int od1 = __shfl_xor(best[t].d1, 32), oi1 = __shfl_xor(best[t].i1, 32);
        const int od2 = __shfl_xor(best[t].d2, 32), oi2 = __shfl_xor(best[t].i2, 32);
        l2_insert(best[t], od1, oi1);
        l2_insert(best[t], od2, oi2);
        const int row = q0 + (wave * QT + t) * 32 + col;
        if (half != 0 || row >= qn)
            continue;
        const int i1 = best[t].i1;
        int i2 = best[t].i2;
        if (a.k < 2)
            i2 = -1;
        const float d0 = i1 >= 0 ? sqrtf((float)best[t].d1) : 3.402823466e+38f;
        const float d1 = i2 >= 0 ? sqrtf((float)best[t].d2) : 3.402823466e+38f;
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
}

template <int KS, int QT>
int launch(const L2iArgs &a, int batch, int max_nq, hipStream_t s)
{
    L2iArgs args = a;
    args.tiles_per_item = (max_nq + 128 * QT - 1) / (128 * QT);
    const long long blocks = (long long)batch * args.tiles_per_item;
    if (blocks > 0x7FFFFFFFLL)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match_int8: grid too large (%lld blocks)", blocks);
    hipLaunchKernelGGL((l2_mfma_kernel<KS, QT>), dim3((unsigned)blocks), dim3(256), 0, s, args);
    return check_launch("slam_l2_match_int8");
}

}  // namespace
}  // namespace slam

using namespace slam;

extern "C" int slam_l2_pack_int8(const float *desc, int n, int dim, int8_t *out, int32_t *norm2, int32_t *bad,
                                 void *stream)
{
    if (n < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_pack_int8: negative size");
    if (dim != 128 && dim != 64)
        return fail(SLAM_ERR_UNSUPPORTED, "slam_l2_pack_int8: dim must be 128 or 64 (got %d)", dim);
    if (n == 0)
        return SLAM_OK;
    if (!desc || !out || !norm2 || !bad)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_pack_int8: null buffer");
    if ((reinterpret_cast<uintptr_t>(desc) | reinterpret_cast<uintptr_t>(out)) & 15)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_pack_int8: tables must be 16-byte aligned");
    const long long threads = (long long)n * (dim / 16);
    const long long blocks = (threads + 255) / 256;
    if (blocks > 0x7FFFFFFFLL)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_pack_int8: grid too large (%lld blocks)", blocks);
    if (dim == 128)
        hipLaunchKernelGGL(l2_pack_kernel<128>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, desc, n, out,
                           norm2, bad);
    else
        hipLaunchKernelGGL(l2_pack_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, desc, n, out,
                           norm2, bad);
    return check_launch("slam_l2_pack_int8");
}

extern "C" int slam_l2_match_int8(const int8_t *q, const int32_t *q_norm2, const int8_t *t, const int32_t *t_norm2,
                                  const int32_t *seg, int batch, int max_nq, int max_nt, int dim, int k, int mode,
                                  double thr, const float *yq, const float *yt, int32_t *out_idx, float *out_dist,
                                  uint8_t *keep, void *stream)
{
    if (batch < 0 || max_nq < 0 || max_nt < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match_int8: negative size");
    if (dim != 128 && dim != 64)
        return fail(SLAM_ERR_UNSUPPORTED, "slam_l2_match_int8: dim must be 128 or 64 (got %d)", dim);
    if (k != 1 && k != 2)
        return fail(SLAM_ERR_UNSUPPORTED, "slam_l2_match_int8: k must be 1 or 2 (got %d)", k);
    if (mode < 0 || mode > 2)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match_int8: mode must be 0, 1 or 2 (got %d)", mode);
    if (mode == 1 && k != 2)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match_int8: the ratio test (mode 1) needs k = 2");
    if (batch == 0 || max_nq == 0)
        return SLAM_OK;
    if (!q || !q_norm2 || !seg || !out_idx || !out_dist || (max_nt > 0 && (!t || !t_norm2)))
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match_int8: null buffer");
    if (mode != 0 && !keep)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match_int8: mode %d needs a keep buffer", mode);
    if (mode == 2 && (!yq || !yt))
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match_int8: the row filter (mode 2) needs yq and yt");
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(t)) & 15)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match_int8: descriptor tables must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(out_idx) | reinterpret_cast<uintptr_t>(out_dist)) & 7)
        return fail(SLAM_ERR_BAD_ARG, "slam_l2_match_int8: outputs must be 8-byte aligned");
    L2iArgs a;
    a.q = q;
    a.qn2 = q_norm2;
    a.t = t;
    a.tn2 = t_norm2;
    a.seg = seg;
    a.tiles_per_item = 0;
    a.k = k;
    a.mode = mode;
    a.thr = thr;
    a.yq = yq;
    a.yt = yt;
    a.out_idx = out_idx;
    a.out_dist = out_dist;
    a.keep = keep;
    // four query tiles per wave when that still gives the chip enough blocks, else one
    const long long wide_blocks = (long long)batch * ((max_nq + 511) / 512);
    hipStream_t s = (hipStream_t)stream;
    if (dim == 128)
        return wide_blocks >= 512 ? launch<4, 4>(a, batch, max_nq, s) : launch<4, 1>(a, batch, max_nq, s);
    return wide_blocks >= 512 ? launch<2, 4>(a, batch, max_nq, s) : launch<2, 1>(a, batch, max_nq, s);
}
