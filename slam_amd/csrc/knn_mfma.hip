// Batched Hamming 2-NN on the matrix cores (gfx950, v_mfma_i32_32x32x32_i8).
//
// Hamming distance as a dot product: expand every descriptor bit to an int8 value,
// +64 for a 0 bit and -64 for a 1 bit (0 for query bits past desc_bytes).  Then for a
// query q and a train row t
//        sum_k q_k * t_k = 4096 * (Dbits - 2 * ham(q, t)),
// exactly, in the int32 accumulator.  Seeding the accumulator of train row j with
// -(4096*Dbits + j) makes the MFMA result  -(8192*ham + j) = -key: the larger the
// accumulator, the better the match, ties towards the lower train index -- cv2's order --
// so the running top-2 per query is one v_max_i32 + one v_med3_i32 per result.
//
// Work decomposition
//   block  = 4 wavefronts; wave w owns QT tiles of 32 queries whose expanded descriptors
//            stay in VGPRs as MFMA B operands for the whole scan
//   trains = super-tiles of 64 rows: the block expands them once (bits -> int8, 4 VALU
//            ops per 4 bits) into LDS, rows 528 B apart so that the A-operand
//            ds_read_b128 of 32 rows is conflict free; every wave then runs
//            2 x QT x (stride_bytes/4) MFMAs on the super-tile
//   top-2  = per lane over its 16 accumulators (16 train rows x 1 query), the two lane
//            halves of a query are merged with one shuffle at the end
//   j      = 13 bits per chunk of 8192 train rows; longer train sets run chunk by chunk
//            and fold into the (distance << 22 | index) keys of the epilogue.
#include "knn_common.h"

namespace slam {

namespace {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

constexpr int TS = 64;            // train rows per super-tile
constexpr int CHUNK = 8192;       // train rows per key chunk (13 index bits)
constexpr int JBITS = 13;

// 4 bits -> 4 int8 lanes: bit b of n lands in byte b as +64 (0) / -64 (1).
__device__ __forceinline__ uint32_t expand4(uint32_t n)
{
    const uint32_t spread = __umul24(n, 0x00204081u) & 0x01010101u;  // byte b = bit b
    return (spread << 7) | 0x40404040u;
}

// 16 bits -> 16 int8 lanes (one MFMA operand register quad).
__device__ __forceinline__ i32x4 expand16(uint32_t h)
{
    i32x4 v;
    v.x = (int)expand4(h & 15u);
    v.y = (int)expand4((h >> 4) & 15u);
    v.z = (int)expand4((h >> 8) & 15u);
    v.w = (int)expand4((h >> 12) & 15u);
    return v;
}

__device__ __forceinline__ int med3i(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Byte mask that zeroes the int8 lanes of expanded bits at positions >= valid (0..4).
__device__ __forceinline__ uint32_t lane_mask(int valid)
{
    return valid >= 4 ? 0xFFFFFFFFu : (valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u));
}

template <int KS, int QT>   // KS = k-steps of 32 bits (8 for 32-byte rows, 16 for 64-byte), QT = query tiles per wave
__global__ void __launch_bounds__(BLOCK, 2) hamming_mfma_kernel(const KnnArgs a)
{
    constexpr int W = KS;               // 32-bit words per packed descriptor row
    constexpr int C = W / 4;            // 16-byte chunks per packed row
    constexpr int ROW = KS * 32 + 16;   // bytes between expanded rows in LDS
    __shared__ __attribute__((aligned(16))) uint8_t lds[TS * ROW];

    const int item = blockIdx.x / a.tiles_per_item;
    const int tile = blockIdx.x - item * a.tiles_per_item;
    const int32_t *seg = a.seg + SLAM_SEG_STRIDE * item;
    const int qs = seg[0], qn = seg[1], ts = seg[2], tn = seg[3], os = seg[4];
    const int q0 = tile * (4 * QT * 32);
    if (q0 >= qn)
        return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;   // query within a tile / train row within an A fragment
    const int half = lane >> 5;  // which 16 of the 32 k-values of a step this lane feeds
    const int dbits = 8 * a.desc_bytes;

    // ---- B operands: this wave's queries, expanded, pad bits zeroed --------------------
    i32x4 bq[QT][KS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int row = q0 + (wave * QT + t) * 32 + col;
        const bool ok = row < qn;
        const uint4 *src = a.q + (size_t)(qs + (ok ? row : 0)) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const uint4 v = ok ? src[c] : make_uint4(0, 0, 0, 0);
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ks = 4 * c + e;
                i32x4 x = expand16((w4[e] >> (16 * half)) & 0xFFFFu);
                const int first = 32 * ks + 16 * half;   // bit position of this lane's first k-value
                x.x &= (int)lane_mask(dbits - first);
                x.y &= (int)lane_mask(dbits - first - 4);
                x.z &= (int)lane_mask(dbits - first - 8);
                x.w &= (int)lane_mask(dbits - first - 12);
                if (!ok)
                    x = i32x4{0, 0, 0, 0};
                bq[t][ks] = x;
            }
        }
    }

    // accumulator seed of the 16 train rows this lane sees in a tile, for tile base 0 of a chunk:
    // -(4096*Dbits) - row, row = (reg & 3) + 8 * (reg >> 2) + 4 * half
    i32x16 seed;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        seed[r] = -(4096 * dbits) - ((r & 3) + 8 * (r >> 2) + 4 * half);

    uint32_t g1[QT], g2[QT];   // running (distance << 22 | index) keys over finished chunks
#pragma unroll
    for (int t = 0; t < QT; ++t)
        g1[t] = g2[t] = KEY_EMPTY;

    // staging assignment: thread -> (row = tid % 64, 16-byte part = tid / 64 [+4 for 64-byte rows])
    const int srow = tid & 63;
    const int spart = tid >> 6;

    for (int chunk0 = 0; chunk0 < tn; chunk0 += CHUNK) {
        const int cn = min(CHUNK, tn - chunk0);
        int b1[QT], b2[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t)
            b1[t] = b2[t] = INT32_MIN;
        i32x16 cinit = seed;

        for (int st = 0; st < cn; st += TS) {
            // ---- expand one super-tile of train rows into LDS ---------------------------
            __syncthreads();   // previous super-tile fully consumed
#pragma unroll
            for (int pp = 0; pp < C; pp += 4) {
                const int part = spart + pp;
                if (part < C) {
                    const int row = st + srow;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (row < cn)
                        v = a.t[(size_t)(ts + chunk0 + row) * C + part];
                    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
                    uint8_t *dst = lds + srow * ROW + part * 128;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        *reinterpret_cast<i32x4 *>(dst + 32 * e) = expand16(w4[e] & 0xFFFFu);
                        *reinterpret_cast<i32x4 *>(dst + 32 * e + 16) = expand16(w4[e] >> 16);
                    }
                }
            }
            __syncthreads();

            // ---- two 32-row tiles x QT query tiles on the matrix cores -------------------
#pragma unroll
            for (int tt = 0; tt < TS / 32; ++tt) {
                const int tbase = st + 32 * tt;   // uniform
                if (tbase < cn) {
                    i32x16 acc[QT];
                    const uint8_t *arow = lds + (32 * tt + col) * ROW + 16 * half;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const i32x4 af = *reinterpret_cast<const i32x4 *>(arow + 32 * ks);
#pragma unroll
                        for (int t = 0; t < QT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bq[t][ks], ks == 0 ? cinit : acc[t], 0, 0,
                                                                           0);
                    }
                    if (tbase + 32 > cn) {
                        // ragged last tile: rows past the end must never win
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int j = tbase + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
                            for (int t = 0; t < QT; ++t)
                                acc[t][r] = j < cn ? acc[t][r] : INT32_MIN;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            b2[t] = med3i(b1[t], b2[t], acc[t][r]);
                            b1[t] = max(b1[t], acc[t][r]);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        cinit[r] -= 32;
                }
            }
        }

        // ---- fold the chunk's winners into global keys ---------------------------------
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int v[2] = {b1[t], b2[t]};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (v[e] != INT32_MIN) {
                    const uint32_t key = (uint32_t)(-v[e]);
                    const uint32_t gk = ((key >> JBITS) << KEY_SHIFT) | (uint32_t)(chunk0 + (int)(key & (CHUNK - 1)));
                    insert_key(g1[t], g2[t], gk);
                }
            }
        }
    }

    // ---- merge the two lane halves of each query, filter, store -------------------------
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const uint32_t o1 = __shfl_xor(g1[t], 32);
        const uint32_t o2 = __shfl_xor(g2[t], 32);
        insert_key(g1[t], g2[t], o1);
        insert_key(g1[t], g2[t], o2);
        const int row = q0 + (wave * QT + t) * 32 + col;
        if (half == 0 && row < qn)
            knn_store(a, qs, ts, os, row, g1[t], g2[t]);
    }
}

template <int KS, int QT>
int launch(const KnnArgs &a, int batch, int max_nq, hipStream_t s)
{
    KnnArgs args = a;
    args.tiles_per_item = (max_nq + 128 * QT - 1) / (128 * QT);
    const long long blocks = (long long)batch * args.tiles_per_item;
    if (blocks > 0x7FFFFFFFLL)
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: grid too large (%lld blocks)", blocks);
    hipLaunchKernelGGL((hamming_mfma_kernel<KS, QT>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, args);
    return check_launch("slam_hamming_match(mfma)");
}

}  // namespace

int launch_knn_mfma(const KnnArgs &a, int batch, int max_nq, int stride_bytes, hipStream_t s)
{
    auto blocks = [&](int qt) { return (long long)batch * ((max_nq + 128 * qt - 1) / (128 * qt)); };
    if (stride_bytes == 64) {
        if (blocks(2) >= 512)
            return launch<16, 2>(a, batch, max_nq, s);
        return launch<16, 1>(a, batch, max_nq, s);
    }
    if (blocks(2) >= 512)
        return launch<8, 2>(a, batch, max_nq, s);
    return launch<8, 1>(a, batch, max_nq, s);
}

}  // namespace slam
