// Batched Hamming 2-NN on the matrix cores (gfx950, v_mfma_i32_32x32x32_i8).
//
// Hamming distance as an int8 dot product.  For a query q and a train row t
//        ham(q, t) = pop(q) + pop(t) - 2 pop(q & t) = pop(q) - sum_k tb_k (2 qb_k - 1),
// so with train bits expanded to {0, tau} and query bits to {-sigma, +sigma}, sigma*tau = 1024,
//        sum_k q_k t_k = 1024 * (pop(q) - ham(q, t))
// exactly, in the int32 accumulator.  Subtracting the train row's index j inside its chunk of
// 1024 rows gives  -(1024 * (ham - pop(q)) + j): the larger the value, the better the match for
// this query, ties towards the lower train index -- cv2's order -- so the running top-2 per
// query is one v_max_i32 + one v_med3_i32 per result.  pop(q) is added back when a chunk folds
// into the (distance << 22 | index) keys of the epilogue.
//
// The index term.  When the last 32-bit word of a descriptor row has two pad bytes (the
// 61-byte AKAZE descriptor in a 64-byte row does), their k-slots carry j itself: the train side
// stores (j & 31, j >> 5), the query side the weights (-1, -32), and the MFMA delivers
// 1024*(pop(q)-ham) - j with no further work (EMBED).  Otherwise one v_sub per result subtracts
// the wave-uniform part of j; the per-lane part (4 * lane half) is a constant of the lane,
// irrelevant to its own top-2, and is added when the chunk folds.
//
// Bit expansion.  Which bit lands in which k-slot does not matter as long as both sides agree,
// so a 32-bit word w becomes one 32-wide k-step with ONE v_and_b32 per four bits:
//        dword p     = w        & (0x10101010 << p)      bytes in {0, 16, 32, 64, -128}[p]
//        dword 4 + p = (w << 4) & (0x10101010 << p)      p = 0..3
// (tau = 16, 32, 64, -128; the query side carries sigma = 64, 32, 16, -8 in the same slots).
//
// Work decomposition
//   block  = 4 wavefronts; wave w owns QT tiles of 32 queries whose expanded descriptors stay
//            in VGPRs as MFMA B operands for the whole scan
//   trains = super-tiles of 64 rows, double-buffered in LDS (rows 528 B apart: conflict-free
//            ds_read_b128 A operands); the expansion of super-tile s+1 and the top-2 update of
//            the previous 32-row tile are sliced between the MFMAs of the current tile, one
//            barrier per super-tile
//   top-2  = per lane over its 16 accumulators (16 train rows x 1 query); the two lane halves
//            of a query are merged with one shuffle at the end.
#include "knn_common.h"

namespace slam {

namespace {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

constexpr int TS = 64;            // train rows per super-tile
constexpr int CHUNK = 1024;       // train rows per key chunk
constexpr int JBITS = 10;

// "No candidate": real values lie above -2^20 (|ham - pop(q)| <= 512, j < 1024); the sentinel may
// have a row index subtracted from it like any other value and still stays below -2^29.
constexpr int SENT = -(1 << 30);
__device__ __forceinline__ bool is_candidate(int v) { return v > -(1 << 29); }

__device__ __forceinline__ int med3i(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Train side: 32 bits -> 32 int8 slots (two operand register quads).
__device__ __forceinline__ void expand_train(uint32_t w, i32x4 &hi, i32x4 &lo)
{
    const uint32_t s = w << 4;
    hi.x = (int)(w & 0x10101010u);
    hi.y = (int)(w & 0x20202020u);
    hi.z = (int)(w & 0x40404040u);
    hi.w = (int)(w & 0x80808080u);
    lo.x = (int)(s & 0x10101010u);
    lo.y = (int)(s & 0x20202020u);
    lo.z = (int)(s & 0x40404040u);
    lo.w = (int)(s & 0x80808080u);
}

// Query side: the same slots carry +sigma for a 1 bit and -sigma for a 0 bit.
__device__ __forceinline__ i32x4 expand_query(uint32_t src)
{
    auto plane = [&](int p, uint32_t pos, uint32_t neg) {
        const uint32_t m = ((src >> (4 + p)) & 0x01010101u) * 0xFFu;   // 0xFF where the bit is set
        return (int)((m & (pos * 0x01010101u)) | (~m & (neg * 0x01010101u)));
    };
    i32x4 v;
    v.x = plane(0, 0x40u, 0xC0u);   // +-64 against tau = 16
    v.y = plane(1, 0x20u, 0xE0u);   // +-32 against tau = 32
    v.z = plane(2, 0x10u, 0xF0u);   // +-16 against tau = 64
    v.w = plane(3, 0xF8u, 0x08u);   // -+8  against tau = -128
    return v;
}

// KS = k-steps of 32 bits (8 for 32-byte rows, 16 for 64-byte), QT = query tiles per wave,
// EMBED = the row index rides in two pad k-slots (needs desc_bytes <= 4*KS - 3)
template <int KS, int QT, bool EMBED>
__global__ void __launch_bounds__(BLOCK, 2) hamming_mfma_kernel(const KnnArgs a)
{
    constexpr int C = KS / 4;           // 16-byte chunks per packed row
    constexpr int ROW = KS * 32 + 16;   // bytes between expanded rows in LDS
    constexpr int RPS = 16 / KS;        // accumulator registers handled per k-step slice
    __shared__ __attribute__((aligned(16))) uint8_t lds[2][TS * ROW];

    // Workgroups are dealt to the 8 XCDs round-robin by blockIdx; renumber them so that each XCD
    // works through a contiguous run of (item, tile) pairs: the tiles of one item then share
    // that XCD's L2 copy of the item's train rows instead of fetching it into every L2.
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
    int popq[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int row = q0 + (wave * QT + t) * 32 + col;
        const bool ok = row < qn;
        const uint4 *src = a.q + (size_t)(qs + (ok ? row : 0)) * C;
        popq[t] = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const uint4 v = src[c];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ks = 4 * c + e;
                const uint32_t w = w4[e] & word_mask(ks, a.desc_bytes);
                popq[t] += __popc(w);
                bq[t][ks] = expand_query(half ? (w << 4) : w);
                if (EMBED && ks == KS - 1 && half == 0)   // slots of pad bytes 1 and 2: weights -1 and -32
                    bq[t][ks].x = (int)(((uint32_t)bq[t][ks].x & 0xFF0000FFu) | 0x00E0FF00u);
            }
        }
    }

    uint32_t g1[QT], g2[QT];   // running (distance << 22 | index) keys over finished chunks
    int b1[QT], b2[QT];        // two largest accumulators of the current chunk
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        g1[t] = g2[t] = KEY_EMPTY;
        b1[t] = b2[t] = SENT;
    }

    // staging assignment: thread -> (row = tid % 64, 16-byte part = tid / 64)
    const int srow = tid & 63;
    const int spart = tid >> 6;
    const bool stager = spart < C;
    uint32_t smask[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        smask[e] = word_mask(4 * spart + e, a.desc_bytes);
    auto load_raw = [&](int st) __attribute__((always_inline)) -> uint4 {
        const int row = st * TS + srow;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (stager && row < tn)
            v = a.t[(size_t)(ts + row) * C + spart];
        return v;
    };
    // `j`: index of the staged row inside its key chunk (EMBED: written into the two index slots)
    auto expand_store = [&](uint8_t *buf, uint32_t w, int e, int j) __attribute__((always_inline)) {
        if (stager) {
            i32x4 hi, lo;
            expand_train(w & smask[e], hi, lo);
            if (EMBED && e == 3 && spart == C - 1)
                hi.x |= ((j & 31) << 8) | ((j >> 5) << 16);
            uint8_t *dst = buf + srow * ROW + spart * 128 + 32 * e;
            *reinterpret_cast<i32x4 *>(dst) = hi;
            *reinterpret_cast<i32x4 *>(dst + 16) = lo;
        }
    };
    // offset of accumulator register r's train row inside a 32-row tile, for this lane
    auto row_off = [&](int r) __attribute__((always_inline)) { return (r & 3) + 8 * (r >> 2) + 4 * half; };
    auto top2 = [&](int t, int v) __attribute__((always_inline)) {
        b2[t] = med3i(b1[t], b2[t], v);
        b1[t] = max(b1[t], v);
    };

    const int n_st = (tn + TS - 1) / TS;
    i32x16 accA[QT], accB[QT];
    const i32x16 zero = {};
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            accB[t][r] = SENT;

    if (n_st > 0) {
        const uint4 r0 = load_raw(0);
        const uint32_t w0[4] = {r0.x, r0.y, r0.z, r0.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            expand_store(lds[0], w0[e], e, srow);
    }
    uint4 raw = n_st > 1 ? load_raw(1) : make_uint4(0, 0, 0, 0);
    __syncthreads();

    // One 32-row tile on the matrix cores with, between its MFMAs, slices of: the top-2 update
    // of the previous tile (`pend`, chunk-local first row pend_jb; all-sentinel when there is
    // none) and the expansion of two words of the next super-tile.
    auto tile_phase = [&](i32x16 (&acc)[QT], const i32x16 (&pend)[QT], const uint8_t *abase, int pend_jb,
                          uint8_t *nbuf, uint32_t wa, int ea, uint32_t wb, int eb, bool expand, int next_j)
                          __attribute__((always_inline)) {
        const uint8_t *arow = abase + col * ROW + 16 * half;
        i32x4 af = *reinterpret_cast<const i32x4 *>(arow);
        // `pend` was written by the MFMAs of the previous phase and is read below by inline asm
        // (v_med3_i32), around which the compiler inserts no wait states: spend the ones an MFMA
        // result needs before a VALU read here, once per phase.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 9\n\ts_nop 9");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            i32x4 afn = af;
            if (ks + 1 < KS)
                afn = *reinterpret_cast<const i32x4 *>(arow + 32 * (ks + 1));
#pragma unroll
            for (int t = 0; t < QT; ++t)
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bq[t][ks], ks == 0 ? zero : acc[t], 0, 0, 0);
#pragma unroll
            for (int rr = 0; rr < RPS; ++rr) {
                const int r = ks * RPS + rr;
                const int sj = pend_jb + ((r & 3) + 8 * (r >> 2));   // wave-uniform part of j
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    top2(t, EMBED ? pend[t][r] : pend[t][r] - sj);
            }
            if (expand && ks == 1)
                expand_store(nbuf, wa, ea, next_j);
            if (expand && ks == KS / 2 + 1)
                expand_store(nbuf, wb, eb, next_j);
            af = afn;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto top2_masked = [&](const i32x16 (&acc)[QT], int row0) __attribute__((always_inline)) {
        const int jb = row0 & (CHUNK - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool in = row0 + row_off(r) < tn;
#pragma unroll
            for (int t = 0; t < QT; ++t)
                top2(t, in ? (EMBED ? acc[t][r] : acc[t][r] - (jb + (r & 3) + 8 * (r >> 2))) : SENT);
        }
    };

    for (int st = 0; st < n_st; ++st) {
        const int cur = st & 1;
        const int row0 = st * TS;                       // first train row of the super-tile
        const int jb = row0 & (CHUNK - 1);              // its index inside the key chunk
        const bool have_next = st + 1 < n_st;           // then this super-tile is full
        uint4 rawn = make_uint4(0, 0, 0, 0);
        if (st + 2 < n_st)
            rawn = load_raw(st + 2);
        uint8_t *nbuf = lds[cur ^ 1];

        // accB still holds the second tile of the previous super-tile (same chunk: jb - 32)
        const int next_j = ((row0 + TS) & (CHUNK - 1)) + srow;   // chunk-local index of the row this thread stages
        tile_phase(accA, accB, lds[cur], jb - 32, nbuf, raw.x, 0, raw.y, 1, have_next, next_j);
        bool b_valid = false;
        if (row0 + 32 < tn) {
            tile_phase(accB, accA, lds[cur] + 32 * ROW, jb, nbuf, raw.z, 2, raw.w, 3, have_next, next_j);
            b_valid = true;
        } else {
            top2_masked(accA, row0);
        }
        raw = rawn;

        const bool chunk_end = ((st + 1) * TS) % CHUNK == 0 || !have_next;
        if (chunk_end) {
            if (b_valid)
                top2_masked(accB, row0 + 32);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    accB[t][r] = SENT;
                // fold the chunk's winners into global keys
                const int chunk0 = row0 & ~(CHUNK - 1);
                const int v[2] = {b1[t], b2[t]};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (is_candidate(v[e])) {
                        const int key = -v[e];                       // 1024 * (ham - pop(q)) + j', j' in [0, 1024)
                        const int ham = (key >> JBITS) + popq[t];    // arithmetic shift floors
                        const int j = (key & (CHUNK - 1)) + (EMBED ? 0 : 4 * half);
                        const uint32_t gk = ((uint32_t)ham << KEY_SHIFT) | (uint32_t)(chunk0 + j);
                        insert_key(g1[t], g2[t], gk);
                    }
                }
                b1[t] = b2[t] = SENT;
            }
        }
        __syncthreads();
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

template <int KS, int QT, bool EMBED>
int launch(const KnnArgs &a, int batch, int max_nq, hipStream_t s)
{
    KnnArgs args = a;
    args.tiles_per_item = (max_nq + 128 * QT - 1) / (128 * QT);
    const long long blocks = (long long)batch * args.tiles_per_item;
    if (blocks > 0x7FFFFFFFLL)
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: grid too large (%lld blocks)", blocks);
    hipLaunchKernelGGL((hamming_mfma_kernel<KS, QT, EMBED>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, args);
    return check_launch("slam_hamming_match(mfma)");
}

}  // namespace

int launch_knn_mfma(const KnnArgs &a, int batch, int max_nq, int stride_bytes, hipStream_t s)
{
    auto blocks = [&](int qt) { return (long long)batch * ((max_nq + 128 * qt - 1) / (128 * qt)); };
    const bool wide = blocks(2) >= 512;
    const bool embed = a.desc_bytes <= stride_bytes - 3;   // two pad bytes in the last word carry the row index
    if (stride_bytes == 64) {
        if (embed)
            return wide ? launch<16, 2, true>(a, batch, max_nq, s) : launch<16, 1, true>(a, batch, max_nq, s);
        return wide ? launch<16, 2, false>(a, batch, max_nq, s) : launch<16, 1, false>(a, batch, max_nq, s);
    }
    if (embed)
        return wide ? launch<8, 2, true>(a, batch, max_nq, s) : launch<8, 1, true>(a, batch, max_nq, s);
    return wide ? launch<8, 2, false>(a, batch, max_nq, s) : launch<8, 1, false>(a, batch, max_nq, s);
}

}  // namespace slam
