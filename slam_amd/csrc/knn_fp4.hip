// Brute-force Hamming 2-NN on the FP4 matrix-core path of gfx950 -- the fastest way this chip has
// to take a dot product of bits: v_mfma_f32_32x32x64_f8f6f4 with e2m1 operands, 64 four-bit slots
// per instruction.  Train rows go straight from memory into registers: no LDS, no barrier, every
// wavefront on its own.
//
// A descriptor bit occupies one slot and stays where it is inside its 32-bit word:
//
//   bit 4n, 4n+1, 4n+2 of a train word become the fp4 values {0, 0.5}, {0, 1}, {0, 2} (one AND
//   each: 0001 = 0.5, 0010 = 1.0, 0100 = 2.0) and meet +-2, +-1, +-0.5 on the query side (+ for a
//   set query bit), so a slot contributes tb (2 qb - 1);
//   bit 4n+3 is the sign bit of its nibble: OR-ing 0001 under it gives -0.5 for a set bit and +0.5
//   for a clear one, the query side carries -1.0 for a set bit and +1.0 for a clear one, so the
//   slot contributes +0.5 where the bits agree and -0.5 where they differ.
//
//   With pop3(q) = set query bits outside the 4n+3 positions and n4 = number of 4n+3 slots of a
//   row (bits past the descriptor's length are cleared on both sides and count as agreeing):
//       dot(q, t) = pop3(q) + n4 / 2 - ham(q, t)
//   One operand register per VALU instruction, four per 32 descriptor bits.  (The sparser coding
//   {0, 0.5} against +-2 for bit 4n+3 needs a shift more and measured 2 % slower.)
//
// The row index rides below the distance: the accumulator starts from
// 2048 - (row offset inside the 32-row tile) / 2048 (the C operand of a tile's first MFMA), so a
// finished accumulator holds 2048 + dot - j / 2048: largest first = smallest distance first, then
// lowest index (cv2.BFMatcher's scan order), at no instruction per pair.  Every value is a
// positive multiple of 2^-11 below 4096, exact in fp32, and its bit pattern orders like an
// integer.  The two running winners are moved to the next tile's base (+32 / 2048) once per tile;
// every 2048 rows they are folded into the (distance << 22 | index) keys shared with the VALU
// kernel.  (A block scale of 2^11 on the B operand instead of the scaled-down C operand gives the
// same numbers but needs the two-word v_mfma_scale form: 1.5 % slower.)
//
// A wavefront keeps QT x 32 queries as B operands in VGPRs for the whole scan.  Lane (col, half)
// feeds k-slots of train row `col` of the current 32-row tile: half 0 the first half of the row's
// words, half 1 the second, so its share of a 64-byte row is 32 contiguous bytes (two 16-byte
// loads, issued two tiles ahead; the rows of an item stay in L2 for the wavefronts that share
// them).  Per tile: 4 KM expansion instructions, QT KM MFMAs, and the running top two at 1.25
// instructions per accumulator (0.5 for k = 1).  The update of a tile is issued in the shadow of
// the next tile's MFMAs (two accumulator sets, the order pinned): an MFMA occupies the matrix pipe
// for 8 passes and needs the issue port for one, which leaves room for about 7 VALU instructions.
// 210 VGPRs, 2 waves per SIMD.  At this point the loop is limited by the clock the chip holds
// under the matrix cores' load: denser operands (raw words in place of the expansion) run slower.
#include "knn_common.h"

namespace slam {

namespace {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x8 = int __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

constexpr int JBITS = 11;
constexpr int CHUNK = 1 << JBITS;     // train rows per key chunk
constexpr float UNIT = 1.0f / CHUNK;  // one row index
constexpr float BIAS = 2048.0f;       // |dot - j / 2048| < 1024
constexpr float SENT = 0.0f;          // "no candidate": stays far below every real value when moved along
__device__ __forceinline__ bool is_candidate(float v) { return v > 1024.0f; }

// The median goes through the builtin and the maxima are plain integer max() on the bit patterns
// (v_max3_i32): the compiler then sees ordinary VALU reads of the MFMA results and inserts the
// wait states they need.
__device__ __forceinline__ float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ __forceinline__ float max3_bits(float a, float b, float c)
{
    return __int_as_float(max(max(__float_as_int(a), __float_as_int(b)), __float_as_int(c)));
}

// Query side of 32 descriptor bits (w already masked to the descriptor's length).
__device__ __forceinline__ i32x4 expand_query(uint32_t w)
{
    const uint32_t n = ~w;
    i32x4 v;
    v.x = (int)(0x44444444u | ((n & 0x11111111u) << 3));   // +-2.0 against 0.5
    v.y = (int)(0x22222222u | ((n & 0x22222222u) << 2));   // +-1.0 against 1.0
    v.z = (int)(0x11111111u | ((n & 0x44444444u) << 1));   // +-0.5 against 2.0
    v.w = (int)(0x22222222u | (w & 0x88888888u));          // -+1.0 against -+0.5
    return v;
}

// fp4 operands take four of the eight operand registers; all-zero scale arguments select the
// unscaled instruction
__device__ __forceinline__ f32x16 mfma_fp4(i32x4 a, i32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(i32x8{a.x, a.y, a.z, a.w, 0, 0, 0, 0},
                                                           i32x8{b.x, b.y, b.z, b.w, 0, 0, 0, 0}, c, 4, 4, 0, 0, 0, 0);
}

// KM = MFMAs (64 bits each) per row: 8 for 64-byte rows, 4 for 32-byte; QT = query tiles per wave;
// K2 = keep the second neighbour too; MASKED = some word before a lane's last one is cut by the
// descriptor length (then every train word is masked first)
template <int KM, int QT, bool K2, bool MASKED>
__global__ void __launch_bounds__(BLOCK, 2) hamming_fp4_kernel(const KnnArgs a)
{
    constexpr int C = KM / 2;    // 16-byte pieces per packed row
    constexpr int LC = KM / 4;   // 16-byte pieces per lane: its half of the row
    constexpr int STEPS = 8 / KM;   // top-2 steps issued behind one MFMA

    // Workgroups are dealt to the 8 XCDs round-robin by blockIdx; renumber them so that each XCD
    // works through a contiguous run of (item, tile) pairs: the tiles of one item then share
    // that XCD's L2 copy of the item's train rows.
    const int nb = gridDim.x, xq = nb >> 3, xr = nb & 7;
    const int xcd = blockIdx.x & 7, nth = blockIdx.x >> 3;
    const int bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + nth;
    const int item = bid / a.tiles_per_item;
    const int tile = bid - item * a.tiles_per_item;
    const int32_t *seg = a.seg + SLAM_SEG_STRIDE * item;
    const int qs = seg[0], qn = seg[1], ts = seg[2], tn = seg[3], os = seg[4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq0 = (tile * 4 + wave) * (QT * 32);      // this wave's first query
    if (wq0 >= qn)
        return;                                          // waves never meet: no barrier in this kernel
    const int col = lane & 31;   // query within a tile / train row within a tile
    const int half = lane >> 5;  // which half of a row's words this lane feeds

    uint32_t wm[KM];             // masks of the lane's words KM half + m
#pragma unroll
    for (int m = 0; m < KM; ++m)
        wm[m] = word_mask(KM * half + m, a.desc_bytes);

    // ---- B operands: this wave's queries ------------------------------------------------------
    i32x4 bq[QT][KM];
    int popq[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int row = wq0 + t * 32 + col;
        const uint4 *src = a.q + (size_t)(qs + (row < qn ? row : 0)) * C + half * LC;
        int pop = 0;
#pragma unroll
        for (int c = 0; c < LC; ++c) {
            const uint4 v = src[c];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t w = w4[e] & wm[4 * c + e];
                pop += __popc(w & 0x77777777u);
                bq[t][4 * c + e] = expand_query(w);
            }
        }
        popq[t] = pop + __shfl_xor(pop, 32) + 8 * KM;     // pop3(q) + n4 / 2
    }

    // train-side constants of the lane's last word (the only one the descriptor may cut when !MASKED)
    const uint32_t lx = 0x11111111u & wm[KM - 1], ly = 0x22222222u & wm[KM - 1], lz = 0x44444444u & wm[KM - 1],
                   lw = 0x88888888u & wm[KM - 1];
    // (w & 0x88888888) | 0x11111111 is one v_and_or_b32 only with one of the constants in a VGPR
    // (one scalar operand per instruction); the asm keeps the compiler from turning it back into a literal
    uint32_t c8;
    asm("v_mov_b32 %0, 0x88888888" : "=v"(c8));
    auto expand_train = [&](uint32_t w, int m) __attribute__((always_inline)) -> i32x4 {
        i32x4 v;
        if (MASKED)
            w &= wm[m];
        if (!MASKED && m == KM - 1) {
            v.x = (int)(w & lx);
            v.y = (int)(w & ly);
            v.z = (int)(w & lz);
            v.w = (int)((w & lw) | 0x11111111u);
        } else {
            v.x = (int)(w & 0x11111111u);            // bit 4n     -> 0.5
            v.y = (int)(w & 0x22222222u);            // bit 4n + 1 -> 1.0
            v.z = (int)(w & 0x44444444u);            // bit 4n + 2 -> 2.0
            v.w = (int)((w & c8) | 0x11111111u);     // bit 4n + 3 -> -0.5 / +0.5
        }
        return v;
    };

    uint32_t g1[QT], g2[QT];   // running (distance << 22 | index) keys over finished chunks
    float b1[QT], b2[QT];      // two largest values of the current chunk
    float md[QT];              // median of an even update step, waiting for the odd one
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        g1[t] = g2[t] = KEY_EMPTY;
        b1[t] = b2[t] = md[t] = SENT;
    }

    // train row of accumulator register r inside a 32-row tile, for this lane
    auto row_off = [&](int r) __attribute__((always_inline)) { return (r & 3) + 8 * (r >> 2) + 4 * half; };
    f32x16 cneg;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        cneg[r] = BIAS - UNIT * (float)row_off(r);

    struct Raw {
        uint32_t w[KM];
    };
    const uint4 *tbase = a.t + (size_t)ts * C + half * LC;
    // the lane's words of train row 32 i + col (clamped into the item: a ragged tile's extra rows are masked later)
    auto load_tile = [&](int i) __attribute__((always_inline)) -> Raw {
        Raw r;
        int row = 32 * i + col;
        row = row < tn ? row : tn - 1;
        const uint4 *src = tbase + (size_t)row * C;
#pragma unroll
        for (int c = 0; c < LC; ++c) {
            const uint4 v = src[c];
            r.w[4 * c] = v.x;
            r.w[4 * c + 1] = v.y;
            r.w[4 * c + 2] = v.z;
            r.w[4 * c + 3] = v.w;
        }
        return r;
    };
    auto mfma_tile = [&](const Raw &raw, f32x16 (&acc)[QT]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < KM; ++m) {
            const i32x4 af = expand_train(raw.w[m], m);
#pragma unroll
            for (int t = 0; t < QT; ++t)
                acc[t] = mfma_fp4(af, bq[t][m], m == 0 ? cneg : acc[t]);
        }
    };
    // Step c of 8 of the top-2 update of query tile t against a finished tile's accumulators (two
    // accumulators per step: with b1 >= b2, the second largest of {b1, b2, x, y} is
    // max(b2, med3(b1, x, y)); the medians of two steps are folded into b2 by one three-way maximum;
    // step 0 first moves the winners to this tile's base): 2 to 4 instructions.
    auto top2_step = [&](const f32x16 (&acc)[QT], int t, int c) __attribute__((always_inline)) {
        if (c == 0) {
            b1[t] += 32.0f * UNIT;
            if (K2)
                b2[t] += 32.0f * UNIT;
        }
        const float x = acc[t][2 * c], y = acc[t][2 * c + 1];
        if (K2) {
            const float m = med3f(b1[t], x, y);
            if (c & 1)
                b2[t] = max3_bits(b2[t], md[t], m);
            else
                md[t] = m;
        }
        b1[t] = max3_bits(b1[t], x, y);
        // keeps the step where it is written: the optimiser otherwise moves these pure operations
        // before the scheduling barriers around them take effect
        if (K2)
            asm volatile("" : "+v"(b1[t]), "+v"(b2[t]), "+v"(md[t]));
        else
            asm volatile("" : "+v"(b1[t]));
    };
    // One tile's MFMAs with the top-2 update of the previous tile issued in their shadow.  The
    // order (expansion of a word; then per query tile an MFMA followed by update steps) is pinned:
    // left alone, the scheduler gathers the update at the two ends of the loop body, where nothing
    // hides it.
    auto mfma_tile_top2 = [&](const Raw &raw, f32x16 (&acc)[QT], const f32x16 (&old)[QT]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < KM; ++m) {
            const i32x4 af = expand_train(raw.w[m], m);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                acc[t] = mfma_fp4(af, bq[t][m], m == 0 ? cneg : acc[t]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < STEPS; ++e)
                    top2_step(old, t, STEPS * m + e);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // The same update in one piece (pipeline drain and the ragged last tile).  limit < 32: rows from
    // `limit` on do not exist.
    auto top2_tile = [&](const f32x16 (&acc)[QT], int limit) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            b1[t] += 32.0f * UNIT;
            if (K2)
                b2[t] += 32.0f * UNIT;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                v[r] = acc[t][r];
            if (limit < 32) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    v[r] = row_off(r) < limit ? v[r] : SENT;
            }
            float m[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (K2)
                    m[i] = med3f(b1[t], v[2 * i], v[2 * i + 1]);
                b1[t] = max3_bits(b1[t], v[2 * i], v[2 * i + 1]);
            }
            if (K2) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    b2[t] = max3_bits(b2[t], m[2 * i], m[2 * i + 1]);
            }
        }
    };
    // b1/b2 hold 2048 + dot - (j - base) / 2048, base = chunk-local first row of the tile merged last
    auto fold_chunk = [&](int chunk0, int base) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const float v[2] = {b1[t], b2[t]};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (is_candidate(v[e])) {
                    const int key = -(int)((v[e] - BIAS) * (float)CHUNK - (float)base);   // 2048 (ham - popq) + j, j in [0, 2048)
                    const int ham = (key >> JBITS) + popq[t];                             // arithmetic shift floors
                    const int j = key & (CHUNK - 1);
                    insert_key(g1[t], g2[t], ((uint32_t)ham << KEY_SHIFT) | (uint32_t)(chunk0 + j));
                }
            }
            b1[t] = b2[t] = SENT;
        }
    };

    for (int chunk0 = 0; chunk0 < tn; chunk0 += CHUNK) {
        const int rows = min(CHUNK, tn - chunk0);
        const int nfull = rows >> 5, rem = rows & 31;
        const int t0 = chunk0 >> 5;                      // item-wide index of the chunk's first tile
        Raw rx = load_tile(t0), ry = load_tile(t0 + 1);  // even / odd tiles of the chunk
        f32x16 accA[QT], accB[QT];
        int i = 0;                                       // full tiles whose MFMAs are issued
        if (nfull > 0) {
            mfma_tile(rx, accA);
            rx = load_tile(t0 + 2);
            i = 1;
            while (i + 1 < nfull) {
                mfma_tile_top2(ry, accB, accA);
                ry = load_tile(t0 + i + 2);
                __builtin_amdgcn_sched_barrier(0);
                mfma_tile_top2(rx, accA, accB);
                rx = load_tile(t0 + i + 3);
                __builtin_amdgcn_sched_barrier(0);
                i += 2;
            }
            if (i < nfull) {
                mfma_tile_top2(ry, accB, accA);
                top2_tile(accB, 32);
                ++i;
            } else {
                top2_tile(accA, 32);
            }
        }
        if (rem) {
            Raw rt;
#pragma unroll
            for (int m = 0; m < KM; ++m)
                rt.w[m] = (i & 1) ? ry.w[m] : rx.w[m];
            mfma_tile(rt, accA);
            top2_tile(accA, rem);
        }
        fold_chunk(chunk0, 32 * (nfull + (rem ? 1 : 0) - 1));
    }

    // ---- merge the two lane halves of each query, filter, store -------------------------------
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const uint32_t o1 = __shfl_xor(g1[t], 32);
        const uint32_t o2 = __shfl_xor(g2[t], 32);
        insert_key(g1[t], g2[t], o1);
        insert_key(g1[t], g2[t], o2);
        const int row = wq0 + t * 32 + col;
        if (half == 0 && row < qn)
            knn_store(a, qs, ts, os, row, g1[t], g2[t]);
    }
}

template <int KM, int QT, bool K2, bool MASKED>
int launch(const KnnArgs &a, int batch, int max_nq, hipStream_t s)
{
    KnnArgs args = a;
    args.tiles_per_item = (max_nq + 128 * QT - 1) / (128 * QT);
    const long long blocks = (long long)batch * args.tiles_per_item;
    if (blocks > 0x7FFFFFFFLL)
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: grid too large (%lld blocks)", blocks);
    hipLaunchKernelGGL((hamming_fp4_kernel<KM, QT, K2, MASKED>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, args);
    return check_launch("slam_hamming_match(fp4)");
}

template <int KM, int QT>
int launch_km(const KnnArgs &a, int batch, int max_nq, hipStream_t s)
{
    // a word before a lane's last one is cut iff the descriptor ends before the last word of the second half
    const bool masked = a.desc_bytes < 4 * (2 * KM - 1);
    const bool k2 = a.k >= 2;
    if (masked)
        return k2 ? launch<KM, QT, true, true>(a, batch, max_nq, s) : launch<KM, QT, false, true>(a, batch, max_nq, s);
    return k2 ? launch<KM, QT, true, false>(a, batch, max_nq, s) : launch<KM, QT, false, false>(a, batch, max_nq, s);
}

}  // namespace

int launch_knn_fp4(const KnnArgs &a, int batch, int max_nq, int stride_bytes, hipStream_t s)
{
    // two query tiles per wavefront halve the expansion work per MFMA; one gives small launches twice the wavefronts
    const bool wide = (long long)batch * ((max_nq + 255) / 256) >= 512;
    if (stride_bytes == 64)
        return wide ? launch_km<8, 2>(a, batch, max_nq, s) : launch_km<8, 1>(a, batch, max_nq, s);
    return wide ? launch_km<4, 2>(a, batch, max_nq, s) : launch_km<4, 1>(a, batch, max_nq, s);
}

}  // namespace slam
