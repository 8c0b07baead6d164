// Brute-force Hamming 2-NN on the FP4 matrix-core path of gfx950 -- the fastest way this chip has
// to take a dot product of bits: v_mfma_scale_f32_32x32x64_f8f6f4 with e2m1 operands.
//
//   ham(q, t) = pop(q) - sum_k tb_k (2 qb_k - 1)
//
// A train bit becomes the fp4 value {0, v}, v = 0.5, 1, 2 by its position inside its nibble (one
// AND per 8 bits: the bit stays where it is, 0001 = 0.5, 0010 = 1.0, 0100 = 2.0; the nibble's top
// bit is moved down first); the query bit of the same slot is +-1/v, so every product is +-1 or 0
// and the fp32 accumulator holds pop(q) - ham exactly.  The block scale of the B operand is 2^10,
// which makes room for the row index below it: key = 1024 (pop(q) - ham) - j, largest first =
// smallest distance first, then lowest index (cv2.BFMatcher's scan order).  One instruction
// covers 64 bits of 32 x 32 descriptor pairs in about the time the int8 form takes for 32.
// The index costs no instruction per pair: the accumulator starts from BIAS - (row offset inside
// the tile) (the C operand of the first MFMA; the bias keeps every value positive), which makes
// every value relative to its tile's first row, and the two running winners are moved to the
// next tile's base (+32) once per tile.
//
// Structure as the int8 kernel (knn_mfma.hip): block = 4 waves, a wave keeps QT x 32 queries as
// B operands in VGPRs for the whole scan; train rows are expanded 64 at a time into LDS (double
// buffered, row stride 32 KM + 16 bytes: conflict-free ds_read_b128 A fragments); a tile's MFMAs
// are followed by its top-2 update, three waves per SIMD cover each other's waits; one barrier
// per super-tile.  The values are exact integers held in fp32 (below 2^23); the running top two
// per lane take 1.25 instructions per accumulator (v_max3_i32 on the bit patterns, v_med3_f32);
// every 1024 rows they are folded into the (distance << 22 | index) keys shared with the other
// kernels.
#include "knn_common.h"

namespace slam {

namespace {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x8 = int __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

constexpr int TS = 64;            // train rows per super-tile
constexpr int CHUNK = 1024;       // train rows per key chunk
constexpr int JBITS = 10;

// Every value is kept positive -- BIAS + key, with |key| < 2^20 -- so that the fp32 bit patterns
// order like the values and integer maxima can be taken on them.  "No candidate" is 0: it may be
// moved along with the running winners (+32 per tile, at most 32 tiles per chunk) and stays
// far below every real value.
constexpr float BIAS = 4194304.0f;         // 2^22
constexpr float SENT = 0.0f;
__device__ __forceinline__ bool is_candidate(float v) { return v > 2097152.0f; }

constexpr int SCALE_ONE = 0x7F7F7F7F;      // E8M0 block scales, 2^(e - 127): 1
constexpr int SCALE_1024 = 0x89898989;     //                                 1024

// The median goes through the builtin and the maxima are plain integer max() on the bit patterns
// (v_max3_i32): the compiler then sees ordinary VALU reads of the MFMA results and inserts the
// wait states they need -- it pads nothing around inline asm.
__device__ __forceinline__ float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ __forceinline__ float max3_bits(float a, float b, float c)
{
    return __int_as_float(max(max(__float_as_int(a), __float_as_int(b)), __float_as_int(c)));
}

// Train side: 32 bits -> 32 fp4 slots (one operand register quad).
__device__ __forceinline__ i32x4 expand_train(uint32_t w)
{
    i32x4 v;
    v.x = (int)(w & 0x11111111u);            // bit 4n     -> 0.5
    v.y = (int)(w & 0x22222222u);            // bit 4n + 1 -> 1.0
    v.z = (int)(w & 0x44444444u);            // bit 4n + 2 -> 2.0
    v.w = (int)((w >> 3) & 0x11111111u);     // bit 4n + 3 -> 0.5
    return v;
}

// Query side: the same slots carry +1/v for a 1 bit and -1/v for a 0 bit.
__device__ __forceinline__ i32x4 expand_query(uint32_t w)
{
    const uint32_t n = ~w;
    i32x4 v;
    v.x = (int)(0x44444444u | ((n & 0x11111111u) << 3));   // +-2.0
    v.y = (int)(0x22222222u | ((n & 0x22222222u) << 2));   // +-1.0
    v.z = (int)(0x11111111u | ((n & 0x44444444u) << 1));   // +-0.5
    v.w = (int)(0x44444444u | (n & 0x88888888u));          // +-2.0
    return v;
}

__device__ __forceinline__ i32x8 widen(i32x4 v) { return i32x8{v.x, v.y, v.z, v.w, 0, 0, 0, 0}; }

// KM = MFMAs (64 bits each) per row: 8 for 64-byte rows, 4 for 32-byte; QT = query tiles per wave;
// K2 = keep the second neighbour too (k = 2); k = 1 saves the median per accumulator
template <int KM, int QT, bool K2>
__global__ void __launch_bounds__(BLOCK, 3) hamming_fp4_kernel(const KnnArgs a)
{
    constexpr int C = KM / 2;           // 16-byte chunks per packed row
    constexpr int ROW = KM * 32 + 16;   // bytes between expanded rows in LDS
    __shared__ __attribute__((aligned(16))) uint8_t lds[2][TS * ROW];

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
    const int q0 = tile * (4 * QT * 32);
    if (q0 >= qn)
        return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31;   // query within a tile / train row within an A fragment
    const int half = lane >> 5;  // which 32 of the 64 k-slots of an MFMA this lane feeds

    // ---- B operands: this wave's queries; MFMA m takes word 2m + half of the row ------------
    i32x4 bq[QT][KM];
    int popq[QT];
    const uint32_t hmask = 0u - (uint32_t)half;     // all ones in the upper lane half
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
            for (int e = 0; e < 4; ++e)
                popq[t] += __popc(w4[e] & word_mask(4 * c + e, a.desc_bytes));
#pragma unroll
            for (int p = 0; p < 2; ++p)        // words 4c + 2p (half 0) and 4c + 2p + 1 (half 1) -> MFMA 2c + p
                bq[t][2 * c + p] = expand_query((w4[2 * p] & ~hmask) | (w4[2 * p + 1] & hmask));
        }
    }

    uint32_t g1[QT], g2[QT];   // running (distance << 22 | index) keys over finished chunks
    float b1[QT], b2[QT];      // two largest keys of the current chunk
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
    // word 4 spart + e of the row -> 16 bytes of fp4 at 16 * (word index): MFMA (word >> 1), half (word & 1)
    auto expand_store = [&](uint8_t *buf, uint32_t w, int e) __attribute__((always_inline)) {
        if (stager)
            *reinterpret_cast<i32x4 *>(buf + srow * ROW + 16 * (4 * spart + e)) = expand_train(w & smask[e]);
    };
    // offset of accumulator register r's train row inside a 32-row tile, for this lane
    auto row_off = [&](int r) __attribute__((always_inline)) { return (r & 3) + 8 * (r >> 2) + 4 * half; };
    // Running top two of query tile t against the 16 accumulators v[0..15] of a tile, two at a
    // time: with b1 >= b2, the second largest of {b1, b2, x, y} is max(b2, med3(b1, x, y)); the
    // medians of four such steps are folded into b2 by two three-way maxima.  1.25 instructions
    // per accumulator (0.5 when only the nearest neighbour is wanted).
    auto top2_tile = [&](int t, const float (&v)[16]) __attribute__((always_inline)) {
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
    };
    // b1/b2 hold key + (chunk-local first row of the tile merged last); moving on by one tile:
    auto rebase = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            b1[t] += 32.0f;
            b2[t] += 32.0f;
        }
    };

    const int n_st = (tn + TS - 1) / TS;
    f32x16 cneg;               // BIAS - (row offset of accumulator register r inside a tile), without the lane's 4 * half
#pragma unroll
    for (int r = 0; r < 16; ++r)
        cneg[r] = BIAS - (float)((r & 3) + 8 * (r >> 2));

    if (n_st > 0) {
        const uint4 r0 = load_raw(0);
        const uint32_t w0[4] = {r0.x, r0.y, r0.z, r0.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            expand_store(lds[0], w0[e], e);
    }
    uint4 raw = n_st > 1 ? load_raw(1) : make_uint4(0, 0, 0, 0);
    __syncthreads();

    // One 32-row tile: its MFMAs, the expansion of two words of the next super-tile, then the
    // top-2 update over the accumulators (rows past the end of the item masked out when the
    // tile is the ragged last one).  The lane's 4 * half is added when the chunk is folded.
    auto tile_phase = [&](const uint8_t *abase, int row0, uint8_t *nbuf, uint32_t wa, int ea, uint32_t wb, int eb,
                          bool expand) __attribute__((always_inline)) {
        const uint8_t *arow = abase + col * ROW + 16 * half;
        f32x16 acc[QT];
#pragma unroll
        for (int m = 0; m < KM; ++m) {
            const i32x4 af = *reinterpret_cast<const i32x4 *>(arow + 32 * m);
#pragma unroll
            for (int t = 0; t < QT; ++t)
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(af), widen(bq[t][m]), m == 0 ? cneg : acc[t],
                                                                         4, 4, 0, SCALE_ONE, 0, SCALE_1024);
        }
        if (expand) {
            expand_store(nbuf, wa, ea);
            expand_store(nbuf, wb, eb);
        }
        rebase();
        if (row0 + 32 <= tn) {              // the common case: no select instructions
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    v[r] = acc[t][r];
                top2_tile(t, v);
            }
        } else {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    v[r] = row0 + row_off(r) < tn ? acc[t][r] : SENT;
                top2_tile(t, v);
            }
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

        tile_phase(lds[cur], row0, nbuf, raw.x, 0, raw.y, 1, have_next);
        const bool b_valid = row0 + 32 < tn;
        if (b_valid)
            tile_phase(lds[cur] + 32 * ROW, row0 + 32, nbuf, raw.z, 2, raw.w, 3, have_next);
        raw = rawn;

        const bool chunk_end = ((st + 1) * TS) % CHUNK == 0 || !have_next;
        if (chunk_end) {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                // fold the chunk's winners into global keys
                const int chunk0 = row0 & ~(CHUNK - 1);
                const float base = (float)(jb + (b_valid ? 32 : 0));   // first row of the tile merged last
                const float v[2] = {b1[t], b2[t]};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (is_candidate(v[e])) {
                        const int key = -(int)(v[e] - BIAS - base);  // 1024 * (ham - pop(q)) + j', j' in [0, 1024)
                        const int ham = (key >> JBITS) + popq[t];    // arithmetic shift floors
                        const int j = (key & (CHUNK - 1)) + 4 * half;
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

template <int KM, int QT, bool K2>
int launch(const KnnArgs &a, int batch, int max_nq, hipStream_t s)
{
    KnnArgs args = a;
    args.tiles_per_item = (max_nq + 128 * QT - 1) / (128 * QT);
    const long long blocks = (long long)batch * args.tiles_per_item;
    if (blocks > 0x7FFFFFFFLL)
        return fail(SLAM_ERR_BAD_ARG, "slam_hamming_match: grid too large (%lld blocks)", blocks);
    hipLaunchKernelGGL((hamming_fp4_kernel<KM, QT, K2>), dim3((unsigned)blocks), dim3(BLOCK), 0, s, args);
    return check_launch("slam_hamming_match(fp4)");
}

}  // namespace

int launch_knn_fp4(const KnnArgs &a, int batch, int max_nq, int stride_bytes, hipStream_t s)
{
    auto blocks = [&](int qt) { return (long long)batch * ((max_nq + 128 * qt - 1) / (128 * qt)); };
    const bool wide = blocks(2) >= 512;
    if (stride_bytes == 64) {
        if (a.k >= 2)
            return wide ? launch<8, 2, true>(a, batch, max_nq, s) : launch<8, 1, true>(a, batch, max_nq, s);
        return wide ? launch<8, 2, false>(a, batch, max_nq, s) : launch<8, 1, false>(a, batch, max_nq, s);
    }
    if (a.k >= 2)
        return wide ? launch<4, 2, true>(a, batch, max_nq, s) : launch<4, 1, true>(a, batch, max_nq, s);
    return wide ? launch<4, 2, false>(a, batch, max_nq, s) : launch<4, 1, false>(a, batch, max_nq, s);
}

}  // namespace slam
