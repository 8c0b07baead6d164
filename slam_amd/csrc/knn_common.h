// Shared pieces of the Hamming k-NN kernels (VALU popcount and fp4 matrix cores).
#pragma once
#include "common.h"

namespace slam {

constexpr int KEY_SHIFT = 22;
constexpr uint32_t KEY_IDX_MASK = (1u << KEY_SHIFT) - 1u;
constexpr uint32_t KEY_EMPTY = 0xFFFFFFFFu;
constexpr int TT = 256;     // train rows per LDS tile
constexpr int BLOCK = 256;  // threads per block

struct KnnArgs {
    const uint4 *q;
    const uint4 *t;
    const int32_t *seg;
    int tiles_per_item;
    int desc_bytes;
    int k;
    int mode;
    double thr;
    const float *yq;
    const float *yt;
    int32_t *out_idx;
    int32_t *out_dist;
    uint8_t *keep;
};

__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Mask that keeps the bytes of 32-bit word `w` that lie inside the descriptor.
__device__ __forceinline__ uint32_t word_mask(int w, int desc_bytes)
{
    const int lo = 4 * w;
    if (desc_bytes >= lo + 4)
        return 0xFFFFFFFFu;
    if (desc_bytes <= lo)
        return 0u;
    return (1u << (8 * (desc_bytes - lo))) - 1u;
}

__device__ __forceinline__ void insert_key(uint32_t &b1, uint32_t &b2, uint32_t key)
{
    b2 = med3u(b1, b2, key);
    b1 = min(b1, key);
}


// Decode the two best (distance << 22 | index) keys of one query, apply the fused
// filter and store one result row.
__device__ __forceinline__ void knn_store(const KnnArgs &a, int qs, int ts, int os, int row, uint32_t k1, uint32_t k2)
{
    const int i0 = (k1 == KEY_EMPTY) ? -1 : (int)(k1 & KEY_IDX_MASK);
    const int d0 = (k1 == KEY_EMPTY) ? INT32_MAX : (int)(k1 >> KEY_SHIFT);
    int i1 = (k2 == KEY_EMPTY) ? -1 : (int)(k2 & KEY_IDX_MASK);
    int d1 = (k2 == KEY_EMPTY) ? INT32_MAX : (int)(k2 >> KEY_SHIFT);
    if (a.k < 2) {
        i1 = -1;
        d1 = INT32_MAX;
    }
    const size_t g = (size_t)os + row;
    reinterpret_cast<int2 *>(a.out_idx)[g] = make_int2(i0, i1);
    reinterpret_cast<int2 *>(a.out_dist)[g] = make_int2(d0, d1);
    if (a.mode == 1) {
        a.keep[g] = (i1 >= 0) && ((double)d0 < a.thr * (double)d1);
    } else if (a.mode == 2) {
        bool kp = false;
        if (i0 >= 0) {
            const double dev = fabs((double)a.yq[qs + row] - (double)a.yt[ts + i0]);
            kp = dev <= a.thr;
        }
        a.keep[g] = kp;
    }
}

// Launches the fp4-MFMA kernel (knn_fp4.hip).
int launch_knn_fp4(const KnnArgs &a, int batch, int max_nq, int stride_bytes, hipStream_t s);

}  // namespace slam
