// Layout of the scoring workspace and the one routine that fills it.  The filtered scoring
// kernels read per-point records and per-set magnitude bounds from a caller-owned workspace;
// the records are written either by the scoring entry point itself (score_prep_kernel) or, one
// launch earlier, by the consensus join while it has the point in registers anyway.
//
// Workspace (in doubles, n_rec = n_sets * cap):
//   [n_rec][8]            fp64 point records  X Y Z uL vL uR vR pad
//   [n_rec][4]            fp32 point records (8 floats, threshold units, principal point removed)
//                            X Y Z (uL-cx)/thr (vL-cy)/thr (uR-cx)/thr (vR-cy)/thr A2
//                         A2 >= (|X|+|Y|+|Z|)^2, rounded up
//   [n_sets][4]           per-set bounds kept as the bit patterns of non-negative doubles:
//                         Amax = max(|X|+|Y|+|Z|), Umax = max(|uL|+|vL|+|uR|+|vR|),
//                         Wmax = max(|u-cx|+|fx|, |v-cy|+|fy|)/thr over both cameras, pad
#pragma once
#include "common.h"

namespace slam {

constexpr int SCORE_REC = 8;
constexpr int SCORE_REC32 = 8;   // floats
constexpr int SCORE_HDR = 4;
constexpr int SCORE_WORK_PER_REC = SCORE_REC + 4;   // doubles of workspace per point record

__device__ __forceinline__ double *score_work_recs64(double *work) { return work; }
__device__ __forceinline__ float *score_work_recs32(double *work, size_t n_rec)
{
    return reinterpret_cast<float *>(work + SCORE_REC * n_rec);
}
__device__ __forceinline__ double *score_work_hdr(double *work, size_t n_rec)
{
    return work + SCORE_WORK_PER_REC * n_rec;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}

// The bit pattern under which a magnitude takes part in the per-set maxima: non-negative doubles
// order like their patterns; a NaN maps to the largest pattern and ends up disabling the filters
// for the set.
__device__ __forceinline__ unsigned long long bound_pattern(double v)
{
    return (v == v) ? (unsigned long long)__double_as_longlong(v) : ~0ull;
}

// Writes the fp64 and fp32 records of point slot o and folds the point into the running bounds.
__device__ __forceinline__ void score_write_record(double *work, size_t n_rec, size_t o, double X, double Y, double Z,
                                                   float fuL, float fvL, float fuR, float fvR, double fx, double fy,
                                                   double cx, double cy, double thr, unsigned long long &am,
                                                   unsigned long long &um, unsigned long long &wm)
{
    const double uL = (double)fuL, vL = (double)fvL, uR = (double)fuR, vR = (double)fvR;
    double2 *rec = reinterpret_cast<double2 *>(score_work_recs64(work) + SCORE_REC * o);
    rec[0] = make_double2(X, Y);
    rec[1] = make_double2(Z, uL);
    rec[2] = make_double2(vL, uR);
    rec[3] = make_double2(vR, 0.0);
    const double A = (fabs(X) + fabs(Y)) + fabs(Z);
    const double U = (fabs(uL) + fabs(vL)) + (fabs(uR) + fabs(vR));
    const double ul = (uL - cx) / thr, vl = (vL - cy) / thr, ur = (uR - cx) / thr, vr = (vR - cy) / thr;
    const double W = fmax(fmax(fabs(ul), fabs(ur)) + fabs(fx / thr), fmax(fabs(vl), fabs(vr)) + fabs(fy / thr));
    float4 *r32 = reinterpret_cast<float4 *>(score_work_recs32(work, n_rec) + SCORE_REC32 * o);
    // A2 is an upper bound of A^2 after the rounding to fp32
    r32[0] = make_float4((float)X, (float)Y, (float)Z, (float)ul);
    r32[1] = make_float4((float)vl, (float)ur, (float)vr, (float)((A * A) * 1.000001));
    const unsigned long long pa = bound_pattern(A), pu = bound_pattern(U), pw = bound_pattern(W);
    am = pa > am ? pa : am;
    um = pu > um ? pu : um;
    wm = pw > wm ? pw : wm;
}

}  // namespace slam
