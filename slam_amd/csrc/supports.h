// The supporter test of one pose hypothesis for one consensus match -- what the reference's
// find_supporters evaluates per match (utils/utils.py:389-410): project the point into both
// cameras of the later frame, compare both pixel distances with the threshold.  fp64, the
// reference's operation order, one rounding per operation (the library is built with
// -ffp-contract=off), strict `<`.  This function DEFINES a verdict: every filtered scoring path
// falls back to it, and the inlier tables are built from it.
#pragma once
#include "common.h"

namespace slam {

struct Mat33 {
    double m[9];
};
struct Vec3 {
    double v[3];
};

__device__ __forceinline__ bool supports(const double *__restrict__ rt, const Mat33 &K, const Vec3 &tr, double X,
                                         double Y, double Z, double uL, double vL, double uR, double vR,
                                         double thresh)
{
    const double L0 = ((rt[0] * X + rt[1] * Y) + rt[2] * Z) + rt[3];
    const double L1 = ((rt[4] * X + rt[5] * Y) + rt[6] * Z) + rt[7];
    const double L2 = ((rt[8] * X + rt[9] * Y) + rt[10] * Z) + rt[11];
    const double R0 = L0 + tr.v[0], R1 = L1 + tr.v[1], R2 = L2 + tr.v[2];
    const double hl0 = (K.m[0] * L0 + K.m[1] * L1) + K.m[2] * L2;
    const double hl1 = (K.m[3] * L0 + K.m[4] * L1) + K.m[5] * L2;
    const double hl2 = (K.m[6] * L0 + K.m[7] * L1) + K.m[8] * L2;
    const double hr0 = (K.m[0] * R0 + K.m[1] * R1) + K.m[2] * R2;
    const double hr1 = (K.m[3] * R0 + K.m[4] * R1) + K.m[5] * R2;
    const double hr2 = (K.m[6] * R0 + K.m[7] * R1) + K.m[8] * R2;
    const double plx = hl0 / hl2, ply = hl1 / hl2;
    const double prx = hr0 / hr2, pry = hr1 / hr2;
    const double dlx = uL - plx, dly = vL - ply;
    const double drx = uR - prx, dry = vR - pry;
    const double dl = sqrt(dlx * dlx + dly * dly);
    const double dr = sqrt(drx * drx + dry * dry);
    return (dl < thresh) && (dr < thresh);
}

}  // namespace slam
