// PnP-RANSAC hypothesis scoring (find_supporters, utils/utils.py:389-410) on gfx950.
//
// The verdict for one (hypothesis, point) evaluation is DEFINED by `supports()` below:
// fp64, the reference's operation order, one IEEE rounding per operation (this file is
// compiled with -ffp-contract=off), strict '<' on both pixel distances.
//
// Two kernels produce that verdict:
//
//  score_kernel       lane = point, one wavefront per 2 hypotheses; evaluates
//                     supports() for every pair; ballot/popcount gives the count and
//                     the supporter bit mask.  Used when masks are wanted, for
//                     general K, or without a workspace.
//
//  score_fast_kernel  lane = hypothesis (pose in VGPRs), the point loop is
//                     wave-uniform so point data arrives through scalar loads and
//                     costs no VALU.  Each evaluation first runs a division- and
//                     sqrt-free *filter* in fused fp64 arithmetic:
//                         dx = u*hz - hx,  dy = v*hz - hy,   (hx,hy,hz) = (K @ Rt) @ [X,1]
//                         s = dx^2 + dy^2   versus   T = (thresh*hz)^2
//                     (the reference's test multiplied through by hz^2).  With a
//                     per-point relative guard band kappa the filter answers "in" if
//                     s - T < -kappa*(s+T), "out" if s - T > kappa*(s+T) for either
//                     camera, and otherwise -- or when hz is tiny against the
//                     magnitude of the terms it was summed from, or anything is
//                     NaN/Inf -- the lane falls through to supports().  The band is
//                     ~2^-26 * (|K| + |uv| + thresh)/thresh, at least 16x the worst
//                     rounding gap between the two evaluation orders, so the filter
//                     never contradicts supports(): counts are bit-identical, only
//                     ~4x cheaper (about 40 fp64 VALU ops per evaluation instead of
//                     ~60 plus four divisions and two square roots).
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "common.h"
#include "score_records.h"
#include "supports.h"

namespace slam {
namespace {

struct ScoreArgs {
    const double *Rt;
    int H;             // hypotheses scored per set
    int Hs;            // hypotheses per set in memory (row stride of Rt, counts and masks)
    const int32_t *skip;   // nullable: sets with a non-zero entry are left untouched
    Mat33 K;
    Vec3 tr;
    const double *X, *Y, *Z;
    const float *uL, *vL, *uR, *vR;
    int cap;
    const int32_t *n_dev;
    int n_host;
    double thresh;
    int32_t *counts;
    unsigned long long *masks;
    int mask_words;
    int blocks_per_set;
    double *work;      // point records + per-set bounds (fast path only)
    size_t n_rec;      // n_sets * cap
    double trmax;      // max |t_right[i]|
};

constexpr int SCORE_HPW = 2;  // hypotheses per wavefront (share the point loads)

__global__ void __launch_bounds__(256) score_kernel(const ScoreArgs a)
{
    const int set = blockIdx.x / a.blocks_per_set;
    const int blk = blockIdx.x - set * a.blocks_per_set;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h0 = (blk * 4 + wave) * SCORE_HPW;
    if (h0 >= a.H || (a.skip && a.skip[set]))
        return;
    int n = a.n_dev ? a.n_dev[set] : a.n_host;
    n = min(max(n, 0), a.cap);
    const size_t pbase = (size_t)set * a.cap;
    double rt[SCORE_HPW][12];
    bool live[SCORE_HPW];
#pragma unroll
    for (int e = 0; e < SCORE_HPW; ++e) {
        live[e] = (h0 + e) < a.H;
        const double *src = a.Rt + ((size_t)set * a.Hs + (live[e] ? h0 + e : h0)) * 12;
#pragma unroll
        for (int c = 0; c < 12; ++c)
            rt[e][c] = src[c];
    }
    int cnt[SCORE_HPW];
#pragma unroll
    for (int e = 0; e < SCORE_HPW; ++e)
        cnt[e] = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool in = i < n;
        const size_t o = pbase + (in ? i : 0);
        const double X = a.X[o], Y = a.Y[o], Z = a.Z[o];
        const double uL = (double)a.uL[o], vL = (double)a.vL[o], uR = (double)a.uR[o], vR = (double)a.vR[o];
#pragma unroll
        for (int e = 0; e < SCORE_HPW; ++e) {
            const bool s = in && supports(rt[e], a.K, a.tr, X, Y, Z, uL, vL, uR, vR, a.thresh);
            const unsigned long long m = __ballot(s);
            cnt[e] += __popcll(m);
            if (a.masks && live[e] && lane == 0)
                a.masks[((size_t)set * a.Hs + h0 + e) * a.mask_words + (base >> 6)] = m;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < SCORE_HPW; ++e)
            if (live[e])
                a.counts[(size_t)set * a.Hs + h0 + e] = cnt[e];
    }
}

// Clears mask words that the scoring loop does not reach (words past ceil(n/64)).
__global__ void __launch_bounds__(256) clear_mask_tail_kernel(unsigned long long *masks, int n_sets, int H,
                                                              int mask_words, const int32_t *n_dev, int n_host, int cap)
{
    const size_t total = (size_t)n_sets * H * mask_words;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(e % mask_words);
        const int set = (int)(e / ((size_t)H * mask_words));
        int n = n_dev ? n_dev[set] : n_host;
        n = min(max(n, 0), cap);
        if (w >= (n + 63) / 64)
            masks[e] = 0ull;
    }
}

// ---- fast path ---------------------------------------------------------------
// (workspace layout: score_records.h)
constexpr int SCORE_QUEUE = 16;  // undecided points a lane can park before it falls back to fp64 for all

__device__ __forceinline__ double *score_recs64(const ScoreArgs &a) { return score_work_recs64(a.work); }
__device__ __forceinline__ float *score_recs32(const ScoreArgs &a) { return score_work_recs32(a.work, a.n_rec); }
__device__ __forceinline__ double *score_hdr(const ScoreArgs &a) { return score_work_hdr(a.work, a.n_rec); }

// Widens the point data to fp64 records, writes the fp32 records and reduces the per-set
// magnitude bounds (non-negative doubles order like their bit patterns, so an integer atomic
// max does it).  A NaN magnitude maps to the largest pattern and ends up disabling the
// filters for the set.
__global__ void __launch_bounds__(256) score_prep_kernel(const ScoreArgs a, int chunks_per_set)
{
    __shared__ unsigned long long red[3][4];
    const int set = blockIdx.x / chunks_per_set;
    const int chunk = blockIdx.x - set * chunks_per_set;
    if (a.skip && a.skip[set])
        return;
    int n = a.n_dev ? a.n_dev[set] : a.n_host;
    n = min(max(n, 0), a.cap);
    const int i = chunk * 256 + threadIdx.x;
    unsigned long long am = 0ull, um = 0ull, wm = 0ull;
    if (i < n) {
        const size_t o = (size_t)set * a.cap + i;
        const double X = a.X[o], Y = a.Y[o], Z = a.Z[o];
        score_write_record(a.work, a.n_rec, o, X, Y, Z, a.uL[o], a.vL[o], a.uR[o], a.vR[o], a.K.m[0], a.K.m[4], a.K.m[2],
                           a.K.m[5], a.thresh, am, um, wm);
    }
    am = wave_max_u64(am);
    um = wave_max_u64(um);
    wm = wave_max_u64(wm);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = am;
        red[1][threadIdx.x >> 6] = um;
        red[2][threadIdx.x >> 6] = wm;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long m = red[threadIdx.x][0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            m = red[threadIdx.x][w] > m ? red[threadIdx.x][w] : m;
        unsigned long long *hdr = reinterpret_cast<unsigned long long *>(score_hdr(a)) + SCORE_HDR * set;
        atomicMax(hdr + threadIdx.x, m);
    }
}

// One hypothesis (one lane) of the fp64 filter: K @ Rt rows, guard-band constants, and the
// verdict for one point record.  RECT: t_right[1] == t_right[2] == 0 (rectified stereo, the
// KITTI case): both cameras share hz, hy and the threshold term, and one max() serves both.
struct Lane64 {
    double rt[12];
    double px[4], py[4];
    double cr0, cr1, cr2;
    double S, kq, kq2, thr;
    bool filter_ok;

    __device__ __forceinline__ void setup(const ScoreArgs &a, const double *src, double Amax, double Umax)
    {
#pragma unroll
        for (int c = 0; c < 12; c += 2) {
            const double2 v = *reinterpret_cast<const double2 *>(src + c);
            rt[c] = v.x;
            rt[c + 1] = v.y;
        }
        const double fx = a.K.m[0], fy = a.K.m[4], cx = a.K.m[2], cy = a.K.m[5];
        thr = a.thresh;
        double Bh = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                Bh = fmax(Bh, fabs(rt[4 * r + c]));
        // hz is trusted only above 2^-20 * (magnitude bound of the terms it was summed from)
        S = (Bh * Amax + (fmax(fmax(fabs(rt[3]), fabs(rt[7])), fabs(rt[11])) + a.trmax)) * 9.5367431640625e-07;
        // guard band: kappa = 2^-25 * mag / thresh + 2^-26, folded with the threshold into
        //   "in"  <=>  s * kq  < hz^2      (kq  = (1+kappa)/(1-kappa) / thresh^2)
        //   "out" <=>  s > hz^2 * kq2      (kq2 = (1+kappa)/(1-kappa) * thresh^2)
        const double mag = fabs(fx) + fabs(fy) + fabs(cx) + fabs(cy) + Umax + fabs(thr);
        const double kappa = 2.98023223876953125e-8 * (mag / thr) + 1.490116119384765625e-8;
        const double kp = (1.0 + kappa) / (1.0 - kappa);
        kq = kp / (thr * thr);
        kq2 = kp * (thr * thr);
        // Anything that makes the filter unsafe sends every evaluation of this lane to supports():
        // NaN/Inf in the pose (fmax drops NaNs) or the bounds, thresh <= 0 or NaN, a band that is not small.
        filter_ok = (thr > 0.0) && (kappa < 0.25) && (kq > 0.0) && (kq < __builtin_inf()) && (kq2 > 0.0) &&
                    (kq2 < __builtin_inf()) && (S < __builtin_inf());
#pragma unroll
        for (int c = 0; c < 12; ++c)
            filter_ok = filter_ok && (fabs(rt[c]) < __builtin_inf());
        // homogeneous pixel rows of K @ Rt for the pinhole K: hx = fx*L0 + cx*L2, hy = fy*L1 + cy*L2, hz = L2
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            px[c] = __builtin_fma(fx, rt[c], cx * rt[8 + c]);
            py[c] = __builtin_fma(fy, rt[4 + c], cy * rt[8 + c]);
        }
        // the right camera adds a constant to each homogeneous coordinate
        cr0 = __builtin_fma(fx, a.tr.v[0], cx * a.tr.v[2]);
        cr1 = __builtin_fma(fy, a.tr.v[1], cy * a.tr.v[2]);
        cr2 = a.tr.v[2];
    }

    template <bool RECT>
    __device__ __forceinline__ bool eval(const ScoreArgs &a, const double (&p)[SCORE_REC]) const
    {
        const double X = p[0], Y = p[1], Z = p[2], uL = p[3], vL = p[4], uR = p[5], vR = p[6];
        // filter: fused arithmetic, no division, no square root
        const double hx = __builtin_fma(px[0], X, __builtin_fma(px[1], Y, __builtin_fma(px[2], Z, px[3])));
        const double hy = __builtin_fma(py[0], X, __builtin_fma(py[1], Y, __builtin_fma(py[2], Z, py[3])));
        const double hz = __builtin_fma(rt[8], X, __builtin_fma(rt[9], Y, __builtin_fma(rt[10], Z, rt[11])));
        const double hzr = RECT ? hz : hz + cr2;
        const double hyr = RECT ? hy : hy + cr1;
        const double dxl = __builtin_fma(uL, hz, -hx);
        const double dyl = __builtin_fma(vL, hz, -hy);
        const double dxr = __builtin_fma(uR, hzr, -(hx + cr0));
        const double dyr = __builtin_fma(vR, hzr, -hyr);
        const double sl = __builtin_fma(dxl, dxl, dyl * dyl);
        const double sr = __builtin_fma(dxr, dxr, dyr * dyr);
        const double hz2 = hz * hz;
        bool unsafe, sure_in, sure_out;
        if (RECT) {
            const double m = fmax(sl, sr);   // drops a NaN operand: the unordered test below catches it
            unsafe = !(fabs(hz) > S) || __builtin_isunordered(sl, sr);
            sure_in = m * kq < hz2;
            sure_out = m > hz2 * kq2;
        } else {
            const double hzr2 = hzr * hzr;
            unsafe = !(fabs(hz) > S) || !(fabs(hzr) > S);
            sure_in = (sl * kq < hz2) && (sr * kq < hzr2);
            sure_out = (sl > hz2 * kq2) || (sr > hzr2 * kq2);
        }
        bool s = sure_in;
        if (unsafe || !filter_ok || !(sure_in || sure_out))
            s = supports(rt, a.K, a.tr, X, Y, Z, uL, vL, uR, vR, thr);   // float32 -> fp64 widening is exact
        return s;
    }
};

template <bool RECT>
__global__ void __launch_bounds__(256) score_fast_kernel(const ScoreArgs a)
{
    const int set = blockIdx.x / a.blocks_per_set;
    const int blk = blockIdx.x - set * a.blocks_per_set;
    const int h = blk * 256 + threadIdx.x;
    if ((blk * 256 + (int)(threadIdx.x & ~63u)) >= a.H || (a.skip && a.skip[set]))
        return;  // whole wavefront past the last hypothesis, or a set that is not to be scored
    const bool live = h < a.H;
    const double *hdr = score_hdr(a) + SCORE_HDR * set;
    Lane64 L;
    L.setup(a, a.Rt + ((size_t)set * a.Hs + (live ? h : a.H - 1)) * 12, hdr[0], hdr[1]);

    int n = a.n_dev ? a.n_dev[set] : a.n_host;
    n = min(max(n, 0), a.cap);
    n = __builtin_amdgcn_readfirstlane(n);
    const double *__restrict__ recs = score_recs64(a) + (size_t)SCORE_REC * ((size_t)set * a.cap);
    int cnt = 0;

    auto load = [&](double (&p)[SCORE_REC], int i) {
        const double *r = recs + (size_t)SCORE_REC * min(i, n - 1);
#pragma unroll
        for (int c = 0; c < SCORE_REC; ++c)
            p[c] = r[c];
    };

    // Records arrive through the scalar cache (the loop is wave-uniform), two points per
    // step and one step ahead: a scalar wait drains every outstanding scalar load, so the
    // loads of the NEXT pair are issued right after the wait for the current one.
    if (n > 0) {
        double a0[SCORE_REC], a1[SCORE_REC], b0[SCORE_REC], b1[SCORE_REC];
        load(a0, 0);
        load(a1, 1);
        int i = 0;
        for (; i + 4 <= n; i += 4) {
            load(b0, i + 2);
            load(b1, i + 3);
            cnt += L.eval<RECT>(a, a0) ? 1 : 0;
            cnt += L.eval<RECT>(a, a1) ? 1 : 0;
            load(a0, i + 4);
            load(a1, i + 5);
            cnt += L.eval<RECT>(a, b0) ? 1 : 0;
            cnt += L.eval<RECT>(a, b1) ? 1 : 0;
        }
        for (; i < n; ++i) {
            cnt += L.eval<RECT>(a, a0) ? 1 : 0;
#pragma unroll
            for (int c = 0; c < SCORE_REC; ++c)
                a0[c] = a1[c];
            load(a1, i + 2);
        }
    }
    if (live)
        a.counts[(size_t)set * a.Hs + h] = cnt;
}

// Mixed-precision kernel for rectified stereo: lane = hypothesis, points through scalar
// loads, and a first-level filter in fp32 (twice the fp64 VALU rate) in front of the fp64
// filter above.  Everything is scaled to threshold units with the principal point removed:
//     dx = uc*hz - gx,  dy = vc*hz - gy,   uc = (u-cx)/thr,  gx = (fx/thr)*L0,  hz = L2
//     supporter  <=>  a := max over cameras of sqrt(dx^2+dy^2)  <  b := |hz|
// which is the reference's test multiplied through by |hz|/thr.  With u = 2^-24 the fp32
// evaluation (inputs rounded once, fma chains) obeys
//     |a^ - a| + |b^ - b|  <=  G + 2.01u (a^ + b^),    G = u * Wb * (Brot*A + T),
//     Wb = 8.63*Wmax + 5.2,  A = |X|+|Y|+|Z|,  Brot = max |R_ij|,  T = max |t_i| + max |t_right,i|
// (5 roundings per term of an fma chain, sqrt(2) for the two coordinates, the slack of the
// fp64 reference evaluation itself folded into the constants).  The lane is "sure" when
//     |m^ - w^|  >  c1*(m^ + w^) + ca*A2 + cb
// with m^ = max(sl, sr), w^ = hz^2, c1 = tau/2 + 4.03u, ca*A2 + cb >= G^2/tau and
// tau = 2*u*Wb*Brot (the arithmetic-geometric-mean split of G*(a^+b^)); all four inflated by
// 1e-3 against their own fp32 rounding.  A sure lane's verdict is the sign of m^ - w^; an
// unsure point (about 4 in 10 000 evaluations on KITTI-like data) is parked in a per-lane
// queue and settled by Lane64::eval after the loop.  "Sure in" is impossible when |hz| is below the rounding noise of its own
// sum, and "sure out" there needs a projected pixel hundreds of thresholds away, so the
// division in supports() cannot disagree.
//
// SPLIT: for few hypotheses per set (the first slice of the batched RANSAC) a block takes 64
// hypotheses and its four wavefronts a quarter of the points each, summed through LDS at the
// end -- four times the wavefronts in flight for the same work.
template <bool SPLIT>
__global__ void __launch_bounds__(256) score_mixed_kernel(const ScoreArgs a)
{
    const int set = blockIdx.x / a.blocks_per_set;
    const int blk = blockIdx.x - set * a.blocks_per_set;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int h = SPLIT ? blk * 64 + (int)(threadIdx.x & 63u) : blk * 256 + (int)threadIdx.x;
    if ((SPLIT ? blk * 64 : blk * 256 + 64 * wave) >= a.H || (a.skip && a.skip[set]))
        return;  // whole wavefront (SPLIT: block) past the last hypothesis, or a set that is not to be scored
    const bool live = h < a.H;
    const double *hdr = score_hdr(a) + SCORE_HDR * set;
    const double Amax = hdr[0], Umax = hdr[1], Wmax = hdr[2];
    const double *pose = a.Rt + ((size_t)set * a.Hs + (live ? h : a.H - 1)) * 12;

    // ---- fp32 lane constants ----------------------------------------------------------
    // The fp64 lane state is only needed to settle undecided points after the loop; it is
    // rebuilt there so that it does not occupy registers (and cost occupancy) during the loop.
    float qx[4], qy[4], qz[4], crx, c1, ca, cb;
    bool f32_ok, dead;
    {
        Lane64 L;
        L.setup(a, pose, Amax, Umax);
        const double fx = a.K.m[0], fy = a.K.m[4], cx = a.K.m[2], cy = a.K.m[5], thr = a.thresh;
        const double sfx = fx / thr, sfy = fy / thr;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            qx[c] = (float)(sfx * L.rt[c]);
            qy[c] = (float)(sfy * L.rt[4 + c]);
            qz[c] = (float)L.rt[8 + c];
        }
        crx = (float)(sfx * a.tr.v[0]);
        double Brot = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                Brot = fmax(Brot, fabs(L.rt[4 * r + c]));
        const double T = fmax(fmax(fabs(L.rt[3]), fabs(L.rt[7])), fabs(L.rt[11])) + a.trmax;
        const double u24 = 5.9604644775390625e-08;
        const double Wb = 8.63 * Wmax + 5.2;
        const double tau = 2.0 * u24 * Wb * Brot;
        const double c1d = (0.5 * tau + 4.03 * u24) * 1.001;
        const double cad = (2.0 * u24 * u24 * Wb * Wb * Brot * Brot / tau) * 1.001;
        const double cbd = (2.0 * u24 * u24 * Wb * Wb * T * T / tau) * 1.001 + 1e-30;
        c1 = (float)c1d;
        ca = (float)cad;
        // The fp32 stage needs a sane camera and finite, moderate magnitudes; otherwise every
        // point of this lane goes to the fp64 path.
        const double mag = fabs(fx) + fabs(fy) + fabs(cx) + fabs(cy) + Umax + fabs(thr);
        f32_ok = L.filter_ok && (mag / thr < 262144.0) && (Wmax >= 0.015625) && (Wmax < 262144.0) &&
                 (fabs(cx) <= 64.0 * thr * Wmax) && (fabs(cy) <= 64.0 * thr * Wmax) && (Brot > 1e-30) &&
                 (tau < 0.125) && (cbd < 1e30) && (cad < 1e30) && (cad > 1e-30) && (Amax < 1e18) && (T < 1e18);
        cb = f32_ok ? (float)cbd : __builtin_nanf("");   // a NaN bound fails every "sure" test
        // A pose with a NaN or Inf entry supports nothing: with the pinhole K the depth
        // hz = (0*L0 + 0*L1) + L2 of supports() is then NaN or Inf, and so is every projection.
        // (Failed minimal solves arrive as NaN poses; they must not drag their wavefront into
        // the fp64 recount.)
        dead = false;
#pragma unroll
        for (int c = 0; c < 12; ++c)
            dead = dead || !(fabs(L.rt[c]) < __builtin_inf());
    }

    int n = a.n_dev ? a.n_dev[set] : a.n_host;
    n = min(max(n, 0), a.cap);
    n = __builtin_amdgcn_readfirstlane(n);
    // this wavefront's points: pairs [pair0, pair0 + n_pairs), and the odd last point for its owner
    int pair0 = 0, n_pairs = n >> 1;
    if (SPLIT) {
        const int q = ((n >> 1) + 3) >> 2;
        pair0 = min(wave * q, n >> 1);
        n_pairs = min(q, (n >> 1) - pair0);
    }
    const bool owns_tail = !SPLIT || wave == 3;
    const int pt0 = 2 * pair0, pt_end = owns_tail ? n : pt0 + 2 * n_pairs;
    const float *__restrict__ recs32 =
        score_recs32(a) + (size_t)SCORE_REC32 * ((size_t)set * a.cap) + (size_t)SCORE_REC32 * pt0;
    const double *__restrict__ recs64 = score_recs64(a) + (size_t)SCORE_REC * ((size_t)set * a.cap);
    int cnt = 0;

    // d = m - w (sign = verdict) and whether the lane is sure of it
    auto eval32 = [&](const float *p, float &d, bool &sure) __attribute__((always_inline)) {
        const float X = p[0], Y = p[1], Z = p[2], ul = p[3], vl = p[4], ur = p[5], vr = p[6], A2 = p[7];
        const float hz = __builtin_fmaf(qz[0], X, __builtin_fmaf(qz[1], Y, __builtin_fmaf(qz[2], Z, qz[3])));
        const float gx = __builtin_fmaf(qx[0], X, __builtin_fmaf(qx[1], Y, __builtin_fmaf(qx[2], Z, qx[3])));
        const float gy = __builtin_fmaf(qy[0], X, __builtin_fmaf(qy[1], Y, __builtin_fmaf(qy[2], Z, qy[3])));
        const float gxr = gx + crx;
        const float dxl = __builtin_fmaf(ul, hz, -gx);
        const float dyl = __builtin_fmaf(vl, hz, -gy);
        const float dxr = __builtin_fmaf(ur, hz, -gxr);
        const float dyr = __builtin_fmaf(vr, hz, -gy);
        const float sl = __builtin_fmaf(dxl, dxl, dyl * dyl);
        const float sr = __builtin_fmaf(dxr, dxr, dyr * dyr);
        const float m = fmaxf(sl, sr);   // non-finite point data never gets here (bounds poison f32_ok)
        const float w = hz * hz;
        d = m - w;
        const float e = m + w;
        const float r = __builtin_fmaf(c1, e, __builtin_fmaf(ca, A2, cb));
        sure = fabsf(d) > r;
    };
    auto load = [&](float (&p)[2 * SCORE_REC32], int pair, int n_pairs) __attribute__((always_inline)) {
        const float *r = recs32 + (size_t)(2 * SCORE_REC32) * min(pair, n_pairs - 1);
#pragma unroll
        for (int c = 0; c < 2 * SCORE_REC32; ++c)
            p[c] = r[c];
    };
    // Verdict bits are shifted into `bits` (newest in bit 0) and counted 32 at a time.  A point
    // the lane is not sure of contributes a 0 bit and its index goes to the lane's queue in LDS;
    // the queues are settled in fp64 after the loop.  A lane whose queue overflows (or whose fp32
    // stage is disabled) recounts every point in fp64 instead.
    __shared__ unsigned queue[SCORE_QUEUE][256];
    unsigned bits = 0;
    int qn = 0;
    bool full = !f32_ok && !dead;
    auto push = [&](int i) __attribute__((always_inline)) {
        if (dead)
            return;
        if (!full && qn < SCORE_QUEUE)
            queue[qn++][threadIdx.x] = (unsigned)i;
        else
            full = true;
    };
    auto step = [&](const float (&p)[2 * SCORE_REC32], int pair) __attribute__((always_inline)) {
        float d0, d1;
        bool s0, s1;
        eval32(p, d0, s0);
        eval32(p + SCORE_REC32, d1, s1);
        bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(d0), 31);
        bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(d1), 31);
        if (__builtin_expect(__ballot(!(s0 && s1)) != 0ull, 0)) {
            if (pair < n_pairs) {   // the clamped duplicate of an odd pair count is dropped below
                if (!s0) {
                    bits &= ~2u;
                    push(pt0 + 2 * pair);
                }
                if (!s1) {
                    bits &= ~1u;
                    push(pt0 + 2 * pair + 1);
                }
            }
        }
    };

    // Pairs of points, two pairs per trip (ping-pong scalar prefetch: a scalar wait drains every
    // outstanding scalar load, so the loads of the next pair are issued right after the wait for
    // the current one), 16 pairs = 32 verdict bits per popcount.  An odd pair count runs one
    // clamped duplicate whose two bits are dropped.
    const int n_trips = (n_pairs + 1) & ~1;
    if (n_pairs > 0 && __ballot(!full) != 0ull) {
        float pa[2 * SCORE_REC32], pb[2 * SCORE_REC32];
        load(pa, 0, n_pairs);
        for (int g = 0; g < n_trips; g += 16) {
            const int ge = min(g + 16, n_trips);
            for (int j = g; j < ge; j += 2) {
                load(pb, j + 1, n_pairs);
                step(pa, j);
                load(pa, j + 2, n_pairs);
                step(pb, j + 1);
            }
            if (ge == n_trips && (n_pairs & 1))
                bits &= ~3u;
            cnt += __popc(bits);
            bits = 0;
        }
    }
    if ((n & 1) && owns_tail) {
        if (__ballot(!full) != 0ull) {
            // odd point count: the last point alone through the fp32 stage
            const float *r = recs32 + (size_t)SCORE_REC32 * (n - 1 - pt0);
            float p1[SCORE_REC32];
#pragma unroll
            for (int c = 0; c < SCORE_REC32; ++c)
                p1[c] = r[c];
            float d;
            bool sure;
            eval32(p1, d, sure);
            if (sure)
                cnt += d < 0.0f ? 1 : 0;
            else
                push(n - 1);
        } else {
            push(n - 1);
        }
    }
    // settle the queues: lane-private point indices, fp64 filter / reference evaluation
    // (the reference evaluation itself: only the pose has to be live here, which keeps the
    // kernel at 8 wavefronts per SIMD without spills)
    // (only lanes that have something to settle read their pose again)
    double rt[12];
#pragma unroll
    for (int c = 0; c < 12; ++c)
        rt[c] = 0.0;
    if (qn > 0 || full) {
#pragma unroll
        for (int c = 0; c < 12; c += 2) {
            const double2 v = *reinterpret_cast<const double2 *>(pose + c);
            rt[c] = v.x;
            rt[c + 1] = v.y;
        }
    }
    auto eval64 = [&](int i) __attribute__((always_inline)) -> bool {
        const double *r = recs64 + (size_t)SCORE_REC * i;
        return supports(rt, a.K, a.tr, r[0], r[1], r[2], r[3], r[4], r[5], r[6], a.thresh);
    };
    for (int q = 0; __ballot(q < qn) != 0ull; ++q) {
        if (q < qn)
            cnt += eval64((int)queue[q][threadIdx.x]) ? 1 : 0;
    }
    if (__ballot(full) != 0ull) {
        int cnt64 = 0;
        for (int i = pt0; i < pt_end; ++i)
            cnt64 += eval64(i) ? 1 : 0;
        if (full)
            cnt = cnt64;
    }
    if (dead)
        cnt = 0;
    if (SPLIT) {
        __shared__ int quarter[4][64];
        quarter[wave][threadIdx.x & 63u] = cnt;
        __syncthreads();
        if (wave == 0 && live)
            a.counts[(size_t)set * a.Hs + h] =
                (quarter[0][threadIdx.x] + quarter[1][threadIdx.x]) + (quarter[2][threadIdx.x] + quarter[3][threadIdx.x]);
    } else if (live) {
        a.counts[(size_t)set * a.Hs + h] = cnt;
    }
}


// Matrix-core form of the fp32 first level for rectified stereo: one wavefront scores 32 hypotheses
// against all points of its set, 32 points at a time.  The rigid transform of a 32 x 32 tile
// (hypothesis i, point j) -- hz, gx, gy = q . [X Y Z 1] -- and the point-dependent part of the guard
// band, ca_i*A2_j + cb_i, are seven v_mfma_f32_32x32x2_f32 (exact fp32: each is bit for bit
// fma(a1,b1, fma(a0,b0, c)), tools/score_mfma_proxy.hip); their results arrive with lane = point
// column and register = hypothesis row, so the four pixel coordinates of a point are per-lane
// values loaded once per tile, and what is left for the vector ALU is 16 instructions per
// evaluation (4 fma residuals, 2 mul + 2 fma squares, add, max, mul, sub, add, fma, cmp, alignbit)
// instead of 26.  The MFMAs of tile t+1 are issued between the evaluations of tile t (two
// accumulator sets), the point records arrive through a ring of four register buffers loaded two
// tiles ahead.
//
// NOT the default, and slower than score_mixed_kernel (1.16 against 0.91 ms on 127 sets x 10 000
// hypotheses x 1190 points): on gfx950 an fp32 MFMA does not run beside fp32 vector instructions,
// it takes their place -- a wavefront's MFMA plus N v_fma_f32 cost the sum of the two
// (tools/mfma_f32_coexec.hip; SQ_VALU_MFMA_COEXEC_CYCLES of this kernel is 0), and seven MFMAs per
// tile cost slightly more vector-pipe time than the 160 fma they replace.  Kept selectable
// (SLAM_SCORE_MFMA) as a cross-check with identical counts and so that the measurement can be repeated
// (tools/ab_score.py).
//
// Verdicts are the sign bits of d = max(sl, sr) - hz^2, shifted into one word per (lane, register)
// and counted 32 tiles at a time; an evaluation the filter is not sure of (|d| <= r) gets a 0 bit
// and is parked as (row, point) in the wavefront's LDS queue, settled with supports() after the loop.
//
// Error bound, as for score_mixed_kernel but for this accumulation order: a term of hz passes
// through at most four fused operations (fma(q1,Y, q0*X), then fma(q3,1, fma(q2,Z, .))) after the
// two input roundings, six roundings in all; with the roundings of the residuals (one more on the
// pixel coordinate, two more for the right camera's offset and its sum)
//     |a^ - a| + |b^ - b|  <=  G + 2.01u (a^ + b^),   G = u * Wb * (Brot*A + T),   Wb = 11.5*Wmax + 6.2
// (11.34 and 6.01 by the count above, the rest is slack).  The split of G*(a^+b^) into
// tau/2*(a^+b^)^2 + G^2/(2 tau) holds for every tau > 0: here tau = 2*u*Wb is one value per set, so
// c1 = tau/2 + 4.03u is a scalar, and ca = u*Wb*Brot^2, cb = u*Wb*T^2 carry the hypothesis.
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int MFMA_QUEUE = 512;   // parked evaluations per wavefront before it recounts everything in fp64

struct MfmaTile {
    f16v Z, X, Y, R;
};

struct MfmaOps {       // hypothesis-side MFMA operands of a lane (row = lane % 32, k = lane / 32)
    float z0, z1, x0, x1, y0, y1, r;
};

// Evaluations V0 .. V0+3 of a tile held by this lane, in two halves written stage by stage across
// the four (at two wavefronts per SIMD a dependent instruction must not follow its producer
// directly: the four evaluations are independent chains); the MFMA `MID` is issued between the
// halves and `END` after the second.  `any` collects the "not sure" lane masks.
#define SLAM_MFMA_EACH4 _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)
#define SLAM_MFMA_EVAL4(V0, MID, END)                                                                        \
    {                                                                                                        \
        float gxr_[4], dxl_[4], dyl_[4], dxr_[4], dyr_[4], t1_[4], t2_[4], sl_[4], sr_[4], m_[4], w_[4],     \
            d_[4], e_[4], r_[4];                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        SLAM_MFMA_EACH4 gxr_[i_] = cur.X[V0 + i_] + crx;                                                     \
        SLAM_MFMA_EACH4 dxl_[i_] = __builtin_fmaf(ul, cur.Z[V0 + i_], -cur.X[V0 + i_]);                      \
        SLAM_MFMA_EACH4 dyl_[i_] = __builtin_fmaf(vl, cur.Z[V0 + i_], -cur.Y[V0 + i_]);                      \
        SLAM_MFMA_EACH4 dyr_[i_] = __builtin_fmaf(vr, cur.Z[V0 + i_], -cur.Y[V0 + i_]);                      \
        SLAM_MFMA_EACH4 dxr_[i_] = __builtin_fmaf(ur, cur.Z[V0 + i_], -gxr_[i_]);                            \
        SLAM_MFMA_EACH4 t1_[i_] = dyl_[i_] * dyl_[i_];                                                       \
        SLAM_MFMA_EACH4 t2_[i_] = dyr_[i_] * dyr_[i_];                                                       \
        SLAM_MFMA_EACH4 sl_[i_] = __builtin_fmaf(dxl_[i_], dxl_[i_], t1_[i_]);                               \
        MID;                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        SLAM_MFMA_EACH4 sr_[i_] = __builtin_fmaf(dxr_[i_], dxr_[i_], t2_[i_]);                               \
        SLAM_MFMA_EACH4 w_[i_] = cur.Z[V0 + i_] * cur.Z[V0 + i_];                                            \
        SLAM_MFMA_EACH4 m_[i_] = fmaxf(sl_[i_], sr_[i_]);                                                    \
        SLAM_MFMA_EACH4 d_[i_] = m_[i_] - w_[i_];                                                            \
        SLAM_MFMA_EACH4 e_[i_] = m_[i_] + w_[i_];                                                            \
        SLAM_MFMA_EACH4 bits[V0 + i_] = __builtin_amdgcn_alignbit(bits[V0 + i_], __float_as_uint(d_[i_]), 31); \
        SLAM_MFMA_EACH4 r_[i_] = __builtin_fmaf(c1, e_[i_], cur.R[V0 + i_]);                                 \
        SLAM_MFMA_EACH4 um[V0 + i_] = __ballot(!(fabsf(d_[i_]) > r_[i_]));                                   \
        SLAM_MFMA_EACH4 any |= um[V0 + i_];                                                                  \
        END;                                                                                                 \
    }

// One step of the scoring loop: the 16 evaluations of tile `cur` with the seven MFMAs of the next
// tile placed between them, one after every 32 vector instructions (an MFMA holds the matrix pipe
// for 16 passes but the issue port for one; the scheduler is kept from gathering them).
__device__ __forceinline__ unsigned long long mfma_step(const MfmaTile &cur, MfmaTile &nxt, const MfmaOps &o, float b0,
                                                        float b1, float b2, float ul, float vl, float ur, float vr,
                                                        float crx, float c1, unsigned (&bits)[16],
                                                        unsigned long long (&um)[16])
{
    f16v zero;
#pragma unroll
    for (int v = 0; v < 16; ++v)
        zero[v] = 0.0f;
    unsigned long long any = 0ull;
    SLAM_MFMA_EVAL4(0, nxt.Z = __builtin_amdgcn_mfma_f32_32x32x2f32(o.z0, b0, zero, 0, 0, 0),
                    nxt.X = __builtin_amdgcn_mfma_f32_32x32x2f32(o.x0, b0, zero, 0, 0, 0))
    SLAM_MFMA_EVAL4(4, nxt.Y = __builtin_amdgcn_mfma_f32_32x32x2f32(o.y0, b0, zero, 0, 0, 0),
                    nxt.R = __builtin_amdgcn_mfma_f32_32x32x2f32(o.r, b2, zero, 0, 0, 0))
    SLAM_MFMA_EVAL4(8, nxt.Z = __builtin_amdgcn_mfma_f32_32x32x2f32(o.z1, b1, nxt.Z, 0, 0, 0),
                    nxt.X = __builtin_amdgcn_mfma_f32_32x32x2f32(o.x1, b1, nxt.X, 0, 0, 0))
    SLAM_MFMA_EVAL4(12, nxt.Y = __builtin_amdgcn_mfma_f32_32x32x2f32(o.y1, b1, nxt.Y, 0, 0, 0), (void)0)
    __builtin_amdgcn_sched_barrier(0);
    return any;
}

__device__ __forceinline__ void mfma_first(MfmaTile &t, const MfmaOps &o, float b0, float b1, float b2)
{
    f16v zero;
#pragma unroll
    for (int v = 0; v < 16; ++v)
        zero[v] = 0.0f;
    t.Z = __builtin_amdgcn_mfma_f32_32x32x2f32(o.z0, b0, zero, 0, 0, 0);
    t.X = __builtin_amdgcn_mfma_f32_32x32x2f32(o.x0, b0, zero, 0, 0, 0);
    t.Y = __builtin_amdgcn_mfma_f32_32x32x2f32(o.y0, b0, zero, 0, 0, 0);
    t.R = __builtin_amdgcn_mfma_f32_32x32x2f32(o.r, b2, zero, 0, 0, 0);
    t.Z = __builtin_amdgcn_mfma_f32_32x32x2f32(o.z1, b1, t.Z, 0, 0, 0);
    t.X = __builtin_amdgcn_mfma_f32_32x32x2f32(o.x1, b1, t.X, 0, 0, 0);
    t.Y = __builtin_amdgcn_mfma_f32_32x32x2f32(o.y1, b1, t.Y, 0, 0, 0);
}

__global__ void __launch_bounds__(256, 2) score_mfma_kernel(const ScoreArgs a)
{
    __shared__ unsigned queue[4][MFMA_QUEUE];
    __shared__ int rowcnt[4][32];
    __shared__ unsigned tally[4][16][64];
    const int set = blockIdx.x / a.blocks_per_set;
    const int blk = blockIdx.x - set * a.blocks_per_set;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63u), col = lane & 31;
    const bool hi = lane >= 32;
    const int h_base = blk * 128 + wave * 32;
    if (h_base >= a.H || (a.skip && a.skip[set]))
        return;   // no block-wide barrier below: every wavefront is on its own
    const int h = h_base + col;
    const bool live = h < a.H;
    const double *hdr = score_hdr(a) + SCORE_HDR * set;
    const double Amax = hdr[0], Umax = hdr[1], Wmax = hdr[2];
    const double *pose = a.Rt + ((size_t)set * a.Hs + (live ? h : a.H - 1)) * 12;

    // ---- per-set constants and this lane's hypothesis (row = col) -------------------------------
    const double fx = a.K.m[0], fy = a.K.m[4], cx = a.K.m[2], cy = a.K.m[5], thr = a.thresh;
    const double sfx = fx / thr, sfy = fy / thr;
    const double u24 = 5.9604644775390625e-08;
    const double Wb = 11.5 * Wmax + 6.2;
    const double tau = 2.0 * u24 * Wb;
    const float c1 = (float)((0.5 * tau + 4.03 * u24) * 1.001);
    const float crx = (float)(sfx * a.tr.v[0]);
    const double mag = fabs(fx) + fabs(fy) + fabs(cx) + fabs(cy) + Umax + fabs(thr);
    const bool set_ok = (thr > 0.0) && (mag / thr < 262144.0) && (Wmax >= 0.015625) && (Wmax < 262144.0) &&
                        (fabs(cx) <= 64.0 * thr * Wmax) && (fabs(cy) <= 64.0 * thr * Wmax) && (tau < 0.125) &&
                        (Amax < 1e18) && (fabs(sfx) < 1e18) && (fabs(sfy) < 1e18) && (fabs(a.tr.v[0]) < 1e18);
    double rt[12];
#pragma unroll
    for (int c = 0; c < 12; c += 2) {
        const double2 v = *reinterpret_cast<const double2 *>(pose + c);
        rt[c] = v.x;
        rt[c + 1] = v.y;
    }
    bool dead = !live;   // a pose with a NaN or Inf entry supports nothing (see score_mixed_kernel)
#pragma unroll
    for (int c = 0; c < 12; ++c)
        dead = dead || !(fabs(rt[c]) < __builtin_inf());
    double Brot = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            Brot = fmax(Brot, fabs(rt[4 * r + c]));
    const double T = fmax(fmax(fabs(rt[3]), fabs(rt[7])), fabs(rt[11])) + a.trmax;
    const double cad = u24 * Wb * Brot * Brot * 1.001;
    const double cbd = u24 * Wb * T * T * 1.001 + 1e-30;
    const bool ok = set_ok && (T < 1e18) && (Brot < 1e18) && (cad < 1e30) && (cbd < 1e30);
    // rows that take no part in the fp32 stage (dead, or to be recounted in fp64) get operands that
    // keep them quiet: everything 0 and a negative bound, so that they are always "sure" and "out"
    const bool quiet = dead || !ok;
    MfmaOps ops;
    ops.z0 = quiet ? 0.0f : (float)(hi ? rt[9] : rt[8]);
    ops.z1 = quiet ? 0.0f : (float)(hi ? rt[11] : rt[10]);
    ops.x0 = quiet ? 0.0f : (float)(sfx * (hi ? rt[1] : rt[0]));
    ops.x1 = quiet ? 0.0f : (float)(sfx * (hi ? rt[3] : rt[2]));
    ops.y0 = quiet ? 0.0f : (float)(sfy * (hi ? rt[5] : rt[4]));
    ops.y1 = quiet ? 0.0f : (float)(sfy * (hi ? rt[7] : rt[6]));
    ops.r = quiet ? (hi ? -1.0f : 0.0f) : (float)(hi ? cbd : cad);
    const unsigned dead_rows = (unsigned)__ballot(dead);                 // bit i = row i (lanes 0..31)
    unsigned full_rows = (unsigned)__ballot(!dead && !ok);               // rows recounted in fp64

    int n = a.n_dev ? a.n_dev[set] : a.n_host;
    n = min(max(n, 0), a.cap);
    n = __builtin_amdgcn_readfirstlane(n);
    const float4 *__restrict__ recs32 =
        reinterpret_cast<const float4 *>(score_recs32(a) + (size_t)SCORE_REC32 * ((size_t)set * a.cap));
    const double *__restrict__ recs64 = score_recs64(a) + (size_t)SCORE_REC * ((size_t)set * a.cap);
    const int n_tiles = (n + 31) >> 5;

    unsigned bits[16];
    unsigned long long um[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        bits[v] = 0u;
        tally[wave][v][lane] = 0u;
    }
    int qn = 0;   // wave-uniform number of parked evaluations (may exceed MFMA_QUEUE: then everything is recounted)

    auto load = [&](int tile, float4 &ra, float4 &rb) __attribute__((always_inline)) {
        const float4 *p = recs32 + 2 * (size_t)min(tile * 32 + col, n - 1);
        ra = p[0];
        rb = p[1];
    };
    // park the evaluations of tile `tile` the filter was not sure of
    auto park = [&](int tile, unsigned long long valid) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const unsigned long long m = um[v] & valid;
            if (m != 0ull) {
                const int row = 8 * (v >> 2) + (v & 3) + (hi ? 4 : 0);
                if ((m >> lane) & 1ull) {
                    bits[v] &= ~1u;
                    const int slot = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                          __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (slot < MFMA_QUEUE)
                        queue[wave][slot] = ((unsigned)(tile * 32 + col) << 5) | (unsigned)row;
                }
                qn += __popcll(m);
            }
        }
    };
    auto finish = [&](int tile, unsigned long long any) __attribute__((always_inline)) {
        const int left = n - tile * 32;                       // points of this tile
        if (__builtin_expect(left < 32, 0)) {
            // the last, partial tile: lanes past the end evaluated a clamped duplicate; drop their bits
            const unsigned long long half_mask = (1ull << left) - 1ull;
            const unsigned long long valid = half_mask | (half_mask << 32);
            if (!((valid >> lane) & 1ull)) {
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    bits[v] &= ~1u;
            }
            if (any & valid)
                park(tile, valid);
        } else if (__builtin_expect(any != 0ull, 0)) {
            park(tile, ~0ull);
        }
        if ((tile & 31) == 31) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                tally[wave][v][lane] += __popc(bits[v]);
                bits[v] = 0u;
            }
        }
    };

    if (n_tiles > 0 && (~(dead_rows | full_rows)) != 0u) {
        MfmaTile t0, t1;
        float4 a0, b0, a1, b1, a2, b2, a3, b3;
        load(0, a0, b0);
        load(1, a1, b1);
        mfma_first(t0, ops, hi ? a0.y : a0.x, hi ? 1.0f : a0.z, hi ? 1.0f : b0.w);
        int t = 0;
        // four tiles per trip so that every buffer index is static: tile t+u uses record buffer u and
        // accumulator set u & 1; the loads of tile t+u+2 and the MFMAs of tile t+u+1 go out first
#define SLAM_MFMA_STEP(U, TCUR, RA, RB, TNEXT, NA, NB, LA, LB)                                                      \
        if (t + U < n_tiles) {                                                                                         \
            load(t + U + 2, LA, LB);                                                                                   \
            const unsigned long long any_ =                                                                            \
                mfma_step(TCUR, TNEXT, ops, hi ? NA.y : NA.x, hi ? 1.0f : NA.z, hi ? 1.0f : NB.w, RA.w, RB.x, RB.y,    \
                          RB.z, crx, c1, bits, um);                                                                    \
            finish(t + U, any_);                                                                                       \
        }
        for (; t < n_tiles; t += 4) {
            SLAM_MFMA_STEP(0, t0, a0, b0, t1, a1, b1, a2, b2)
            SLAM_MFMA_STEP(1, t1, a1, b1, t0, a2, b2, a3, b3)
            SLAM_MFMA_STEP(2, t0, a2, b2, t1, a3, b3, a0, b0)
            SLAM_MFMA_STEP(3, t1, a3, b3, t0, a0, b0, a1, b1)
        }
#undef SLAM_MFMA_STEP
    }
    // ---- counts per (lane, register) -> counts per row -----------------------------------------
    if (lane < 32)
        rowcnt[wave][lane] = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        int c = (int)(tally[wave][v][lane] + __popc(bits[v]));
#pragma unroll
        for (int off = 16; off > 0; off >>= 1)
            c += __shfl_xor(c, off);
        if (col == 0)
            rowcnt[wave][8 * (v >> 2) + (v & 3) + (hi ? 4 : 0)] = c;
    }
    if (qn > MFMA_QUEUE)          // more undecided evaluations than the queue holds: fp64 for every live row
        full_rows = ~dead_rows;
    // ---- settle the queue: one parked (row, point) per lane and trip ------------------------------
    for (int q0 = 0; q0 < min(qn, MFMA_QUEUE); q0 += 64) {
        const int q = q0 + lane;
        if (q < min(qn, MFMA_QUEUE)) {
            const unsigned e = queue[wave][q];
            const int row = (int)(e & 31u), pt = (int)(e >> 5);
            if (!((full_rows >> row) & 1u)) {
                const double *ps = a.Rt + ((size_t)set * a.Hs + h_base + row) * 12;
                double prt[12];
#pragma unroll
                for (int c = 0; c < 12; ++c)
                    prt[c] = ps[c];
                const double *r = recs64 + (size_t)SCORE_REC * pt;
                if (supports(prt, a.K, a.tr, r[0], r[1], r[2], r[3], r[4], r[5], r[6], a.thresh))
                    atomicAdd(&rowcnt[wave][row], 1);
            }
        }
    }
    // ---- rows the fp32 stage could not take: all points in fp64, lane = point ------------------
    for (unsigned rows = full_rows; rows != 0u; rows &= rows - 1u) {
        const int row = __builtin_ctz(rows);
        const double *ps = a.Rt + ((size_t)set * a.Hs + min(h_base + row, a.H - 1)) * 12;
        double prt[12];
#pragma unroll
        for (int c = 0; c < 12; ++c)
            prt[c] = ps[c];
        int c64 = 0;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            const double *r = recs64 + (size_t)SCORE_REC * min(i, n - 1);
            const bool s = (i < n) && supports(prt, a.K, a.tr, r[0], r[1], r[2], r[3], r[4], r[5], r[6], a.thresh);
            c64 += __popcll(__ballot(s));
        }
        if (lane == 0)
            rowcnt[wave][row] = c64;
    }
    if (lane < 32 && live)
        a.counts[(size_t)set * a.Hs + h] = ((dead_rows >> lane) & 1u) ? 0 : rowcnt[wave][lane];
}

}  // namespace
}  // namespace slam

using namespace slam;

extern "C" int slam_score_hypotheses(const double *Rt, int n_sets, int H, const double *K, const double *t_right,
                                     const double *X, const double *Y, const double *Z, const float *uL,
                                     const float *vL, const float *uR, const float *vR, int cap, const int32_t *n_dev,
                                     int n_host, double thresh, int32_t *counts, uint64_t *masks, int mask_words,
                                     double *work, void *stream)
{
    return slam_score_hypotheses_range(Rt, n_sets, H, 0, H, nullptr, K, t_right, X, Y, Z, uL, vL, uR, vR, cap, n_dev,
                                       n_host, thresh, counts, masks, mask_words, work, 0, stream);
}

extern "C" int slam_score_hypotheses_range(const double *Rt, int n_sets, int H_total, int h_begin, int H,
                                           const int32_t *skip, const double *K, const double *t_right,
                                           const double *X, const double *Y, const double *Z, const float *uL,
                                           const float *vL, const float *uR, const float *vR, int cap,
                                           const int32_t *n_dev, int n_host, double thresh, int32_t *counts,
                                           uint64_t *masks, int mask_words, double *work, int work_ready,
                                           void *stream)
{
    if (n_sets < 0 || H < 0 || cap < 0 || h_begin < 0 || H_total < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_score_hypotheses: negative size");
    if ((long long)h_begin + H > H_total)
        return fail(SLAM_ERR_BAD_ARG, "slam_score_hypotheses_range: range [%d, %d) exceeds %d hypotheses per set",
                    h_begin, h_begin + H, H_total);
    if (masks && (h_begin != 0 || H != H_total || skip))
        return fail(SLAM_ERR_UNSUPPORTED, "slam_score_hypotheses_range: masks need the full range and no skip list");
    if (!K || !t_right)
        return fail(SLAM_ERR_BAD_ARG, "slam_score_hypotheses: null intrinsics");
    if (n_sets == 0 || H == 0)
        return SLAM_OK;
    if (!Rt || !counts)
        return fail(SLAM_ERR_BAD_ARG, "slam_score_hypotheses: null buffer");
    if (cap > 0 && (!X || !Y || !Z || !uL || !vL || !uR || !vR))
        return fail(SLAM_ERR_BAD_ARG, "slam_score_hypotheses: null point buffer");
    if (masks && mask_words < (cap + 63) / 64)
        return fail(SLAM_ERR_BAD_ARG, "slam_score_hypotheses: mask_words %d < ceil(cap/64) = %d", mask_words,
                    (cap + 63) / 64);
    if ((reinterpret_cast<uintptr_t>(Rt) | reinterpret_cast<uintptr_t>(work)) & 15)
        return fail(SLAM_ERR_BAD_ARG, "slam_score_hypotheses: Rt and work must be 16-byte aligned");
    ScoreArgs a;
    a.Rt = Rt ? Rt + (size_t)h_begin * 12 : Rt;
    a.H = H;
    a.Hs = H_total;
    a.skip = skip;
    for (int i = 0; i < 9; ++i)
        a.K.m[i] = K[i];
    a.trmax = 0.0;
    for (int i = 0; i < 3; ++i) {
        a.tr.v[i] = t_right[i];
        const double m = t_right[i] < 0 ? -t_right[i] : t_right[i];
        a.trmax = m > a.trmax ? m : a.trmax;   // a NaN here stays out of trmax; supports() still sees it
    }
    a.X = X;
    a.Y = Y;
    a.Z = Z;
    a.uL = uL;
    a.vL = vL;
    a.uR = uR;
    a.vR = vR;
    a.cap = cap;
    a.n_dev = n_dev;
    a.n_host = n_host;
    a.thresh = thresh;
    a.counts = counts ? counts + h_begin : counts;
    a.masks = reinterpret_cast<unsigned long long *>(masks);
    a.mask_words = mask_words;
    a.work = work;
    a.n_rec = 0;
    hipStream_t s = (hipStream_t)stream;

    // The filter is written for the pinhole pattern K = [fx 0 cx; 0 fy cy; 0 0 1]
    // (for which the skipped products are exact zeros) and finite camera constants.
    bool pinhole = K[1] == 0.0 && K[3] == 0.0 && K[6] == 0.0 && K[7] == 0.0 && K[8] == 1.0;
    for (int i = 0; i < 9; ++i)
        pinhole = pinhole && (K[i] - K[i] == 0.0);
    for (int i = 0; i < 3; ++i)
        pinhole = pinhole && (t_right[i] - t_right[i] == 0.0);
    if (work && !masks && pinhole && cap > 0) {
        a.n_rec = (size_t)n_sets * cap;
        // slam_select_kernels / SLAM_SCORE_IMPL pick the fp64-only filter kernel or force one form of the
        // fp32-first-level kernel (A/B timing, tests); the counts are identical either way.
        const int impl = score_impl();
        if (!work_ready) {
            hipError_t e =
                hipMemsetAsync(work + SCORE_WORK_PER_REC * a.n_rec, 0, sizeof(double) * SCORE_HDR * (size_t)n_sets, s);
            if (e != hipSuccess)
                return fail(SLAM_ERR_LAUNCH, "slam_score_hypotheses: %s", hipGetErrorString(e));
            const int chunks = (cap + 255) / 256;
            if ((long long)n_sets * chunks > 0x7FFFFFFFLL)
                return fail(SLAM_ERR_BAD_ARG, "slam_score_hypotheses: grid too large");
            hipLaunchKernelGGL(score_prep_kernel, dim3((unsigned)(n_sets * chunks)), dim3(256), 0, s, a, chunks);
        }
        a.blocks_per_set = (H + 255) / 256;
        const long long blocks = (long long)n_sets * a.blocks_per_set;
        if (blocks > 0x7FFFFFFFLL)
            return fail(SLAM_ERR_BAD_ARG, "slam_score_hypotheses: grid too large");
        if (t_right[1] == 0.0 && t_right[2] == 0.0) {
            if (impl == SLAM_SCORE_F64) {
                hipLaunchKernelGGL(score_fast_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, a);
            } else if (impl == SLAM_SCORE_MFMA) {
                // one wavefront per 32 hypotheses, four to a block
                a.blocks_per_set = (H + 127) / 128;
                if ((long long)n_sets * a.blocks_per_set > 0x7FFFFFFFLL)
                    return fail(SLAM_ERR_BAD_ARG, "slam_score_hypotheses: grid too large");
                hipLaunchKernelGGL(score_mfma_kernel, dim3((unsigned)(n_sets * a.blocks_per_set)), dim3(256), 0, s, a);
            } else if (impl == SLAM_SCORE_SPLIT || (impl != SLAM_SCORE_WIDE && (long long)n_sets * ((H + 63) / 64) < 3072)) {
                // fewer than three wavefronts per SIMD: split the points of a set over a block
                a.blocks_per_set = (H + 63) / 64;
                hipLaunchKernelGGL(score_mixed_kernel<true>, dim3((unsigned)(n_sets * a.blocks_per_set)), dim3(256), 0, s,
                                   a);
            } else {
                hipLaunchKernelGGL(score_mixed_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
            }
        } else {
            hipLaunchKernelGGL(score_fast_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
        }
        return check_launch("slam_score_hypotheses(fast)");
    }

    a.blocks_per_set = (H + 4 * SCORE_HPW - 1) / (4 * SCORE_HPW);
    const long long blocks = (long long)n_sets * a.blocks_per_set;
    if (blocks > 0x7FFFFFFFLL)
        return fail(SLAM_ERR_BAD_ARG, "slam_score_hypotheses: grid too large");
    if (masks && mask_words > 0) {
        const size_t total = (size_t)n_sets * H * mask_words;
        const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(clear_mask_tail_kernel, dim3(grid), dim3(256), 0, s, a.masks, n_sets, H, mask_words, n_dev,
                           n_host, cap);
    }
    hipLaunchKernelGGL(score_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return check_launch("slam_score_hypotheses");
}

// ---- the entry point that allocates nothing -------------------------------------------------
// Same scoring as slam_score_hypotheses with a workspace, but the workspace is this array: a
// fixed piece of every GPU's memory that comes with the code object, so the call needs no
// hipMalloc from anyone (callers under graph capture, serving loops, harnesses that forbid
// allocation).  Each GPU has its own copy at its own address; the name below is only a host-side
// handle, the address a kernel can use comes from hipGetSymbolAddress with that GPU current.
static_assert(SLAM_SCORE_WORK_PER_POINT >= slam::SCORE_WORK_PER_REC + slam::SCORE_HDR, "workspace per point slot");
static __device__ __align__(256) double scratch[(size_t)SLAM_SCORE_STATIC_WORK_DOUBLES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex scratch_lock;
double *scratch_ptr[MAX_DEVICES];     // per device, filled on the first call made with that device current

int scratch_for_current_device(double **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return fail(SLAM_ERR_LAUNCH, "slam_score_hypotheses_static: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= MAX_DEVICES)
        return fail(SLAM_ERR_UNSUPPORTED, "slam_score_hypotheses_static: device %d out of range", dev);
    std::lock_guard<std::mutex> guard(scratch_lock);
    if (!scratch_ptr[dev]) {
        double *p = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void **>(&p), HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(SLAM_ERR_LAUNCH, "slam_score_hypotheses_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
        scratch_ptr[dev] = p;
    }
    *out = scratch_ptr[dev];
    return SLAM_OK;
}
}  // namespace

extern "C" int slam_score_hypotheses_static(const double *Rt, int n_sets, int H, const double *K,
                                            const double *t_right, const double *X, const double *Y, const double *Z,
                                            const float *uL, const float *vL, const float *uR, const float *vR, int cap,
                                            const int32_t *n_dev, int n_host, double thresh, int32_t *counts,
                                            void *stream)
{
    // Contract: every size check below comes BEFORE the first HIP call (scratch_for_current_device), so a
    // refused call touches no device and the refusals can be exercised on a machine without a GPU.
    if (n_sets < 0 || H < 0 || cap < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_score_hypotheses_static: negative size");
    const long long need = (long long)SLAM_SCORE_WORK_PER_POINT * n_sets * (cap > 1 ? cap : 1);
    if (need > (long long)SLAM_SCORE_STATIC_WORK_DOUBLES)
        return fail(SLAM_ERR_UNSUPPORTED,
                    "slam_score_hypotheses_static: n_sets * cap = %lld point slots exceed the built-in workspace of %d "
                    "(use slam_score_hypotheses with a caller-owned workspace)",
                    (long long)n_sets * (cap > 1 ? cap : 1), SLAM_SCORE_STATIC_MAX_POINTS);
    if (n_sets == 0 || H == 0)
        return SLAM_OK;
    double *work = nullptr;
    const int st = scratch_for_current_device(&work);
    if (st != SLAM_OK)
        return st;
    // work_ready = 0: the point records and per-set bounds are rebuilt (and the bounds cleared) by
    // every call, so nothing of the previous call's contents is relied on
    return slam_score_hypotheses_range(Rt, n_sets, H, 0, H, nullptr, K, t_right, X, Y, Z, uL, vL, uR, vR, cap, n_dev,
                                       n_host, thresh, counts, nullptr, 0, work, 0, stream);
}
