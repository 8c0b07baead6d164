// Batched PnP-RANSAC around the scoring kernel (gfx950): minimal-sample drawing, P3P pose
// hypotheses, the reference's adaptive stopping rule replayed over the supporter counts,
// Levenberg-Marquardt refinement of each winner on its supporters, and the final choice between
// the winner and its refinement (ransac_for_pnp, utils/utils.py:413-464).
//
// The reference draws with Python's process-global `random`, solves with cv2 (AP3P, then
// SOLVEPNP_ITERATIVE) and walks the iterations one by one.  Here every frame pair draws from its
// own counter-based generator, all hypotheses of all pairs are solved and scored at once, and the
// sequential stopping rule is applied afterwards to the counts, which selects exactly the
// hypothesis the one-by-one loop would have stopped with for the same draws.
#include "common.h"
#include "supports.h"

namespace slam {
namespace {

// ---- sampling --------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t &s)
{
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Four distinct positions in [0, n), uniformly, in draw order: the k-th draw is a rank among the
// n - k positions still free.  Stream = splitmix64 from seed ^ f(set, h); n < 4 gives -1s.
__global__ void __launch_bounds__(256) sample_kernel(uint64_t seed, int n_sets, int H, const int32_t *n_dev,
                                                     int n_host, int32_t *samples)
{
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)n_sets * H)
        return;
    const int set = (int)(gid / H);
    const int h = (int)(gid - (long long)set * H);
    const int n = n_dev ? n_dev[set] : n_host;
    int4 out = make_int4(-1, -1, -1, -1);
    if (n >= 4) {
        uint64_t s = seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(set + 1)) ^ (0x8CB92BA72F3D8DD7ull * (uint64_t)(h + 1));
        int sorted[4];
        int o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t r = splitmix64(s);
            int j = (int)(((r >> 32) * (uint64_t)(n - k)) >> 32);
#pragma unroll
            for (int m = 0; m < k; ++m)
                j += (j >= sorted[m]) ? 1 : 0;
            o[k] = j;
            // insert j into the ascending list of taken positions
            int v = j;
#pragma unroll
            for (int m = 0; m < k; ++m) {
                const int lo = min(sorted[m], v), hi = max(sorted[m], v);
                sorted[m] = lo;
                v = hi;
            }
            sorted[k] = v;
        }
        out = make_int4(o[0], o[1], o[2], o[3]);
    }
    reinterpret_cast<int4 *>(samples)[gid] = out;
}

// ---- P3P ---------------------------------------------------------------------------------
struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double norm(V3 a) { return sqrt(dot(a, a)); }

// largest real root of t^3 + B t^2 + C t + D
__device__ double cubic_largest_root(double B, double C, double D)
{
    const double P = C - B * B / 3.0;
    const double Q = 2.0 * B * B * B / 27.0 - B * C / 3.0 + D;
    const double disc = Q * Q / 4.0 + P * P * P / 27.0;
    double s;
    if (disc > 0.0) {
        const double sd = sqrt(disc);
        s = cbrt(-Q / 2.0 + sd) + cbrt(-Q / 2.0 - sd);
    } else {
        const double m = 2.0 * sqrt(-P / 3.0);
        double arg = (P != 0.0) ? 3.0 * Q / (P * m) : 0.0;
        arg = fmin(1.0, fmax(-1.0, arg));
        s = m * cos(acos(arg) / 3.0);
    }
    double t = s - B / 3.0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double f = ((t + B) * t + C) * t + D;
        const double fp = (3.0 * t + 2.0 * B) * t + C;
        if (fp != 0.0 && isfinite(f / fp))
            t -= f / fp;
    }
    return t;
}

// Real roots of A4 x^4 + ... + A0 (Ferrari through the resolvent cubic, Newton-polished).
// Near-double roots with a slightly negative discriminant are kept as double roots.
__device__ int quartic_real_roots(double A4, double A3, double A2, double A1, double A0, double (&x)[4])
{
    if (!(fabs(A4) > 0.0))
        return 0;
    const double a = A3 / A4, b = A2 / A4, c = A1 / A4, d = A0 / A4;
    const double a2 = a * a;
    const double p = b - 3.0 * a2 / 8.0;
    const double q = c - a * b / 2.0 + a2 * a / 8.0;
    const double r = d - a * c / 4.0 + a2 * b / 16.0 - 3.0 * a2 * a2 / 256.0;
    double y[4];
    int n = 0;
    const double scale = fabs(p) + sqrt(fabs(r)) + 1e-300;
    if (fabs(q) <= 1e-14 * scale * sqrt(scale)) {
        // biquadratic in y
        double disc = p * p - 4.0 * r;
        if (disc < 0.0 && disc > -1e-9 * (p * p + fabs(4.0 * r)))
            disc = 0.0;
        if (disc >= 0.0) {
            const double sd = sqrt(disc);
            const double w[2] = {(-p + sd) / 2.0, (-p - sd) / 2.0};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (w[e] >= 0.0) {
                    y[n++] = sqrt(w[e]);
                    y[n++] = -sqrt(w[e]);
                }
            }
        }
    } else {
        const double z = cubic_largest_root(2.0 * p, p * p - 4.0 * r, -q * q);
        if (!(z > 0.0))
            return 0;
        const double al = sqrt(z);
        const double be = (p + z - q / al) / 2.0;
        const double ga = (p + z + q / al) / 2.0;
        double D1 = z - 4.0 * be, D2 = z - 4.0 * ga;
        if (D1 < 0.0 && D1 > -1e-9 * (z + fabs(4.0 * be)))
            D1 = 0.0;
        if (D2 < 0.0 && D2 > -1e-9 * (z + fabs(4.0 * ga)))
            D2 = 0.0;
        if (D1 >= 0.0) {
            const double sd = sqrt(D1);
            y[n++] = (-al + sd) / 2.0;
            y[n++] = (-al - sd) / 2.0;
        }
        if (D2 >= 0.0) {
            const double sd = sqrt(D2);
            y[n++] = (al + sd) / 2.0;
            y[n++] = (al - sd) / 2.0;
        }
    }
    for (int i = 0; i < n; ++i) {
        double v = y[i] - a / 4.0;
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const double f = (((A4 * v + A3) * v + A2) * v + A1) * v + A0;
            const double fp = ((4.0 * A4 * v + 3.0 * A3) * v + 2.0 * A2) * v + A1;
            const double st = f / fp;
            if (fp != 0.0 && isfinite(st) && fabs(st) < 0.1 * (fabs(v) + 1e-3))
                v -= st;
        }
        x[i] = v;
    }
    return n;
}

struct P3PArgs {
    double fx, fy, cx, cy;
    const double *X, *Y, *Z;
    const float *uL, *vL;
    int cap;
    const int32_t *n_dev;
    int n_host;
    const int32_t *samples;
    int n_sets, H;     // hypotheses solved per set
    int Hs, h0;        // hypotheses per set in memory, first one solved
    const int32_t *skip;
    double *Rt;
};

// Grunert's three-point resection on the first three correspondences of each sample; the fourth
// picks among the up-to-four solutions (the role the fourth point plays in cv2's AP3P,
// utils/utils.py:428,481-492).  lane = hypothesis.  A failed solve writes NaNs: such a pose
// supports nothing, which is what skipping the iteration amounts to in the stopping rule.
__global__ void __launch_bounds__(256) p3p_kernel(const P3PArgs a)
{
    const long long lid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (lid >= (long long)a.n_sets * a.H)
        return;
    const int set = (int)(lid / a.H);
    if (a.skip && a.skip[set])
        return;
    const long long gid = (long long)set * a.Hs + a.h0 + (lid - (long long)set * a.H);
    const double nan = __builtin_nan("");
    double out[12];
#pragma unroll
    for (int c = 0; c < 12; ++c)
        out[c] = nan;
    int n = a.n_dev ? a.n_dev[set] : a.n_host;
    n = min(max(n, 0), a.cap);
    const int4 smp = reinterpret_cast<const int4 *>(a.samples)[gid];
    const int idx[4] = {smp.x, smp.y, smp.z, smp.w};
    bool ok = true;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        ok = ok && idx[e] >= 0 && idx[e] < n;
    if (ok) {
        V3 P[4], f[3];
        double u3 = 0.0, v3 = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const size_t o = (size_t)set * a.cap + idx[e];
            P[e] = {a.X[o], a.Y[o], a.Z[o]};
            const double u = (double)a.uL[o], v = (double)a.vL[o];
            if (e < 3) {
                const V3 ray = {(u - a.cx) / a.fx, (v - a.cy) / a.fy, 1.0};
                f[e] = (1.0 / norm(ray)) * ray;
            } else {
                u3 = u;
                v3 = v;
            }
        }
        const double da = norm(P[1] - P[2]), db = norm(P[0] - P[2]), dc = norm(P[0] - P[1]);
        if (fmin(da, fmin(db, dc)) >= 1e-12) {
            const double ca = dot(f[1], f[2]), cb = dot(f[0], f[2]), cg = dot(f[0], f[1]);
            const double a2 = da * da, b2 = db * db, c2 = dc * dc;
            const double q = (a2 - c2) / b2;
            const double A4 = (q - 1.0) * (q - 1.0) - 4.0 * c2 / b2 * ca * ca;
            const double A3 = 4.0 * (q * (1.0 - q) * cb - (1.0 - (a2 + c2) / b2) * ca * cg + 2.0 * c2 / b2 * ca * ca * cb);
            const double A2 = 2.0 * (q * q - 1.0 + 2.0 * q * q * cb * cb + 2.0 * (b2 - c2) / b2 * ca * ca -
                                     4.0 * (a2 + c2) / b2 * ca * cb * cg + 2.0 * (b2 - a2) / b2 * cg * cg);
            const double A1 = 4.0 * (-q * (1.0 + q) * cb + 2.0 * a2 / b2 * cg * cg * cb - (1.0 - (a2 + c2) / b2) * ca * cg);
            const double A0 = (1.0 + q) * (1.0 + q) - 4.0 * a2 / b2 * cg * cg;
            double roots[4];
            int nr = 0;
            if (isfinite(A4) && isfinite(A3) && isfinite(A2) && isfinite(A1) && isfinite(A0))
                nr = quartic_real_roots(A4, A3, A2, A1, A0, roots);
            // world frame of the triangle
            const V3 w1 = (1.0 / dc) * (P[1] - P[0]);
            V3 w3 = cross(w1, P[2] - P[0]);
            const double w3n = norm(w3);
            double best_err = __builtin_inf();
            if (w3n > 0.0) {
                w3 = (1.0 / w3n) * w3;
                const V3 w2 = cross(w3, w1);
                for (int i = 0; i < nr; ++i) {
                    const double v = roots[i];
                    if (!(v > 0.0))
                        continue;
                    const double den = 2.0 * (cg - v * ca);
                    if (fabs(den) < 1e-14)
                        continue;
                    const double u = ((q - 1.0) * v * v - 2.0 * q * cb * v + 1.0 + q) / den;
                    if (!(u > 0.0))
                        continue;
                    const double s1sq = c2 / (1.0 + u * u - 2.0 * u * cg);
                    if (!(s1sq > 0.0))
                        continue;
                    const double s1 = sqrt(s1sq);
                    const V3 Q0 = s1 * f[0], Q1 = (u * s1) * f[1], Q2 = (v * s1) * f[2];
                    V3 c1 = Q1 - Q0;
                    const double c1n = norm(c1);
                    V3 c3 = cross(c1, Q2 - Q0);
                    const double c3n = norm(c3);
                    if (!(c1n > 0.0) || !(c3n > 0.0))
                        continue;
                    c1 = (1.0 / c1n) * c1;
                    c3 = (1.0 / c3n) * c3;
                    const V3 c2v = cross(c3, c1);
                    // R = c1 w1^T + c2 w2^T + c3 w3^T
                    double R[9];
                    R[0] = c1.x * w1.x + c2v.x * w2.x + c3.x * w3.x;
                    R[1] = c1.x * w1.y + c2v.x * w2.y + c3.x * w3.y;
                    R[2] = c1.x * w1.z + c2v.x * w2.z + c3.x * w3.z;
                    R[3] = c1.y * w1.x + c2v.y * w2.x + c3.y * w3.x;
                    R[4] = c1.y * w1.y + c2v.y * w2.y + c3.y * w3.y;
                    R[5] = c1.y * w1.z + c2v.y * w2.z + c3.y * w3.z;
                    R[6] = c1.z * w1.x + c2v.z * w2.x + c3.z * w3.x;
                    R[7] = c1.z * w1.y + c2v.z * w2.y + c3.z * w3.y;
                    R[8] = c1.z * w1.z + c2v.z * w2.z + c3.z * w3.z;
                    auto rot = [&](V3 p) -> V3 {
                        return {R[0] * p.x + R[1] * p.y + R[2] * p.z, R[3] * p.x + R[4] * p.y + R[5] * p.z,
                                R[6] * p.x + R[7] * p.y + R[8] * p.z};
                    };
                    const V3 t = Q0 - rot(P[0]);
                    // the triangle's side lengths must be reproduced (rejects spurious roots)
                    const V3 e1 = rot(P[1]) + t - Q1, e2 = rot(P[2]) + t - Q2;
                    const double resid = fmax(fmax(fabs(e1.x), fmax(fabs(e1.y), fabs(e1.z))),
                                              fmax(fabs(e2.x), fmax(fabs(e2.y), fabs(e2.z))));
                    const double qmax = fmax(fmax(norm(Q0), norm(Q1)), norm(Q2));
                    if (!(resid <= 1e-6 * fmax(1.0, qmax)))
                        continue;
                    const V3 Y3 = rot(P[3]) + t;
                    if (!(Y3.z > 0.0))
                        continue;
                    const double eu = a.fx * Y3.x / Y3.z + a.cx - u3, ev = a.fy * Y3.y / Y3.z + a.cy - v3;
                    const double err = sqrt(eu * eu + ev * ev);
                    if (err < best_err) {
                        best_err = err;
                        out[0] = R[0]; out[1] = R[1]; out[2] = R[2]; out[3] = t.x;
                        out[4] = R[3]; out[5] = R[4]; out[6] = R[5]; out[7] = t.y;
                        out[8] = R[6]; out[9] = R[7]; out[10] = R[8]; out[11] = t.z;
                    }
                }
            }
        }
    }
    double2 *dst = reinterpret_cast<double2 *>(a.Rt + (size_t)gid * 12);
#pragma unroll
    for (int c = 0; c < 6; ++c)
        dst[c] = make_double2(out[2 * c], out[2 * c + 1]);
}

// ---- stopping rule -----------------------------------------------------------------------
// utils/utils.py:475-478 with p = 0.999, s = 4, clipped to max_it
__device__ __forceinline__ int iterations_needed(double eps, int max_it)
{
    if (eps == 0.0)
        return 0;
    const double v = ceil(log(1.0 - 0.999) / log(1.0 - pow(1.0 - eps, 4.0)));
    return v < (double)max_it ? (int)v : max_it;
}

// One block per point set.  Replays   while i < I and eps != 0: ... (utils/utils.py:422-446)
// over counts[0..H): winner = first strictly-best hypothesis among the iterations the loop would
// have run, iters = how many it would have run.  A count of 0 leaves eps and I unchanged, exactly
// like an iteration whose minimal solve failed.
__global__ void __launch_bounds__(256) ransac_select_kernel(const int32_t *__restrict__ counts,
                                                            const double *__restrict__ Rt, int H, int n_avail,
                                                            const int32_t *__restrict__ n_dev, int n_host, int cap,
                                                            int max_it, int32_t *__restrict__ best,
                                                            int32_t *__restrict__ best_count,
                                                            double *__restrict__ best_Rt, int32_t *__restrict__ iters,
                                                            int32_t *__restrict__ stopped_out)
{
    __shared__ unsigned long long wave_key[4];
    __shared__ int stop_at[4];
    __shared__ unsigned long long carry_s;
    __shared__ int done_s;
    const int set = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int n = n_dev ? n_dev[set] : n_host;
    n = min(max(n, 0), cap);
    const int32_t *c = counts + (size_t)set * H;
    const int limit = min(min(H, max_it), n_avail);   // hypotheses scored so far, at most the loop's cap
    // key orders by count descending, then index ascending; 0 = nothing yet
    unsigned long long carry = 0ull;
    int total_iters = (n >= 4) ? limit : 0;
    unsigned long long final_key = 0ull;
    bool stopped = (n < 4);
    for (int base = 0; base < limit && !stopped; base += 256) {
        const int h = base + tid;
        unsigned long long key = 0ull;
        if (h < limit) {
            const int cnt = max(c[h], 0);
            if (cnt > 0)
                key = ((unsigned long long)(unsigned)cnt << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)h);
        }
        // inclusive prefix max inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_up(key, off);
            if (lane >= off)
                key = o > key ? o : key;
        }
        if (lane == 63)
            wave_key[wave] = key;
        __syncthreads();
        unsigned long long pre = carry;
        for (int w = 0; w < wave; ++w)
            pre = wave_key[w] > pre ? wave_key[w] : pre;
        key = pre > key ? pre : key;   // best over iterations 0..h
        // would the loop stop after iteration h?
        int stop_h = INT32_MAX;
        if (h < limit) {
            const int bc = (int)(key >> 32);
            double eps = 1.0 - ((double)bc / (double)n);
            eps = eps < 0.99 ? eps : 0.99;
            const int I = iterations_needed(eps, max_it);
            if (eps == 0.0 || h + 1 >= I)
                stop_h = h;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            stop_h = min(stop_h, __shfl_xor(stop_h, off));
        if (lane == 0)
            stop_at[wave] = stop_h;
        __syncthreads();
        const int first_stop = min(min(stop_at[0], stop_at[1]), min(stop_at[2], stop_at[3]));
        if (first_stop != INT32_MAX) {
            if (h == first_stop) {
                carry_s = key;
                done_s = first_stop + 1;
            }
            stopped = true;
        } else if (tid == 255) {
            carry_s = key;
            done_s = min(base + 256, limit);
        }
        __syncthreads();
        carry = carry_s;
        final_key = carry;
        total_iters = done_s;
        __syncthreads();
    }
    const int bc = (int)(final_key >> 32);
    const int bh = bc > 0 ? (int)(0xFFFFFFFFu - (unsigned)(final_key & 0xFFFFFFFFull)) : -1;
    if (tid == 0) {
        best[set] = bh;
        best_count[set] = bc;
        if (iters)
            iters[set] = total_iters;
        // the loop is over for this set: the rule fired, or the cap itself was reached, or n < 4
        if (stopped_out)
            stopped_out[set] = (stopped || limit >= min(H, max_it)) ? 1 : 0;
    }
    if (tid < 12 && best_Rt)
        best_Rt[(size_t)set * 12 + tid] = bh >= 0 ? Rt[((size_t)set * H + bh) * 12 + tid] : __builtin_nan("");
}

// ---- refinement ----------------------------------------------------------------------------
constexpr int NSUM = 28;   // 21 upper-triangle entries of J^T J, 6 of J^T r, the cost

struct RefineArgs {
    double fx, fy, cx, cy;
    Mat33 K;          // the supporter test (used when masks is null, and for ref_count)
    Vec3 tr;
    double thresh;
    const double *X, *Y, *Z;
    const float *uL, *vL, *uR, *vR;
    int cap;
    const int32_t *n_dev;
    int n_host;
    const unsigned long long *masks;   // supporters of Rt0 given by the caller, or null: tested here
    int mask_words;
    const double *Rt0;
    double *Rt_out;
    int32_t *ok;
    int32_t *ref_count;                // supporters of Rt_out, or null
    int iterations;
};

__device__ __forceinline__ void exp_so3(const double *w, double *E)
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double A, B;
    if (th2 < 0.0625) {
        // A = sin(th)/th and B = (1 - cos th)/th^2 as series in th^2 (an LM step is a small rotation;
        // below th = 0.25 ten terms leave less than 1e-17)
        A = 1.0;
        B = 0.5;
        double ta = 1.0, tb = 0.5;
#pragma unroll
        for (int n = 1; n <= 9; ++n) {
            ta *= -th2 / (double)((2 * n) * (2 * n + 1));
            tb *= -th2 / (double)((2 * n + 1) * (2 * n + 2));
            A += ta;
            B += tb;
        }
    } else {
        const double th = sqrt(th2);
        A = sin(th) / th;
        B = (1.0 - cos(th)) / th2;
    }
    const double K[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double kk = 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m)
                kk += K[3 * i + m] * K[3 * m + j];
            E[3 * i + j] = (i == j ? 1.0 : 0.0) + A * K[3 * i + j] + B * kk;
        }
}

// The value another lane of the same 16-lane row holds (CTRL: the operand-permute pattern).
template <int CTRL>
__device__ __forceinline__ double row_peer(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int plo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    const int phi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(phi, plo);
}

// Solves the 6x6 system M x = rhs (Gaussian elimination, partial pivoting); false if singular.
// Every index is a compile-time constant (the row exchange is a chain of selects), so the
// matrix lives in registers.
__device__ __forceinline__ bool solve6(double (&M)[6][6], double (&rhs)[6], double (&x)[6])
{
    bool ok = true;
    double inv[6];   // one division per pivot: the thread that solves runs alone, every division is ~30 dependent instructions
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        double big = fabs(M[c][c]);
#pragma unroll
        for (int r = c + 1; r < 6; ++r)
            if (fabs(M[r][c]) > big) {
                big = fabs(M[r][c]);
                piv = r;
            }
        ok = ok && big > 0.0 && isfinite(big);
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            const bool sw = piv == r;
#pragma unroll
            for (int k = c; k < 6; ++k) {   // columns left of c are never read again
                const double up = M[c][k], dn = M[r][k];
                M[c][k] = sw ? dn : up;
                M[r][k] = sw ? up : dn;
            }
            const double up = rhs[c], dn = rhs[r];
            rhs[c] = sw ? dn : up;
            rhs[r] = sw ? up : dn;
        }
        inv[c] = 1.0 / M[c][c];
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {
            const double f = M[r][c] * inv[c];
#pragma unroll
            for (int k = c + 1; k < 6; ++k)   // (column c itself is not read again)
                M[r][k] = __builtin_fma(-f, M[c][k], M[r][k]);
            rhs[r] = __builtin_fma(-f, rhs[c], rhs[r]);
        }
    }
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        double s = rhs[r];
#pragma unroll
        for (int k = r + 1; k < 6; ++k)
            s = __builtin_fma(-M[r][k], x[k], s);
        x[r] = s * inv[r];
    }
    return ok;
}

// Levenberg-Marquardt on the left-image reprojection error over the supporters of the winning
// hypothesis, started from it (the job SOLVEPNP_ITERATIVE does at utils/utils.py:452).
// One block per point set; every iteration is one block-wide reduction of NSUM sums.  Thread t
// owns points t, t+256, ...: the first REFINE_JREG of them stay in registers for all iterations
// (an iteration then touches no memory but LDS), later ones are re-read.
// STEREO adds the right image's residuals of the same points (camera offset a.tr, keypoints
// uR/vR) to the cost: not what the reference minimises, an option for callers who want the
// pose without the heading bias of the left-only cost.
constexpr int REFINE_JREG = 6;

template <bool STEREO>
__global__ void __launch_bounds__(256, 2) refine_kernel(const RefineArgs a)
{
    __shared__ double part[4][NSUM];
    __shared__ double rowsum[16][NSUM];
    __shared__ double sums[NSUM], accepted[NSUM];
    __shared__ double pose[12], cand[12], pose0[12];
    __shared__ int state;   // 0 = iterate, 1 = finished
    const int set = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int n = a.n_dev ? a.n_dev[set] : a.n_host;
    n = min(max(n, 0), a.cap);
    const unsigned long long *mask = a.masks ? a.masks + (size_t)set * a.mask_words : nullptr;
    const size_t pbase = (size_t)set * a.cap;
    if (tid < 12)
        pose[tid] = cand[tid] = pose0[tid] = a.Rt0[(size_t)set * 12 + tid];
    __syncthreads();

    // is point i a supporter of the starting pose?
    auto starts_in = [&](int i, double X, double Y, double Z, float u, float v) -> bool {
        if (mask)
            return (mask[i >> 6] >> (i & 63)) & 1ull;
        return supports(pose0, a.K, a.tr, X, Y, Z, (double)u, (double)v, (double)a.uR[pbase + i],
                        (double)a.vR[pbase + i], a.thresh);
    };

    double pX[REFINE_JREG], pY[REFINE_JREG], pZ[REFINE_JREG];
    float pu[REFINE_JREG], pv[REFINE_JREG];
    unsigned reg_in = 0;                 // bit j: point tid + 256 j (j < REFINE_JREG) is a supporter
    unsigned long long later_in = 0;     // the same for the next 64 points of this thread
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < REFINE_JREG; ++j) {
        const int i = tid + 256 * j;
        const bool valid = i < n;
        pX[j] = valid ? a.X[pbase + i] : 0.0;
        pY[j] = valid ? a.Y[pbase + i] : 0.0;
        pZ[j] = valid ? a.Z[pbase + i] : 1.0;
        pu[j] = valid ? a.uL[pbase + i] : 0.0f;
        pv[j] = valid ? a.vL[pbase + i] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < REFINE_JREG; ++j) {
        const int i = tid + 256 * j;
        if (i < n && starts_in(i, pX[j], pY[j], pZ[j], pu[j], pv[j])) {
            reg_in |= 1u << j;
            ++cnt;
        }
    }
    for (int j = REFINE_JREG, i = tid + 256 * REFINE_JREG; i < n; ++j, i += 256) {
        const bool in = starts_in(i, a.X[pbase + i], a.Y[pbase + i], a.Z[pbase + i], a.uL[pbase + i], a.vL[pbase + i]);
        if (in && j - REFINE_JREG < 64)
            later_in |= 1ull << (j - REFINE_JREG);
        cnt += in ? 1 : 0;
    }

    // cost, J^T J and J^T r at the pose in `cand`, over the supporters; result in sums[]
    auto accumulate = [&]() {
        double acc[NSUM];
#pragma unroll
        for (int k = 0; k < NSUM; ++k)
            acc[k] = 0.0;
        double R[12];
#pragma unroll
        for (int k = 0; k < 12; ++k)
            R[k] = cand[k];
        // one camera's two residual rows for the point y (left frame), seen from a camera at
        // offset o: the projection of z = y + o; d z / d(pose) = d y / d(pose) = [-[y]x | I]
        auto add_cam = [&](double y0, double y1, double y2, double o0, double o1, double o2, double u, double v) {
            const double z0 = y0 + o0, z1 = y1 + o1, z2 = y2 + o2;
            const double iz = 1.0 / z2;
            const double r0 = a.fx * z0 * iz + a.cx - u;
            const double r1 = a.fy * z1 * iz + a.cy - v;
            const double du0 = a.fx * iz, du2 = -a.fx * z0 * iz * iz;
            const double dv1 = a.fy * iz, dv2 = -a.fy * z1 * iz * iz;
            // the u row has no entry for t_y (column 4), the v row none for t_x (column 3)
            const double Ju[6] = {-(-du2 * y1), -(-du0 * y2 + du2 * y0), -(du0 * y1), du0, 0.0, du2};
            const double Jv[6] = {-(dv1 * y2 - dv2 * y1), -(dv2 * y0), -(-dv1 * y0), 0.0, dv1, dv2};
            int k = 0;
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int q = p; q < 6; ++q, ++k) {
                    if (p != 4 && q != 4)
                        acc[k] = __builtin_fma(Ju[p], Ju[q], acc[k]);
                    if (p != 3 && q != 3)
                        acc[k] = __builtin_fma(Jv[p], Jv[q], acc[k]);
                }
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                if (p != 4)
                    acc[21 + p] = __builtin_fma(Ju[p], r0, acc[21 + p]);
                if (p != 3)
                    acc[21 + p] = __builtin_fma(Jv[p], r1, acc[21 + p]);
            }
            acc[27] = __builtin_fma(r0, r0, __builtin_fma(r1, r1, acc[27]));
        };
        auto add_point = [&](int i, double X, double Y, double Z, float u, float v) {
            const double y0 = R[0] * X + R[1] * Y + R[2] * Z + R[3];
            const double y1 = R[4] * X + R[5] * Y + R[6] * Z + R[7];
            const double y2 = R[8] * X + R[9] * Y + R[10] * Z + R[11];
            add_cam(y0, y1, y2, 0.0, 0.0, 0.0, (double)u, (double)v);
            if constexpr (STEREO)
                add_cam(y0, y1, y2, a.tr.v[0], a.tr.v[1], a.tr.v[2], (double)a.uR[pbase + i], (double)a.vR[pbase + i]);
        };
#pragma unroll
        for (int j = 0; j < REFINE_JREG; ++j)
            if ((reg_in >> j) & 1u)
                add_point(tid + 256 * j, pX[j], pY[j], pZ[j], pu[j], pv[j]);
        for (int j = REFINE_JREG, i = tid + 256 * REFINE_JREG; i < n; ++j, i += 256) {
            const double X = a.X[pbase + i], Y = a.Y[pbase + i], Z = a.Z[pbase + i];
            const float u = a.uL[pbase + i], v = a.vL[pbase + i];
            const bool in = j - REFINE_JREG < 64 ? (later_in >> (j - REFINE_JREG)) & 1ull : starts_in(i, X, Y, Z, u, v);
            if (in)
                add_point(i, X, Y, Z, u, v);
        }
        // 16-lane rows first, through the lane-permuting operand path of the vector ALU (four
        // steps, no LDS round trip), then the 16 row sums of the block through LDS
#pragma unroll
        for (int k = 0; k < NSUM; ++k) {
            double v = acc[k];
            v += row_peer<0xB1>(v);    // quad_perm [1,0,3,2]
            v += row_peer<0x4E>(v);    // quad_perm [2,3,0,1]
            v += row_peer<0x141>(v);   // row_half_mirror
            v += row_peer<0x140>(v);   // row_mirror
            if ((lane & 15) == 0)
                rowsum[tid >> 4][k] = v;
        }
        __syncthreads();
        if (tid < NSUM) {
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                t += rowsum[r][tid];
            sums[tid] = t;
        }
        __syncthreads();
    };

    // supporters
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        cnt += __shfl_xor(cnt, off);
    if (lane == 0)
        part[wave][0] = (double)cnt;
    __syncthreads();
    const int n_sup = (int)(part[0][0] + part[1][0] + part[2][0] + part[3][0]);
    __syncthreads();

    bool good = n_sup >= 3;
    if (good) {
        accumulate();
        // thread 0 carries the solver state; the normal equations of the accepted pose wait in LDS
        double cost = sums[27], lam = 1e-3;
        if (tid < 27)
            accepted[tid] = sums[tid];
        if (tid == 0)
            state = isfinite(cost) ? 0 : 2;   // 2 = failed
        __syncthreads();
        for (int it = 0; it < a.iterations && state == 0; ++it) {
            if (tid == 0) {
                double M[6][6], rhs[6], step[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int q = p; q < 6; ++q) {
                        const int k = p * 6 - p * (p - 1) / 2 + (q - p);   // upper triangle, row-major
                        M[p][q] = accepted[k];
                        M[q][p] = accepted[k];
                    }
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    M[p][p] += lam * (M[p][p] + 1e-12);
                    rhs[p] = -accepted[21 + p];
                }
                const bool solved = solve6(M, rhs, step);
                const double tmax = fmax(fmax(fabs(pose[3]), fabs(pose[7])), fabs(pose[11]));
                const double wmax = fmax(fmax(fabs(step[0]), fabs(step[1])), fabs(step[2]));
                const double dmax = fmax(fmax(fabs(step[3]), fabs(step[4])), fabs(step[5]));
                if (!solved) {
                    state = 2;
                } else if (wmax + dmax / (1.0 + tmax) < 1e-11) {
                    state = 1;   // converged: the step no longer moves the pose
                } else {
                    double E[9];
                    exp_so3(step, E);
                    for (int i = 0; i < 3; ++i) {
                        for (int j = 0; j < 3; ++j)
                            cand[4 * i + j] = E[3 * i] * pose[j] + E[3 * i + 1] * pose[4 + j] + E[3 * i + 2] * pose[8 + j];
                        cand[4 * i + 3] = E[3 * i] * pose[3] + E[3 * i + 1] * pose[7] + E[3 * i + 2] * pose[11] + step[3 + i];
                    }
                    // A step this small is the last one that matters: what would follow it is smaller by the
                    // damping factor (below 1e-4 by now) -- take it without paying an evaluation to confirm it.
                    if (wmax + dmax / (1.0 + tmax) < 1e-7) {
                        for (int k = 0; k < 12; ++k)
                            pose[k] = cand[k];
                        state = 1;
                    }
                }
            }
            __syncthreads();
            if (state != 0)
                break;
            accumulate();
            if (tid == 0) {
                const double c2 = sums[27];
                if (isfinite(c2) && c2 < cost) {
#pragma unroll
                    for (int k = 0; k < 12; ++k)
                        pose[k] = cand[k];
                    for (int k = 0; k < 27; ++k)
                        accepted[k] = sums[k];
                    const bool small = cost - c2 < 1e-14 * fmax(cost, 1e-300);
                    cost = c2;
                    if (small)
                        state = 1;
                    lam = fmax(lam * 0.3, 1e-12);
                } else if (isfinite(c2) && c2 - cost <= 1e-12 * cost) {
                    state = 1;   // the step moves the cost by less than its rounding noise: at the optimum
                } else {
                    lam *= 10.0;
                    if (lam > 1e12)
                        state = 1;
                }
            }
            __syncthreads();
        }
        good = state != 2;
    }
    __syncthreads();
    if (tid == 0) {
        double R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
        if (good) {
            // re-orthonormalise against drift: two Newton-Schulz steps R <- R (3 I - R^T R) / 2
            for (int s = 0; s < 2; ++s) {
                double G[9], N[9];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j)
                        G[3 * i + j] = R[i] * R[j] + R[3 + i] * R[3 + j] + R[6 + i] * R[6 + j];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) {
                        double v = 0.0;
                        for (int m = 0; m < 3; ++m)
                            v += R[3 * i + m] * ((m == j ? 3.0 : 0.0) - G[3 * m + j]);
                        N[3 * i + j] = 0.5 * v;
                    }
                for (int k = 0; k < 9; ++k)
                    R[k] = N[k];
            }
            for (int k = 0; k < 9; ++k)
                good = good && isfinite(R[k]);
        }
        double *o = a.Rt_out + (size_t)set * 12;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j)
                o[4 * i + j] = cand[4 * i + j] = good ? R[3 * i + j] : pose0[4 * i + j];
            o[4 * i + 3] = cand[4 * i + 3] = good ? pose[4 * i + 3] : pose0[4 * i + 3];
        }
        if (a.ok)
            a.ok[set] = good ? 1 : 0;
    }
    if (!a.ref_count)
        return;
    // supporters of the pose just written (utils/utils.py:455)
    __syncthreads();
    double F[12];
#pragma unroll
    for (int k = 0; k < 12; ++k)
        F[k] = cand[k];
    int ref = 0;
#pragma unroll
    for (int j = 0; j < REFINE_JREG; ++j) {
        const int i = tid + 256 * j;
        if (i < n && supports(F, a.K, a.tr, pX[j], pY[j], pZ[j], (double)pu[j], (double)pv[j], (double)a.uR[pbase + i],
                              (double)a.vR[pbase + i], a.thresh))
            ++ref;
    }
    for (int i = tid + 256 * REFINE_JREG; i < n; i += 256)
        if (supports(F, a.K, a.tr, a.X[pbase + i], a.Y[pbase + i], a.Z[pbase + i], (double)a.uL[pbase + i],
                     (double)a.vL[pbase + i], (double)a.uR[pbase + i], (double)a.vR[pbase + i], a.thresh))
            ++ref;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        ref += __shfl_xor(ref, off);
    if (lane == 0)
        part[wave][0] = (double)ref;
    __syncthreads();
    if (tid == 0)
        a.ref_count[set] = (int)(part[0][0] + part[1][0] + part[2][0] + part[3][0]);
}

// Final choice (utils/utils.py:453-463): the refined pose replaces the winner when it has at least
// as many supporters; the reported ratio is always the refined pose's.  found = 0 when no
// hypothesis had a supporter (the reference returns None).
__global__ void __launch_bounds__(256) adopt_kernel(int n_sets, const int32_t *n_dev, int n_host, int cap,
                                                    const int32_t *best_count, const int32_t *ref_count,
                                                    const double *best_Rt, const double *ref_Rt, double *out_Rt,
                                                    double *ratio, int32_t *found)
{
    const int set = blockIdx.x * 256 + threadIdx.x;
    if (set >= n_sets)
        return;
    int n = n_dev ? n_dev[set] : n_host;
    n = min(max(n, 0), cap);
    const bool ok = n >= 4 && best_count[set] > 0;
    const bool take_ref = ref_count[set] >= best_count[set];
    for (int k = 0; k < 12; ++k)
        out_Rt[(size_t)set * 12 + k] =
            ok ? (take_ref ? ref_Rt : best_Rt)[(size_t)set * 12 + k] : __builtin_nan("");
    ratio[set] = ok ? (double)ref_count[set] / (double)n : __builtin_nan("");
    found[set] = ok ? 1 : 0;
}

}  // namespace
}  // namespace slam

using namespace slam;

extern "C" int slam_ransac_sample(uint64_t seed, int n_sets, int H, const int32_t *n_dev, int n_host,
                                  int32_t *samples, void *stream)
{
    if (n_sets < 0 || H < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_ransac_sample: negative size");
    if (n_sets == 0 || H == 0)
        return SLAM_OK;
    if (!samples)
        return fail(SLAM_ERR_BAD_ARG, "slam_ransac_sample: null buffer");
    if (reinterpret_cast<uintptr_t>(samples) & 15)
        return fail(SLAM_ERR_BAD_ARG, "slam_ransac_sample: samples must be 16-byte aligned");
    const long long total = (long long)n_sets * H;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7FFFFFFFLL)
        return fail(SLAM_ERR_BAD_ARG, "slam_ransac_sample: grid too large");
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, seed, n_sets, H, n_dev,
                       n_host, samples);
    return check_launch("slam_ransac_sample");
}

static bool pinhole(const double *K)
{
    bool ok = K[1] == 0.0 && K[3] == 0.0 && K[6] == 0.0 && K[7] == 0.0 && K[8] == 1.0 && K[0] != 0.0 && K[4] != 0.0;
    for (int i = 0; i < 9; ++i)
        ok = ok && (K[i] - K[i] == 0.0);
    return ok;
}

extern "C" int slam_p3p_hypotheses(const double *K, const double *X, const double *Y, const double *Z,
                                   const float *uL, const float *vL, int cap, const int32_t *n_dev, int n_host,
                                   const int32_t *samples, int n_sets, int H_total, int h_begin, int H,
                                   const int32_t *skip, double *Rt, void *stream)
{
    if (n_sets < 0 || H < 0 || cap < 0 || h_begin < 0 || H_total < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_p3p_hypotheses: negative size");
    if ((long long)h_begin + H > H_total)
        return fail(SLAM_ERR_BAD_ARG, "slam_p3p_hypotheses: range [%d, %d) exceeds %d hypotheses per set", h_begin,
                    h_begin + H, H_total);
    if (!K)
        return fail(SLAM_ERR_BAD_ARG, "slam_p3p_hypotheses: null intrinsics");
    if (!pinhole(K))
        return fail(SLAM_ERR_BAD_ARG, "slam_p3p_hypotheses: K must be a finite pinhole matrix [fx 0 cx; 0 fy cy; 0 0 1]");
    if (n_sets == 0 || H == 0)
        return SLAM_OK;
    if (!samples || !Rt || (cap > 0 && (!X || !Y || !Z || !uL || !vL)))
        return fail(SLAM_ERR_BAD_ARG, "slam_p3p_hypotheses: null buffer");
    if ((reinterpret_cast<uintptr_t>(samples) | reinterpret_cast<uintptr_t>(Rt)) & 15)
        return fail(SLAM_ERR_BAD_ARG, "slam_p3p_hypotheses: samples and Rt must be 16-byte aligned");
    P3PArgs a;
    a.fx = K[0];
    a.fy = K[4];
    a.cx = K[2];
    a.cy = K[5];
    a.X = X;
    a.Y = Y;
    a.Z = Z;
    a.uL = uL;
    a.vL = vL;
    a.cap = cap;
    a.n_dev = n_dev;
    a.n_host = n_host;
    a.samples = samples;
    a.n_sets = n_sets;
    a.H = H;
    a.Hs = H_total;
    a.h0 = h_begin;
    a.skip = skip;
    a.Rt = Rt;
    const long long total = (long long)n_sets * H;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7FFFFFFFLL)
        return fail(SLAM_ERR_BAD_ARG, "slam_p3p_hypotheses: grid too large");
    hipLaunchKernelGGL(p3p_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("slam_p3p_hypotheses");
}

extern "C" int slam_ransac_select(const int32_t *counts, const double *Rt, int n_sets, int H, int n_avail,
                                  const int32_t *n_dev, int n_host, int cap, int max_iterations, int32_t *best,
                                  int32_t *best_count, double *best_Rt, int32_t *iters, int32_t *stopped,
                                  void *stream)
{
    if (n_sets < 0 || H < 0 || max_iterations < 0 || n_avail < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_ransac_select: negative size");
    if (n_sets == 0)
        return SLAM_OK;
    if (!counts || !best || !best_count || (best_Rt && !Rt))
        return fail(SLAM_ERR_BAD_ARG, "slam_ransac_select: null buffer");
    hipLaunchKernelGGL(ransac_select_kernel, dim3(n_sets), dim3(256), 0, (hipStream_t)stream, counts, Rt, H, n_avail,
                       n_dev, n_host, cap, max_iterations, best, best_count, best_Rt, iters, stopped);
    return check_launch("slam_ransac_select");
}

namespace {
int launch_refine(const char *who, const double *K, const double *t_right, double thresh, const double *X,
                  const double *Y, const double *Z, const float *uL, const float *vL, const float *uR, const float *vR,
                  int cap, const int32_t *n_dev, int n_host, const uint64_t *masks, int mask_words, const double *Rt0,
                  int n_sets, int iterations, double *Rt_out, int32_t *ok, int32_t *ref_count, bool stereo,
                  void *stream)
{
    if (n_sets < 0 || cap < 0 || iterations < 0)
        return fail(SLAM_ERR_BAD_ARG, "%s: negative size", who);
    if (!K)
        return fail(SLAM_ERR_BAD_ARG, "%s: null intrinsics", who);
    if (!pinhole(K))
        return fail(SLAM_ERR_BAD_ARG, "%s: K must be a finite pinhole matrix [fx 0 cx; 0 fy cy; 0 0 1]", who);
    if (n_sets == 0)
        return SLAM_OK;
    if (!Rt0 || !Rt_out || (cap > 0 && (!X || !Y || !Z || !uL || !vL)))
        return fail(SLAM_ERR_BAD_ARG, "%s: null buffer", who);
    RefineArgs a;
    a.fx = K[0];
    a.fy = K[4];
    a.cx = K[2];
    a.cy = K[5];
    for (int i = 0; i < 9; ++i)
        a.K.m[i] = K[i];
    for (int i = 0; i < 3; ++i)
        a.tr.v[i] = t_right ? t_right[i] : 0.0;
    a.thresh = thresh;
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
    a.masks = reinterpret_cast<const unsigned long long *>(masks);
    a.mask_words = mask_words;
    a.Rt0 = Rt0;
    a.Rt_out = Rt_out;
    a.ok = ok;
    a.ref_count = ref_count;
    a.iterations = iterations;
    if (stereo)
        hipLaunchKernelGGL(refine_kernel<true>, dim3(n_sets), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(refine_kernel<false>, dim3(n_sets), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch(who);
}
}  // namespace

extern "C" int slam_refine_poses(const double *K, const double *X, const double *Y, const double *Z, const float *uL,
                                 const float *vL, int cap, const int32_t *n_dev, int n_host, const uint64_t *masks,
                                 int mask_words, const double *Rt0, int n_sets, int iterations, double *Rt_out,
                                 int32_t *ok, void *stream)
{
    if (n_sets > 0 && !masks)
        return fail(SLAM_ERR_BAD_ARG, "slam_refine_poses: null buffer");
    if (n_sets > 0 && cap >= 0 && mask_words < (cap + 63) / 64)
        return fail(SLAM_ERR_BAD_ARG, "slam_refine_poses: mask_words %d < ceil(cap/64)", mask_words);
    return launch_refine("slam_refine_poses", K, nullptr, 0.0, X, Y, Z, uL, vL, nullptr, nullptr, cap, n_dev, n_host,
                         masks, mask_words, Rt0, n_sets, iterations, Rt_out, ok, nullptr, false, stream);
}

extern "C" int slam_refine_winners(const double *K, const double *t_right, double thresh, const double *X,
                                   const double *Y, const double *Z, const float *uL, const float *vL, const float *uR,
                                   const float *vR, int cap, const int32_t *n_dev, int n_host, const double *Rt0,
                                   int n_sets, int iterations, double *Rt_out, int32_t *ok, int32_t *ref_count,
                                   void *stream)
{
    if (n_sets > 0 && (!t_right || !ref_count || (cap > 0 && (!uR || !vR))))
        return fail(SLAM_ERR_BAD_ARG, "slam_refine_winners: null buffer");
    return launch_refine("slam_refine_winners", K, t_right, thresh, X, Y, Z, uL, vL, uR, vR, cap, n_dev, n_host, nullptr,
                         0, Rt0, n_sets, iterations, Rt_out, ok, ref_count, false, stream);
}

extern "C" int slam_refine_winners_stereo(const double *K, const double *t_right, double thresh, const double *X,
                                          const double *Y, const double *Z, const float *uL, const float *vL,
                                          const float *uR, const float *vR, int cap, const int32_t *n_dev, int n_host,
                                          const double *Rt0, int n_sets, int iterations, double *Rt_out, int32_t *ok,
                                          int32_t *ref_count, void *stream)
{
    if (n_sets > 0 && (!t_right || !ref_count || (cap > 0 && (!uR || !vR))))
        return fail(SLAM_ERR_BAD_ARG, "slam_refine_winners_stereo: null buffer");
    return launch_refine("slam_refine_winners_stereo", K, t_right, thresh, X, Y, Z, uL, vL, uR, vR, cap, n_dev, n_host,
                         nullptr, 0, Rt0, n_sets, iterations, Rt_out, ok, ref_count, true, stream);
}

extern "C" int slam_ransac_adopt(int n_sets, const int32_t *n_dev, int n_host, int cap, const int32_t *best_count,
                                 const int32_t *ref_count, const double *best_Rt, const double *ref_Rt, double *out_Rt,
                                 double *ratio, int32_t *found, void *stream)
{
    if (n_sets < 0)
        return fail(SLAM_ERR_BAD_ARG, "slam_ransac_adopt: negative size");
    if (n_sets == 0)
        return SLAM_OK;
    if (!best_count || !ref_count || !best_Rt || !ref_Rt || !out_Rt || !ratio || !found)
        return fail(SLAM_ERR_BAD_ARG, "slam_ransac_adopt: null buffer");
    hipLaunchKernelGGL(adopt_kernel, dim3((n_sets + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_sets, n_dev, n_host,
                       cap, best_count, ref_count, best_Rt, ref_Rt, out_Rt, ratio, found);
    return check_launch("slam_ransac_adopt");
}
