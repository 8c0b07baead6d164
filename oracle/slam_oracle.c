/*
 * CPU oracle for the stereo-SLAM front-end hot path.  TEST INFRASTRUCTURE ONLY:
 * it is the checker the HIP kernels are compared against (tests/, the smoke
 * check and the cpu_baseline leg of bench.py).  Nothing in the product path may
 * call into this file.
 *
 * PARITY STATUS.  The reference ships no tests, fixtures or golden vectors for this
 * path.  The functions below restate the reference's own Python (utils/utils.py,
 * models/Matcher.py, models/LoopClosure.py) operation by operation; ratio test, row
 * filter, triangulation, join and supporter counts are PINNED: tests/test_reference_golden.py
 * holds them against tests/golden/reference_golden.npz, the outputs of the reference's own
 * function bodies executed in the build container (tests/golden/make_reference_golden.py).
 * "Parity unpinned" remains for oracle_hamming_knn / oracle_l2_knn, which restate OpenCV's
 * published BFMatcher::knnMatch behaviour (opencv-python~=4.7.0.72, requirements.txt:4, is not
 * installed here), and for oracle_ransac_sample, which has no counterpart in the reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: every fp64 expression here is evaluated with one
 * IEEE rounding per operation, in the order written, and the HIP kernels are
 * compiled the same way, so inlier decisions agree bit for bit.
 */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_version(void) { return 1; }

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* a1: cv2.BFMatcher(NORM_HAMMING).knnMatch as called at
 * models/Matcher.py:127,140,199,219.
 *
 * OpenCV's brute-force matcher scans the train rows in index order and keeps,
 * per query, a sorted list of the K best (distance, index) pairs: a candidate
 * is admitted only if its distance is strictly smaller than the current K-th
 * best, and it is inserted in front of every kept entry whose distance is
 * strictly larger.  So ties keep the lower train index first and a later equal
 * distance never displaces an earlier one.  Distances are popcount(XOR) over
 * desc_bytes bytes, accumulated as int.
 *
 * Output has k columns per query; columns beyond min(k, nt) hold idx -1 and
 * dist INT_MAX (cv2 returns a shorter list there).
 */
static inline int hamming_bytes(const uint8_t *a, const uint8_t *b, int n)
{
    int d = 0, i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x, y;
        memcpy(&x, a + i, 8);
        memcpy(&y, b + i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    for (; i < n; ++i)
        d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}

void oracle_hamming_knn(const uint8_t *q, int nq, int q_stride,
                        const uint8_t *t, int nt, int t_stride,
                        int desc_bytes, int k,
                        int32_t *out_idx, int32_t *out_dist, int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int i = 0; i < nq; ++i) {
        int32_t bd[8], bi[8];
        const int K = k > 8 ? 8 : k;
        for (int c = 0; c < K; ++c) {
            bd[c] = INT_MAX;
            bi[c] = -1;
        }
        const uint8_t *qi = q + (size_t)i * q_stride;
        for (int j = 0; j < nt; ++j) {
            int d = hamming_bytes(qi, t + (size_t)j * t_stride, desc_bytes);
            if (d < bd[K - 1]) {
                int p = K - 1;
                while (p > 0 && bd[p - 1] > d) {
                    bd[p] = bd[p - 1];
                    bi[p] = bi[p - 1];
                    --p;
                }
                bd[p] = d;
                bi[p] = j;
            }
        }
        for (int c = 0; c < K; ++c) {
            out_idx[(size_t)i * k + c] = bi[c];
            out_dist[(size_t)i * k + c] = bd[c];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* a2: Matcher.apply_threshold / Matcher.apply_thresholds
 * (models/Matcher.py:148-160,178-184): keep iff m.distance < thr * n.distance
 * as Python floats, i.e. double(float32(d1)) < thr * double(float32(d2)).
 * Hamming distances are small integers, exact in float32.  A query with fewer
 * than two neighbours is not kept here (the reference raises on it). */
void oracle_ratio_filter(const int32_t *idx, const int32_t *dist, int n,
                         double thr, uint8_t *keep)
{
    for (int i = 0; i < n; ++i) {
        if (idx[2 * i] < 0 || idx[2 * i + 1] < 0) {
            keep[i] = 0;
            continue;
        }
        double d1 = (double)(float)dist[2 * i];
        double d2 = (double)(float)dist[2 * i + 1];
        keep[i] = d1 < thr * d2;
    }
}

/* a3: _rectified_stereo_classifier (utils/utils.py:89-100) on the k=1 match
 * list gathered by coords_from_kps (utils/utils.py:65-72): inlier iff
 * |y1 - y2| <= thresh in float64 on float32-origin coordinates. */
void oracle_row_filter(const int32_t *idx, int idx_stride, int n,
                       const float *yq, const float *yt, double thr,
                       uint8_t *keep)
{
    for (int i = 0; i < n; ++i) {
        int j = idx[(size_t)i * idx_stride];
        if (j < 0) {
            keep[i] = 0;
            continue;
        }
        double dev = fabs((double)yq[i] - (double)yt[j]);
        keep[i] = dev <= thr;
    }
}

/* ------------------------------------------------------------------------- */
/* a4: least_squares (utils/utils.py:527-560) driven by get_3d_points_cloud
 * (utils/utils.py:318-335).
 *
 * A = [x1*P3 - P1; y1*P3 - P2; x2*Q3 - Q1; y2*Q3 - Q2]; the solution is the
 * right singular vector of the smallest singular value, dehomogenised.  The
 * reference takes it from LAPACK (np.linalg.svd); here it is found by a
 * one-sided Jacobi (Hestenes) iteration on the columns of A, which is what the
 * HIP kernel runs per lane (sums as fma chains, every operation rounded once:
 * the same bits on the CPU and on the GPU).  The numpy twin (oracle/numpy_twin.py) keeps the
 * LAPACK call and is the check on this restatement (agreement ~1e-10
 * relative; the contract is 1e-4).
 *
 * valid[i] = !(Z <= 0 || x1 < x2)   (utils/utils.py:326-327).
 */
#define DLT_MAX_SWEEPS 30
#define DLT_TOL2 1e-30 /* (|gamma|)^2 <= tol2*alpha*beta : columns orthogonal */

void oracle_dlt_point(const double *P, const double *Q, double x1, double y1,
                      double x2, double y2, double out[3])
{
    double a[4][4], v[4][4];
    for (int c = 0; c < 4; ++c) {
        a[0][c] = x1 * P[8 + c] - P[c];
        a[1][c] = y1 * P[8 + c] - P[4 + c];
        a[2][c] = x2 * Q[8 + c] - Q[c];
        a[3][c] = y2 * Q[8 + c] - Q[4 + c];
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            v[r][c] = (r == c) ? 1.0 : 0.0;

    for (int sweep = 0; sweep < DLT_MAX_SWEEPS; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 3; ++p) {
            for (int q = p + 1; q < 4; ++q) {
                const double alpha = fma(a[3][p], a[3][p], fma(a[2][p], a[2][p], fma(a[1][p], a[1][p], a[0][p] * a[0][p])));
                const double beta = fma(a[3][q], a[3][q], fma(a[2][q], a[2][q], fma(a[1][q], a[1][q], a[0][q] * a[0][q])));
                const double gamma = fma(a[3][p], a[3][q], fma(a[2][p], a[2][q], fma(a[1][p], a[1][q], a[0][p] * a[0][q])));
                if (gamma * gamma <= DLT_TOL2 * (alpha * beta))
                    continue;
                rotated = 1;
                /* The Jacobi rotation with tan = g / D, d = beta - alpha, g = 2 gamma,
                 * D = d + sign(d) * hypot(d, g) (the smaller root), written so that it
                 * costs two square roots and one division:
                 * (cs, sn) = (|D|, sign(D) * g) / hypot(D, g). */
                const double d = beta - alpha, g = 2.0 * gamma;
                const double gg = g * g;
                const double r = sqrt(fma(d, d, gg));
                const double D = d + copysign(r, d);
                const double inv = 1.0 / sqrt(fma(D, D, gg));
                const double cs = fabs(D) * inv;
                const double sn = (D < 0.0 ? -g : g) * inv;
                for (int r4 = 0; r4 < 4; ++r4) {
                    const double ap = a[r4][p], aq = a[r4][q];
                    a[r4][p] = fma(cs, ap, -(sn * aq));
                    a[r4][q] = fma(sn, ap, cs * aq);
                    const double vp = v[r4][p], vq = v[r4][q];
                    v[r4][p] = fma(cs, vp, -(sn * vq));
                    v[r4][q] = fma(sn, vp, cs * vq);
                }
            }
        }
        if (!rotated)
            break;
    }
    int jmin = 0;
    double nmin = 0.0;
    for (int c = 0; c < 4; ++c) {
        const double n2 = fma(a[3][c], a[3][c], fma(a[2][c], a[2][c], fma(a[1][c], a[1][c], a[0][c] * a[0][c])));
        if (c == 0 || n2 < nmin) {
            nmin = n2;
            jmin = c;
        }
    }
    double w = v[3][jmin];
    out[0] = v[0][jmin] / w;
    out[1] = v[1][jmin] / w;
    out[2] = v[2][jmin] / w;
}

void oracle_triangulate_dlt(const double *P, const double *Q,
                            const float *x1, const float *y1,
                            const float *x2, const float *y2, int n,
                            double *X, double *Y, double *Z, uint8_t *valid,
                            int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int i = 0; i < n; ++i) {
        double o[3];
        oracle_dlt_point(P, Q, (double)x1[i], (double)y1[i], (double)x2[i],
                         (double)y2[i], o);
        X[i] = o[0];
        Y[i] = o[1];
        Z[i] = o[2];
        valid[i] = !(o[2] <= 0.0 || x1[i] < x2[i]);
    }
}

/* ------------------------------------------------------------------------- */
/* a5: get_matches_that_appear_in_all_4_images (utils/utils.py:338-386) and
 * its twin find_4_images_matches (models/LoopClosure.py:71-109), on dense
 * per-keypoint tables instead of dicts:
 *   prev_map[q]  = right keypoint matched to left keypoint q of the earlier
 *                  frame, or -1 (array_to_dict of the stereo inlier mapping)
 *   prev_valid[q]= q has a kept 3-D point (ind_to_3d_point_prev membership)
 *   cur_map[t]   = likewise for the later frame
 * For every kept temporal match (q -> t) in query order, emit
 * (q, prev_map[q], t, cur_map[t]) iff all three memberships hold.
 * Returns the number of rows written. */
int oracle_join(const int32_t *t_idx, int t_stride, const uint8_t *t_keep,
                int nq, const int32_t *prev_map, const uint8_t *prev_valid,
                const int32_t *cur_map, int n_cur, int32_t *out_rows /*[nq,4]*/)
{
    int m = 0;
    for (int q = 0; q < nq; ++q) {
        if (!t_keep[q])
            continue;
        int t = t_idx[(size_t)q * t_stride];
        if (t < 0 || t >= n_cur)
            continue;
        if (prev_map[q] < 0 || cur_map[t] < 0 || !prev_valid[q])
            continue;
        out_rows[4 * m + 0] = q;
        out_rows[4 * m + 1] = prev_map[q];
        out_rows[4 * m + 2] = t;
        out_rows[4 * m + 3] = cur_map[t];
        ++m;
    }
    return m;
}

/* ------------------------------------------------------------------------- */
/* a6: find_supporters (utils/utils.py:389-410) for H hypotheses.
 *
 * Per (hypothesis, match):  L = Rt @ [X,1];  R = L + t_right;
 * pl = (K@L)[:2]/(K@L)[2];  pr likewise;  dl = ||uvL - pl||, dr = ||uvR - pr||;
 * supporter iff dl < thresh and dr < thresh (strict).  No cheirality test.
 *
 * numpy's own summation order inside '@' is BLAS-defined; this file fixes it:
 * every 3- or 4-term dot product is accumulated left to right, one rounding per
 * multiply and per add, and the norm is sqrt(dx*dx + dy*dy).
 *
 * Layouts: Rt [H,12] row-major 3x4; K[9]; t_right[3] (= M2[:,3]);
 * X,Y,Z [N] f64; uL,vL,uR,vR [N] f32 (float32-origin keypoint coordinates);
 * counts [H]; masks (nullable) [H, ceil(N/64)] u64, bit i%64 of word i/64.
 */
static inline int support_one(const double *Rt, const double *K,
                              const double *tr, double X, double Y, double Z,
                              double uL, double vL, double uR, double vR,
                              double thresh)
{
    double L0 = ((Rt[0] * X + Rt[1] * Y) + Rt[2] * Z) + Rt[3] * 1.0;
    double L1 = ((Rt[4] * X + Rt[5] * Y) + Rt[6] * Z) + Rt[7] * 1.0;
    double L2 = ((Rt[8] * X + Rt[9] * Y) + Rt[10] * Z) + Rt[11] * 1.0;
    double R0 = L0 + tr[0], R1 = L1 + tr[1], R2 = L2 + tr[2];
    double hl0 = (K[0] * L0 + K[1] * L1) + K[2] * L2;
    double hl1 = (K[3] * L0 + K[4] * L1) + K[5] * L2;
    double hl2 = (K[6] * L0 + K[7] * L1) + K[8] * L2;
    double hr0 = (K[0] * R0 + K[1] * R1) + K[2] * R2;
    double hr1 = (K[3] * R0 + K[4] * R1) + K[5] * R2;
    double hr2 = (K[6] * R0 + K[7] * R1) + K[8] * R2;
    double plx = hl0 / hl2, ply = hl1 / hl2;
    double prx = hr0 / hr2, pry = hr1 / hr2;
    double dlx = uL - plx, dly = vL - ply;
    double drx = uR - prx, dry = vR - pry;
    double dl = sqrt(dlx * dlx + dly * dly);
    double dr = sqrt(drx * drx + dry * dry);
    return (dl < thresh) && (dr < thresh);
}

void oracle_score_hypotheses(const double *Rt, int H, const double *K,
                             const double *t_right, const double *X,
                             const double *Y, const double *Z, const float *uL,
                             const float *vL, const float *uR, const float *vR,
                             int N, double thresh, int32_t *counts,
                             uint64_t *masks, int threads)
{
    (void)threads;
    const int words = (N + 63) / 64;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int h = 0; h < H; ++h) {
        const double *rt = Rt + (size_t)h * 12;
        int cnt = 0;
        if (masks)
            for (int w = 0; w < words; ++w)
                masks[(size_t)h * words + w] = 0;
        for (int i = 0; i < N; ++i) {
            int s = support_one(rt, K, t_right, X[i], Y[i], Z[i],
                                (double)uL[i], (double)vL[i], (double)uR[i],
                                (double)vR[i], thresh);
            cnt += s;
            if (masks && s)
                masks[(size_t)h * words + (i >> 6)] |= (uint64_t)1 << (i & 63);
        }
        counts[h] = cnt;
    }
}

/* ---- minimal samples for the batched RANSAC -------------------------------
 * Stand-in for random.sample(points_to_choose_from, k=4) (utils/utils.py:427)
 * in the batched path, where every (set, hypothesis) owns a splitmix64 stream:
 * draw k is a uniform rank among the n - k positions not taken yet.
 */
static uint64_t splitmix64_next(uint64_t *s)
{
    *s += 0x9E3779B97F4A7C15ull;
    uint64_t z = *s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void oracle_ransac_sample(uint64_t seed, int n_sets, int H, const int32_t *n,
                          int32_t *samples)
{
    for (int set = 0; set < n_sets; ++set) {
        for (int h = 0; h < H; ++h) {
            int32_t *out = samples + 4 * ((size_t)set * H + h);
            if (n[set] < 4) {
                out[0] = out[1] = out[2] = out[3] = -1;
                continue;
            }
            uint64_t s = seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(set + 1)) ^
                         (0x8CB92BA72F3D8DD7ull * (uint64_t)(h + 1));
            int taken[4];
            for (int k = 0; k < 4; ++k) {
                uint64_t r = splitmix64_next(&s);
                int j = (int)(((r >> 32) * (uint64_t)(n[set] - k)) >> 32);
                /* rank among the free positions -> position */
                for (int m = 0; m < k; ++m)
                    if (j >= taken[m])
                        ++j;
                out[k] = j;
                int pos = k;
                while (pos > 0 && taken[pos - 1] > j) {
                    taken[pos] = taken[pos - 1];
                    --pos;
                }
                taken[pos] = j;
            }
        }
    }
}

/* ---- float L2 k-NN ------------------------------------------------------------
 * cv2.BFMatcher(NORM_L2).knnMatch on float32 descriptors -- the matcher the
 * reference builds by default (models/Matcher.py:15), same scan and tie rule as
 * the Hamming one above.  Squared distances are summed in float32 in dimension
 * order with one rounding per dimension (fmaf); OpenCV sums the same terms in
 * SIMD lanes, which gives the same value whenever every partial sum is exactly
 * representable -- always for SIFT's integer-valued descriptors.  The reported
 * distance is sqrtf of the sum; missing neighbours hold idx -1, dist FLT_MAX.
 */
void oracle_l2_knn(const float *q, int nq, const float *t, int nt, int dim, int k,
                   int32_t *out_idx, float *out_dist, int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int i = 0; i < nq; ++i) {
        float bd[8];
        int32_t bi[8];
        const int K = k > 8 ? 8 : k;
        for (int c = 0; c < K; ++c) {
            bd[c] = FLT_MAX;
            bi[c] = -1;
        }
        const float *qi = q + (size_t)i * dim;
        for (int j = 0; j < nt; ++j) {
            const float *tj = t + (size_t)j * dim;
            float d = 0.0f;
            for (int c = 0; c < dim; ++c) {
                const float e = qi[c] - tj[c];
                d = fmaf(e, e, d);
            }
            if (d < bd[K - 1]) {
                int p = K - 1;
                while (p > 0 && bd[p - 1] > d) {
                    bd[p] = bd[p - 1];
                    bi[p] = bi[p - 1];
                    --p;
                }
                bd[p] = d;
                bi[p] = j;
            }
        }
        for (int c = 0; c < K; ++c) {
            out_idx[(size_t)i * k + c] = bi[c];
            out_dist[(size_t)i * k + c] = bi[c] >= 0 ? sqrtf(bd[c]) : FLT_MAX;
        }
    }
}
