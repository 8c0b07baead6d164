/*
 * slam_amd -- C ABI of the MI355X (gfx950) stereo-SLAM front-end kernels.
 *
 * Every pointer marked "dev" is a device pointer (e.g. tensor.data_ptr() of a
 * PyTorch-ROCm tensor); "host" pointers are read on the calling thread before
 * the call returns.  `stream` is a hipStream_t (0 = the null stream).  Calls
 * are asynchronous on that stream, allocate no device memory, never
 * synchronise the device, and never throw: they return SLAM_OK or a negative
 * code, with a message available from slam_last_error() on the same thread.
 * The caller owns every buffer.  Calls on distinct streams may overlap.
 *
 * "Reference" below is the Python project this library slots under
 * (TzlilOvadia/SLAM); file:line citations are into that tree.
 */
#ifndef SLAM_AMD_H
#define SLAM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_OK 0
#define SLAM_ERR_BAD_ARG (-1)
#define SLAM_ERR_LAUNCH (-2)
#define SLAM_ERR_UNSUPPORTED (-3)

/* Row width of one entry of a segment table (see slam_hamming_match). */
#define SLAM_SEG_STRIDE 5
/* Row width of one entry of a frame-pair table (see slam_join_consensus). */
#define SLAM_PAIR_STRIDE 8

/* Largest train-set size of one matching (the packed (distance,index) key keeps
 * 22 bits for the index).  cv2's BFMatcher itself stops at 2^18 per image. */
#define SLAM_MAX_TRAIN_ROWS (1 << 22)

int slam_version(void);
const char *slam_last_error(void);

/*
 * Kernel variants with identical results, kept for cross-checks and A/B timing (no counterpart in
 * the reference).  Process-wide; -1 leaves a choice as it is.  The defaults come from the
 * environment variables SLAM_KNN_IMPL (fp4 | valu) and SLAM_SCORE_IMPL (auto | f64 | split | wide | mfma),
 * read once when the library is loaded.
 *   knn:   SLAM_KNN_FP4  Hamming distance on the fp4 matrix cores (default)
 *          SLAM_KNN_VALU xor + popcount on the vector ALU
 *   score: SLAM_SCORE_AUTO  fp32 first-level filter, form chosen from the launch size (default)
 *          SLAM_SCORE_F64   fp64-only filter
 *          SLAM_SCORE_SPLIT / SLAM_SCORE_WIDE  force one form of the fp32 first-level kernel
 *          SLAM_SCORE_MFMA  fp32 first level with the rigid transform on the fp32 matrix cores
 */
#define SLAM_KNN_FP4 0
#define SLAM_KNN_VALU 1
#define SLAM_SCORE_AUTO 0
#define SLAM_SCORE_F64 1
#define SLAM_SCORE_SPLIT 2
#define SLAM_SCORE_WIDE 3
#define SLAM_SCORE_MFMA 4
int slam_select_kernels(int knn, int score);

/*
 * Repack binary descriptors [n, src_stride] (first desc_bytes bytes of each row
 * are the descriptor, e.g. 61 for AKAZE, 32 for ORB) into rows of dst_stride
 * bytes (32 or 64), zero padded.  This is the layout the matcher reads.
 * Replaces: the numpy descriptor arrays cv2 hands to knnMatch
 * (models/Matcher.py:116-119) -- no arithmetic, layout only.
 */
int slam_pack_descriptors(const uint8_t *src_dev, int n, int desc_bytes,
                          int src_stride, uint8_t *dst_dev, int dst_stride,
                          void *stream);

/*
 * Batched brute-force Hamming k-NN (k = 1 or 2) with the fused match filters.
 *
 * Replaces, per batch item:
 *   cv2.BFMatcher(NORM_HAMMING).knnMatch(q, t, k)  models/Matcher.py:127,140,199,219
 *   mode 1: Matcher.apply_thresholds ratio test     models/Matcher.py:148-160,178-184
 *   mode 2: rectified-stereo row filter             utils/utils.py:89-116 (via :302-315)
 *
 * seg [batch, 5] int32 dev: per item (q_start, q_count, t_start, t_count,
 * out_start), all in rows.  Item b matches rows q[q_start .. +q_count) against
 * rows t[t_start .. +t_count) and writes result rows out_start + local query
 * index.  q and t may alias (temporal matching within one descriptor table).
 *
 * Descriptor rows are stride_bytes (32 or 64) apart, 16-byte aligned; only the
 * first desc_bytes bytes of a row take part (pad bytes are ignored, they need
 * not be zero).
 *
 * out_idx, out_dist [rows, 2] int32: train index local to the item (0-based
 * within its t range) and Hamming distance, best first.  Ties go to the lower
 * train index, and the second entry is the next in (distance, index) order,
 * exactly as cv2's scan produces.  Missing neighbours (t_count < 2, or k = 1
 * for column 1) hold idx -1, dist INT32_MAX.
 *
 * mode 0: keep_dev may be NULL.
 * mode 1 (needs k = 2): keep[row] = idx1 >= 0 && (double)d0 < thr * (double)d1.
 * mode 2: keep[row] = idx0 >= 0 && fabs((double)yq[q_start+i] -
 *         (double)yt[t_start+idx0]) <= thr; yq/yt are float32 keypoint rows
 *         indexed like the descriptor rows.
 *
 * max_nq / max_nt: upper bounds on q_count / t_count over the batch (the
 * library cannot read seg without a sync); max_nt <= SLAM_MAX_TRAIN_ROWS.
 */
int slam_hamming_match(const uint8_t *q_dev, const uint8_t *t_dev,
                       const int32_t *seg_dev, int batch, int max_nq,
                       int max_nt, int desc_bytes, int stride_bytes, int k,
                       int mode, double thr, const float *yq_dev,
                       const float *yt_dev, int32_t *out_idx_dev,
                       int32_t *out_dist_dev, uint8_t *keep_dev, void *stream);

/* The unfiltered form: same as slam_hamming_match(..., mode 0, ...). */
int slam_hamming_knn2(const uint8_t *q_dev, const uint8_t *t_dev,
                      const int32_t *seg_dev, int batch, int max_nq, int max_nt,
                      int desc_bytes, int stride_bytes, int k,
                      int32_t *out_idx_dev, int32_t *out_dist_dev, void *stream);

/*
 * Float L2 k-NN with the same segment table, modes and tie rule: replaces
 *   cv2.BFMatcher().knnMatch(q, t, k)   -- NORM_L2 on float32 descriptors, the matcher and
 *   descriptor (SIFT, 128 floats) the reference constructs by default, models/Matcher.py:15,
 *   called at models/Matcher.py:127,140,199,219.
 * q, t: float32 rows of `dim` (128 or 64) values, 16-byte aligned, rows back to back.
 * out_dist [rows, 2] float32 = sqrtf(sum_k (q_k - t_k)^2), the sum taken in fp32 in dimension
 * order (exact, hence identical to cv2's, for integer-valued descriptors such as SIFT's).
 * Missing neighbours hold idx -1, dist FLT_MAX.  mode 1 compares the float32 distances as
 * Python does: (double)d0 < thr * (double)d1.
 */
int slam_l2_match(const float *q_dev, const float *t_dev, const int32_t *seg_dev,
                  int batch, int max_nq, int max_nt, int dim, int k, int mode,
                  double thr, const float *yq_dev, const float *yt_dev,
                  int32_t *out_idx_dev, float *out_dist_dev, uint8_t *keep_dev,
                  void *stream);

/*
 * The same matcher for integer-valued descriptors -- what cv2's SIFT emits (float32 arrays holding
 * integers 0..255; models/Matcher.py:15,111-119) -- on the int8 matrix cores.  Two steps:
 *
 * slam_l2_pack_int8: float32 [n, dim] -> int8 [n, dim] holding value - 128, norm2[n] = the squared
 *   length of the shifted row, and *bad += the number of entries that are not integers in 0..255
 *   (the caller zeroes *bad beforehand; a table with *bad != 0 must go through slam_l2_match).
 *   Done once per descriptor table, like slam_pack_descriptors for the binary ones.
 * slam_l2_match_int8: arguments as slam_l2_match, with the packed tables and their norms in
 *   place of the float tables.  Indices, distances (sqrtf of the exact integer squared distance,
 *   which is what cv2's float32 accumulation yields for such data) and keep flags are identical
 *   to slam_l2_match on the float tables.
 */
int slam_l2_pack_int8(const float *desc_dev, int n, int dim, int8_t *out_dev,
                      int32_t *norm2_dev, int32_t *bad_dev, void *stream);
int slam_l2_match_int8(const int8_t *q_dev, const int32_t *q_norm2_dev,
                       const int8_t *t_dev, const int32_t *t_norm2_dev,
                       const int32_t *seg_dev, int batch, int max_nq, int max_nt,
                       int dim, int k, int mode, double thr,
                       const float *yq_dev, const float *yt_dev,
                       int32_t *out_idx_dev, float *out_dist_dev, uint8_t *keep_dev,
                       void *stream);

/*
 * The two match filters on their own, for callers that already hold match lists.
 * slam_ratio_filter replaces Matcher.apply_threshold / apply_thresholds
 * (models/Matcher.py:148-160,178-184): idx, dist [n,2] int32 dev as written by
 * slam_hamming_match; keep[i] = both neighbours present && (double)d0 < thr*(double)d1.
 * slam_row_filter replaces _rectified_stereo_classifier (utils/utils.py:89-100):
 * y1, y2 [n] float32 dev, already gathered per match; keep[i] = |y1-y2| <= thr in fp64.
 */
int slam_ratio_filter(const int32_t *idx_dev, const int32_t *dist_dev, int n,
                      double thr, uint8_t *keep_dev, void *stream);
int slam_row_filter(const float *y1_dev, const float *y2_dev, int n, double thr,
                    uint8_t *keep_dev, void *stream);

/*
 * Linear (DLT) triangulation of n stereo correspondences.
 * Replaces: least_squares (utils/utils.py:527-560) as driven per point by
 * get_3d_points_cloud (utils/utils.py:318-335).
 *
 * P, Q: host, 12 doubles each, row-major 3x4 (= k @ m1, k @ m2 at :325).
 * x1,y1,x2,y2: dev float32 [n] (left / right pixel coordinates).
 * X,Y,Z: dev float64 [n]; valid: dev u8 [n] = !(Z <= 0 || x1 < x2) (:326-327).
 * The null vector of the 4x4 system is found by a one-sided Jacobi SVD in
 * fp64 (agrees with LAPACK to ~1e-10 relative on KITTI geometry).
 */
int slam_triangulate_dlt(const double *P_host, const double *Q_host,
                         const float *x1_dev, const float *y1_dev,
                         const float *x2_dev, const float *y2_dev, int n,
                         double *X_dev, double *Y_dev, double *Z_dev,
                         uint8_t *valid_dev, void *stream);

/*
 * Triangulation driven by a stereo match table (batched over frames).
 * Replaces: find_stereo_matches + get_3d_points_cloud for a range of frames
 * (utils/utils.py:302-335) without the host round trip.
 *
 * For item b of seg (the table given to the stereo slam_hamming_match call)
 * and local left keypoint i with keep[out_start+i] set and j = idx[out_start+i][0]:
 * triangulate (xq,yq)[q_start+i] with (xt,yt)[t_start+j] into row out_start+i of
 * X,Y,Z/valid.  Rows without a kept match get valid 0 and NaN coordinates.
 * stereo_map [rows] int32 (nullable): j for kept rows, -1 otherwise -- the dense
 * form of array_to_dict(inlier_indices_mapping) (utils/utils.py:156-175).
 */
int slam_triangulate_matched(const double *P_host, const double *Q_host,
                             const float *xq_dev, const float *yq_dev,
                             const float *xt_dev, const float *yt_dev,
                             const int32_t *seg_dev, int batch, int max_nq,
                             const int32_t *idx_dev, const uint8_t *keep_dev,
                             double *X_dev, double *Y_dev, double *Z_dev,
                             uint8_t *valid_dev, int32_t *stereo_map_dev,
                             void *stream);

/*
 * Four-image consensus join, batched over frame pairs, order preserving.
 * Replaces: get_matches_that_appear_in_all_4_images (utils/utils.py:338-386)
 * and find_4_images_matches (models/LoopClosure.py:71-109).
 *
 * pair [n_pairs, 8] int32 dev, one row per frame pair (earlier frame "prev", later
 * frame "cur"):
 *   [0] prevL_row0  first row of prev's left keypoints in the per-left-keypoint
 *                   tables (stereo_map, valid3d, X/Y/Z, xL, yL)
 *   [1] prevL_count number of left keypoints of prev
 *   [2] curL_row0, [3] curL_count   likewise for cur
 *   [4] t_row0      first row of this pair in the temporal match outputs
 *   [5] curR_row0   first row of cur's right keypoints in xR, yR
 *   [6] prevR_row0  first row of prev's right keypoints in xR, yR
 *   [7] unused
 * For each kept temporal match q -> t (ascending q) with stereo_map[prev q] >= 0,
 * stereo_map[cur t] >= 0 and valid3d[prev q]: emit one consensus row.
 *
 * Outputs, slot = pair * cap + m (m < count[pair] <= cap; the join stops
 * writing at cap and still counts, so count > cap signals overflow):
 *   rows [n_pairs*cap, 4] int32 (prev_left, prev_right, cur_left, cur_right)
 *   cX,cY,cZ f64 : 3-D point of prev_left in the earlier left camera frame
 *   cuL,cvL f32  : pixel of cur_left in the later left image
 *   cuR,cvR f32  : pixel of cur_right in the later right image
 *   count [n_pairs] int32
 */
int slam_join_consensus(const int32_t *pair_dev, int n_pairs, int max_nq,
                        const int32_t *t_idx_dev, const uint8_t *t_keep_dev,
                        const int32_t *stereo_map_dev,
                        const uint8_t *valid3d_dev, const double *X_dev,
                        const double *Y_dev, const double *Z_dev,
                        const float *xL_dev, const float *yL_dev,
                        const float *xR_dev, const float *yR_dev, int cap,
                        int32_t *rows_dev, double *cX_dev, double *cY_dev,
                        double *cZ_dev, float *cuL_dev, float *cvL_dev,
                        float *cuR_dev, float *cvR_dev, int32_t *count_dev,
                        void *stream);

/*
 * The same join which also fills the scoring workspace of slam_score_hypotheses for the
 * consensus sets it emits (point records and per-set bounds, for intrinsics K_host and the
 * scoring threshold thresh), so that the scoring call on these sets can be made with
 * work_ready = 1 (slam_score_hypotheses_range) and skips its own preparation pass.
 * score_work_dev: 16 * n_pairs * max(cap, 1) doubles, 16-byte aligned; NULL = plain join.
 */
int slam_join_consensus_prepared(const int32_t *pair_dev, int n_pairs, int max_nq,
                                 const int32_t *t_idx_dev, const uint8_t *t_keep_dev,
                                 const int32_t *stereo_map_dev,
                                 const uint8_t *valid3d_dev, const double *X_dev,
                                 const double *Y_dev, const double *Z_dev,
                                 const float *xL_dev, const float *yL_dev,
                                 const float *xR_dev, const float *yR_dev, int cap,
                                 int32_t *rows_dev, double *cX_dev, double *cY_dev,
                                 double *cZ_dev, float *cuL_dev, float *cvL_dev,
                                 float *cuR_dev, float *cvR_dev, int32_t *count_dev,
                                 const double *K_host, double thresh,
                                 double *score_work_dev, void *stream);

/*
 * PnP-RANSAC hypothesis scoring: supporters of H candidate poses.
 * Replaces: find_supporters (utils/utils.py:389-410) as called once per RANSAC
 * iteration and once after refinement by ransac_for_pnp (utils/utils.py:433,455).
 *
 * Rt [n_sets, H, 12] f64 dev (row-major 3x4 per hypothesis); K host 9 doubles;
 * t_right host 3 doubles (= M2[:,3]).  Point set s occupies slots
 * [s*cap, s*cap + n_s) of X,Y,Z (f64) and uL,vL,uR,vR (f32), with
 * n_s = n_dev[s] if n_dev != NULL else n_host (clamped to cap).
 * counts [n_sets, H] int32; masks (nullable) [n_sets, H, mask_words] u64 with
 * bit (i & 63) of word (i >> 6) set iff point i supports; mask_words >= ceil(cap/64).
 * A point supports iff both reprojection distances are < thresh (strict), all
 * in fp64 with one rounding per operation in the reference's order; no
 * cheirality test.
 * work_dev (nullable): caller-owned scratch of 16 * n_sets * max(cap, 1) doubles, 16-byte aligned.  When it
 * is given, masks_dev is NULL and K is a pinhole matrix [fx 0 cx; 0 fy cy; 0 0 1],
 * evaluations are first run through a division-free filter and only the ones
 * within a guard band of the threshold are evaluated in the reference's order;
 * the counts are identical either way, the call is several times faster.
 */
int slam_score_hypotheses(const double *Rt_dev, int n_sets, int H,
                          const double *K_host, const double *t_right_host,
                          const double *X_dev, const double *Y_dev,
                          const double *Z_dev, const float *uL_dev,
                          const float *vL_dev, const float *uR_dev,
                          const float *vR_dev, int cap, const int32_t *n_dev,
                          int n_host, double thresh, int32_t *counts_dev,
                          uint64_t *masks_dev, int mask_words, double *work_dev,
                          void *stream);

/*
 * slam_score_hypotheses for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference: this is slam_score_hypotheses(..., masks = NULL, work = <built in>), where the
 * workspace is a fixed array inside the library's code object (one copy per GPU, held until the
 * process ends) instead of a caller-owned buffer.  The call itself makes no hipMalloc / hipFree
 * and no allocating library call.  Counts are identical to slam_score_hypotheses.
 *
 * Limit: n_sets * max(cap, 1) <= SLAM_SCORE_STATIC_MAX_POINTS point slots (the workspace is
 * SLAM_SCORE_WORK_PER_POINT doubles per slot = 8 MiB); larger batches are refused with
 * SLAM_ERR_UNSUPPORTED and must bring their own workspace.
 * All calls made with the same GPU current share that GPU's array, whatever their stream: two
 * such calls that overlap in time race on it.  Calls on one stream, or otherwise ordered, are
 * safe -- every call rebuilds the whole workspace contents it reads.
 * The first call on a device resolves the array's address (hipGetSymbolAddress); make that
 * first call outside any graph capture.
 */
#define SLAM_SCORE_WORK_PER_POINT 16
#define SLAM_SCORE_STATIC_MAX_POINTS 65536
#define SLAM_SCORE_STATIC_WORK_DOUBLES (SLAM_SCORE_WORK_PER_POINT * SLAM_SCORE_STATIC_MAX_POINTS)
int slam_score_hypotheses_static(const double *Rt_dev, int n_sets, int H,
                                 const double *K_host, const double *t_right_host,
                                 const double *X_dev, const double *Y_dev,
                                 const double *Z_dev, const float *uL_dev,
                                 const float *vL_dev, const float *uR_dev,
                                 const float *vR_dev, int cap, const int32_t *n_dev,
                                 int n_host, double thresh, int32_t *counts_dev,
                                 void *stream);

/*
 * The same for hypotheses [h_begin, h_begin + H) of each set only (Rt, counts and masks keep
 * their row stride of H_total hypotheses per set), leaving alone every set whose entry in
 * skip_dev [n_sets] (nullable) is non-zero.  This is how the batched RANSAC scores a first
 * slice of the hypotheses, lets the stopping rule decide which sets are finished, and scores
 * the rest only for the others -- the reference never scores a hypothesis after its loop has
 * stopped either (utils/utils.py:425).  Masks need the full range and no skip list.
 * work_ready != 0: work_dev still holds what an earlier call prepared for the same points,
 * camera and threshold (the call then skips that preparation).
 */
int slam_score_hypotheses_range(const double *Rt_dev, int n_sets, int H_total,
                                int h_begin, int H, const int32_t *skip_dev,
                                const double *K_host, const double *t_right_host,
                                const double *X_dev, const double *Y_dev,
                                const double *Z_dev, const float *uL_dev,
                                const float *vL_dev, const float *uR_dev,
                                const float *vR_dev, int cap, const int32_t *n_dev,
                                int n_host, double thresh, int32_t *counts_dev,
                                uint64_t *masks_dev, int mask_words,
                                double *work_dev, int work_ready, void *stream);

/*
 * Winner of a scored hypothesis set: the FIRST hypothesis with the largest
 * supporter count, as ransac_for_pnp's strict '>' update keeps it
 * (utils/utils.py:438-442) when every hypothesis is consumed.
 * counts [n_sets, H] int32 dev; Rt [n_sets, H, 12] f64 dev (nullable with best_Rt).
 * best, best_count [n_sets] int32; best_Rt [n_sets, 12] f64 (nullable).
 */
int slam_select_best(const int32_t *counts_dev, const double *Rt_dev, int n_sets,
                     int H, int32_t *best_dev, int32_t *best_count_dev,
                     double *best_Rt_dev, void *stream);

/*
 * Inlier tables of a batch of frame pairs: the consensus rows whose supporter
 * bit is set, in order, each with the feature locations the track store keeps.
 * Replaces: the Rt_inliers list comprehension (utils/utils.py:277-278) and the
 * location lookups of TrackDatabase.update_tracks_from_frame
 * (models/TrackDatabase.py:38-46).
 *
 * pair as for slam_join_consensus; rows/count/cap are its outputs; masks
 * [n_pairs, mask_words] u64 is the supporter mask of ONE pose per pair
 * (slam_score_hypotheses with H = 1).  out_off [n_pairs] int32 (nullable):
 * first output slot of each pair (e.g. an exclusive prefix sum of the supporter
 * counts, for dense tables); NULL means pair * cap.
 * inl_rows [slots,4] int32 (prev_left, prev_right, cur_left, cur_right);
 * inl_loc [slots,6] f32 (xl, xr, yl of prev; xl, xr, yl of cur); n_inl [n_pairs].
 */
int slam_build_inlier_tables(const int32_t *pair_dev, int n_pairs,
                             const int32_t *rows_dev, const int32_t *count_dev,
                             int cap, const uint64_t *masks_dev, int mask_words,
                             const float *xL_dev, const float *yL_dev,
                             const float *xR_dev, const int32_t *out_off_dev,
                             int32_t *inl_rows_dev, float *inl_loc_dev,
                             int32_t *n_inl_dev, void *stream);

/*
 * The same tables straight from one pose per pair, packed back to back: the final
 * find_supporters of the winning pose (utils/utils.py:455), the Rt_inliers list
 * (utils/utils.py:277-278) and the location lookups (models/TrackDatabase.py:38-46) in one
 * launch -- no mask, no separate scoring pass, no prefix sum.
 *
 * pose [n_pairs,12] f64: the pose whose supporters are tabulated; best_count [n_pairs]: its
 * supporter count as the scoring pass reported it (slam_select_best), from which pair p's first
 * output slot is the sum over the pairs before it.  K, t_right, thresh and the consensus
 * point tables (X .. vR, [n_pairs*cap]) as for slam_score_hypotheses; the verdict per match is
 * that of its fp64 reference evaluation, so n_inl[p] == best_count[p].
 */
int slam_winner_tables(const int32_t *pair_dev, int n_pairs, const int32_t *rows_dev,
                       const int32_t *count_dev, int cap, const double *K,
                       const double *t_right, double thresh, const double *pose_dev,
                       const int32_t *best_count_dev, const double *X_dev,
                       const double *Y_dev, const double *Z_dev, const float *uL_dev,
                       const float *vL_dev, const float *uR_dev, const float *vR_dev,
                       const float *xL_dev, const float *yL_dev, const float *xR_dev,
                       int32_t *inl_rows_dev, float *inl_loc_dev, int32_t *n_inl_dev,
                       void *stream);

/*
 * Track ids and the track-point table of a whole run of consecutive frame pairs, on the device.
 * Replaces: the row-by-row bookkeeping of TrackDatabase.update_tracks_from_frame / add_to_track /
 * get_kp_trackId / generate_new_track_id / prepare_to_next_pair (models/TrackDatabase.py:34-76,
 * 166-184) over the inlier tables of slam_winner_tables / slam_build_inlier_tables.
 *
 * inl_rows [cap_rows,4] int32 / inl_loc [cap_rows,6] f32: the dense inlier tables (pair p's rows
 * follow pair p-1's), n_inl [n_pairs]; pair p is frames (first_frame + p, first_frame + p + 1).
 * kp_cap: every keypoint index in columns 0 and 2 is < kp_cap.
 * A row of pair p continues the track of the LAST row of pair p-1 whose cur_left equals its
 * prev_left, else it opens track number (tracks opened before it in (pair, row) order).
 * Outputs:
 *   row_track, row_depth [cap_rows]   track id of every row, and its position in the track minus 1
 *   track_start [cap_rows + 1]        first point slot of track t; track_start[n_tracks] = n_points
 *   pt_kp, pt_frame [2 cap_rows], pt_loc [2 cap_rows, 3]
 *                                     per track point, tracks back to back in id order, points in
 *                                     frame order: left keypoint index, frame id, (xl, xr, yl) --
 *                                     the tuples of models/TrackDatabase.py:57-58
 *   summary [8] int32: [0] rows, [1] tracks, [2] points, [3] longest track in points,
 *                      [4] flags: 1 = two rows of one pair continue the same row (then the point
 *                      table is not what the reference builds: replay on the host instead),
 *                      2 = a keypoint index >= kp_cap, 4 = sum(n_inl) > cap_rows (nothing done)
 * work: slam_track_assign_work_bytes(n_pairs, cap_rows, kp_cap) bytes, 256-byte aligned.
 */
size_t slam_track_assign_work_bytes(int n_pairs, int cap_rows, int kp_cap);
int slam_track_assign(const int32_t *inl_rows_dev, const float *inl_loc_dev,
                      const int32_t *n_inl_dev, int n_pairs, int cap_rows, int kp_cap,
                      int first_frame, int32_t *row_track_dev, int32_t *row_depth_dev,
                      int32_t *track_start_dev, int32_t *pt_kp_dev, int32_t *pt_frame_dev,
                      float *pt_loc_dev, int32_t *summary_dev, void *work_dev,
                      size_t work_bytes, void *stream);

/* ---- batched PnP-RANSAC around slam_score_hypotheses --------------------------------
 * ransac_for_pnp (utils/utils.py:413-464) for many point sets at once.  The reference
 * draws with Python's global `random`, solves each 4-sample with cv2 (AP3P) and walks the
 * iterations one by one; here all hypotheses of all sets are drawn, solved and scored in
 * one launch each and the stopping rule is applied to the counts afterwards.  Poses from
 * these solvers agree with cv2's to solver tolerance only; the supporter counts of any
 * given pose are exact.
 */

/*
 * Minimal samples: replaces `random.sample(points_to_choose_from, k=4)`
 * (utils/utils.py:427).  For set s and hypothesis h the four distinct positions in
 * [0, n_s) come from a splitmix64 stream keyed by (seed, s, h): draw k is a uniform rank
 * among the n_s - k positions not drawn yet.  samples [n_sets, H, 4] int32 dev,
 * 16-byte aligned; sets with fewer than 4 points get -1s.
 */
int slam_ransac_sample(uint64_t seed, int n_sets, int H, const int32_t *n_dev,
                       int n_host, int32_t *samples_dev, void *stream);

/*
 * Pose hypotheses from minimal samples: replaces
 * solvePnP(kp_left, candidate_4_points, K, flags=cv2.SOLVEPNP_AP3P)
 * (utils/utils.py:428,481-497).  Three-point resection on the first three
 * correspondences of each sample (3-D point of the previous frame, left pixel of the
 * current frame), the fourth correspondence picks among the solutions.
 * samples [n_sets, H_total, 4], Rt [n_sets, H_total, 12] f64 dev, 16-byte aligned; only
 * hypotheses [h_begin, h_begin + H) of the sets whose skip_dev entry (nullable array) is zero
 * are solved.  A failed solve (the reference's None) is written as NaNs, which
 * slam_score_hypotheses counts as 0 supporters.
 * K must be a pinhole matrix [fx 0 cx; 0 fy cy; 0 0 1].
 */
int slam_p3p_hypotheses(const double *K_host, const double *X_dev,
                        const double *Y_dev, const double *Z_dev,
                        const float *uL_dev, const float *vL_dev, int cap,
                        const int32_t *n_dev, int n_host,
                        const int32_t *samples_dev, int n_sets, int H_total,
                        int h_begin, int H, const int32_t *skip_dev,
                        double *Rt_dev, void *stream);

/*
 * The sampling loop's control flow (utils/utils.py:422-446, 475-478) replayed over the
 * supporter counts of hypotheses 0..min(H, max_iterations, n_avail)-1 taken in order
 * (n_avail = how many hypotheses per set have been scored so far; H = row stride):
 * strict '>' best update, epsilon = min(epsilon, 1 - count/n), I = min(ceil(log(1-0.999)
 * / log(1-(1-epsilon)^4)), max_iterations), stop when i >= I or epsilon == 0.
 * best [n_sets] = index of the winner (-1: no hypothesis had a supporter),
 * best_count, best_Rt [n_sets, 12] (NaN without winner), iters [n_sets] = iterations the
 * loop would have run (nullable).  Sets with n < 4 get -1 / 0 iterations.
 * stopped [n_sets] (nullable) = 1 where the loop is over -- the rule fired within the
 * hypotheses available, or min(H, max_iterations) of them were consumed -- and 0 where more
 * hypotheses have to be scored before the outcome is known.
 */
int slam_ransac_select(const int32_t *counts_dev, const double *Rt_dev, int n_sets,
                       int H, int n_avail, const int32_t *n_dev, int n_host, int cap,
                       int max_iterations, int32_t *best_dev,
                       int32_t *best_count_dev, double *best_Rt_dev,
                       int32_t *iters_dev, int32_t *stopped_dev, void *stream);

/*
 * Refinement of each winner on its supporters: replaces
 * solvePnP(kp_left, supporters, K, flags=cv2.SOLVEPNP_ITERATIVE) (utils/utils.py:450-454).
 * Levenberg-Marquardt on the left-image reprojection error over the points whose bit is
 * set in masks [n_sets, mask_words], started from Rt0 [n_sets, 12]; at most `iterations`
 * steps.  Rt_out [n_sets, 12]; ok [n_sets] (nullable) is 0 where fewer than 3 supporters
 * or a non-finite system left Rt_out = Rt0 (the reference's `refined_Rt is None` branch).
 */
int slam_refine_poses(const double *K_host, const double *X_dev, const double *Y_dev,
                      const double *Z_dev, const float *uL_dev, const float *vL_dev,
                      int cap, const int32_t *n_dev, int n_host,
                      const uint64_t *masks_dev, int mask_words,
                      const double *Rt0_dev, int n_sets, int iterations,
                      double *Rt_out_dev, int32_t *ok_dev, void *stream);

/*
 * The tail of ransac_for_pnp in one launch (utils/utils.py:449-455): the supporters of the
 * winner Rt0 are found here with the supporter test of slam_score_hypotheses (no mask is
 * handed over), the pose is refined on them as slam_refine_poses does, and
 * ref_count [n_sets] receives the number of supporters of Rt_out -- the count the final
 * find_supporters call reports.  uR, vR, t_right, thresh as in slam_score_hypotheses.
 */
int slam_refine_winners(const double *K_host, const double *t_right_host, double thresh,
                        const double *X_dev, const double *Y_dev, const double *Z_dev,
                        const float *uL_dev, const float *vL_dev,
                        const float *uR_dev, const float *vR_dev,
                        int cap, const int32_t *n_dev, int n_host,
                        const double *Rt0_dev, int n_sets, int iterations,
                        double *Rt_out_dev, int32_t *ok_dev, int32_t *ref_count_dev,
                        void *stream);

/*
 * slam_refine_winners with the right image's residuals of the same points added to the
 * cost (the right camera sits at t_right from the left one).  No counterpart in the
 * reference, whose refinement sees the left keypoints only (utils/utils.py:452): an
 * option for callers that chain thousands of poses, where the left-only cost on
 * triangulated points leaves a heading bias of constant sign.  Same arguments, same
 * supporter test, same ok / ref_count meaning.
 */
int slam_refine_winners_stereo(const double *K_host, const double *t_right_host, double thresh,
                               const double *X_dev, const double *Y_dev, const double *Z_dev,
                               const float *uL_dev, const float *vL_dev,
                               const float *uR_dev, const float *vR_dev,
                               int cap, const int32_t *n_dev, int n_host,
                               const double *Rt0_dev, int n_sets, int iterations,
                               double *Rt_out_dev, int32_t *ok_dev, int32_t *ref_count_dev,
                               void *stream);

/*
 * Final choice (utils/utils.py:455-463): out_Rt = refined pose when its supporter count
 * is >= the winner's, else the winner; ratio = refined count / n (always the refined
 * pose's, as in the reference); found = 0 (and NaNs) where the set had fewer than 4
 * points or no hypothesis had a supporter (the reference returns None).
 */
int slam_ransac_adopt(int n_sets, const int32_t *n_dev, int n_host, int cap,
                      const int32_t *best_count_dev, const int32_t *ref_count_dev,
                      const double *best_Rt_dev, const double *ref_Rt_dev,
                      double *out_Rt_dev, double *ratio_dev, int32_t *found_dev,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_AMD_H */
