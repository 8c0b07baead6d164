"""Pure numpy/Python restatement of the reference front-end, statement by statement.

TEST INFRASTRUCTURE ONLY (see oracle/cpu.py).  Slow Python loops: use on small
cases.  Each function names the reference lines it follows.  Where the reference
calls into OpenCV (knnMatch) the published behaviour is restated; where it calls
numpy (np.linalg.svd, '@', np.linalg.norm) the same numpy call is made, so for
triangulation this twin *is* the reference arithmetic (LAPACK gesdd) and pins the
C/HIP Jacobi solver to the 1e-4 relative contract.

Parity status: everything here except knn_match is pinned -- held by tests/test_reference_golden.py
against tests/golden/reference_golden.npz, the outputs of the reference's own function bodies
executed in the build container (tests/golden/make_reference_golden.py).  knn_match restates
cv2.BFMatcher.knnMatch from its published behaviour (cv2 is not installed): "parity unpinned".
"""
import math

import numpy as np


class DMatchLike:
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, q, t, d):
        self.queryIdx, self.trainIdx, self.imgIdx, self.distance = int(q), int(t), 0, float(np.float32(d))


_POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.int32)


def knn_match(q, t, k):
    """cv2.BFMatcher(NORM_HAMMING).knnMatch(q, t, k) (models/Matcher.py:127,140,199,219).

    Train rows are scanned in index order; a candidate enters the per-query sorted
    list only if strictly better than the current k-th, and is placed before
    entries that are strictly worse.  Lists have min(k, len(t)) entries.
    """
    out = []
    for i in range(len(q)):
        best = []  # sorted list of (dist, idx)
        for j in range(len(t)):
            d = int(_POP8[np.bitwise_xor(q[i], t[j])].sum())
            if len(best) < k:
                pos = len(best)
                while pos > 0 and best[pos - 1][0] > d:
                    pos -= 1
                best.insert(pos, (d, j))
            elif d < best[-1][0]:
                pos = k - 1
                while pos > 0 and best[pos - 1][0] > d:
                    pos -= 1
                best.insert(pos, (d, j))
                best.pop()
        out.append([DMatchLike(i, j, d) for d, j in best])
    return out


def apply_thresholds(matches, threshold):
    """Matcher.apply_thresholds (models/Matcher.py:178-184)."""
    filtered = []
    for i, (m, n) in enumerate(matches):
        if m.distance < threshold * n.distance:
            filtered.append([m])
    return filtered


def coords_from_kps(matches, pts1, pts2):
    """coords_from_kps (utils/utils.py:42-72); pts are [N,2] float32 keypoint coordinates."""
    query_idxs = np.array([m[0].queryIdx for m in matches])
    train_idxs = np.array([m[0].trainIdx for m in matches])
    indices_mapping = np.stack((query_idxs, train_idxs))
    x1, y1 = np.array([tuple(float(c) for c in pts1[i]) for i in query_idxs]).T
    x2, y2 = np.array([tuple(float(c) for c in pts2[i]) for i in train_idxs]).T
    return x1, y1, x2, y2, indices_mapping


def rectified_stereo_pattern(y1, y2, indices_mapping, thresh=1):
    """rectificatied_stereo_pattern (utils/utils.py:89-116)."""
    deviations = np.abs(y1 - y2)
    inliers_idx = np.argwhere(deviations <= thresh)[:, 0]
    return indices_mapping[:, inliers_idx]


def least_squares(p1, p2, Pmat, Qmat):
    """least_squares (utils/utils.py:527-560), same numpy calls."""
    x1, y1 = p1
    x2, y2 = p2
    P1, P2, P3 = Pmat[0, :], Pmat[1, :], Pmat[2, :]
    Q1, Q2, Q3 = Qmat[0, :], Qmat[1, :], Qmat[2, :]
    A = np.array([x1 * P3 - P1, y1 * P3 - P2, x2 * Q3 - Q1, y2 * Q3 - Q2])
    u, s, vh = np.linalg.svd(A)
    solution_4d = vh[-1]
    return solution_4d[:3] / solution_4d[-1]


def get_3d_points_cloud(inlier_indices_mapping, k, m1, m2, pts1, pts2):
    """get_3d_points_cloud (utils/utils.py:318-335)."""
    points, ind_to_3d = [], {}
    for ind_1, ind_2 in inlier_indices_mapping.T:
        x1, y1 = (float(c) for c in pts1[ind_1])
        x2, y2 = (float(c) for c in pts2[ind_2])
        sol = least_squares((x1, y1), (x2, y2), k @ m1, k @ m2)
        if sol[-1] <= 0 or x1 < x2:
            continue
        points.append(sol)
        ind_to_3d[int(ind_1)] = sol
    return points, ind_to_3d


def array_to_dict(arr):
    """array_to_dict (utils/utils.py:156-175)."""
    return {int(arr[0, i]): int(arr[1, i]) for i in range(arr.shape[1])}


def matches_in_all_4_images(consecutive_matches, prev_map, cur_map, ind_to_3d_prev):
    """get_matches_that_appear_in_all_4_images (utils/utils.py:338-386)."""
    consensus, filtered = [], []
    for matches_list in consecutive_matches:
        m = matches_list[0]
        p, c = m.queryIdx, m.trainIdx
        if p in prev_map and c in cur_map and p in ind_to_3d_prev:
            consensus.append((p, prev_map[p], c, cur_map[c], ind_to_3d_prev[p]))
            filtered.append(matches_list)
    return consensus, filtered


def find_supporters(Rt, m2, consensus_matches, k, pts_left, pts_right, thresh=2):
    """find_supporters (utils/utils.py:389-410), same numpy calls."""
    good = np.zeros(len(consensus_matches))
    num = 0
    for i, m in enumerate(consensus_matches):
        _, _, cur_left_ind, cur_right_ind, point_3d_prev = m
        hom = np.r_[point_3d_prev, 1]
        left = Rt @ hom
        right = left + m2[:, -1]
        lh = k @ left
        lp = lh[:2] / lh[2]
        rh = k @ right
        rp = rh[:2] / rh[2]
        l_real = np.array([float(c) for c in pts_left[cur_left_ind]])
        r_real = np.array([float(c) for c in pts_right[cur_right_ind]])
        ld = np.linalg.norm(l_real - lp)
        rd = np.linalg.norm(r_real - rp)
        s = (ld < thresh) and (rd < thresh)
        good[i] = s
        num += s
    return good, num


def ransac_num_of_iterations(epsilon=0.99, p=0.99, s=4):
    """ransac_num_of_iterations (utils/utils.py:475-478)."""
    if epsilon == 0:
        return 0
    return int(np.ceil(np.log(1 - p) / np.log(1 - (1 - epsilon) ** s)))


def ransac_stop_replay(counts, n_points, max_iterations, solved=None):
    """The control flow of ransac_for_pnp's sampling loop (utils/utils.py:420-446)
    replayed over precomputed per-iteration supporter counts.

    counts[i] is the supporter count of the i-th drawn hypothesis; solved[i] False
    means the minimal solver returned None for it (i advances, nothing else moves).
    Returns (best_iteration or -1, iterations_consumed).
    """
    best, best_it = 0, -1
    eps, I, i = 0.99, max_iterations, 0
    while i < I and eps != 0:
        if solved is not None and not solved[i]:
            i += 1
            continue
        n = int(counts[i])
        if n > best:
            best, best_it = n, i
        eps = min(eps, 1 - (n / n_points))
        I = min(ransac_num_of_iterations(eps, 0.999), max_iterations)
        i += 1
    return best_it, i
