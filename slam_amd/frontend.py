"""The reference's front-end functions (utils/utils.py) with the same names, arguments
and return values, computing on the GPU through the slam_amd library.

Everything that is arithmetic -- Hamming matching, the row filter, DLT triangulation,
supporter counting -- runs in HIP kernels; what stays on the host is the reference's own
glue: Python lists of DMatch, dict lookups of the four-image join, the RANSAC control
flow and its use of the process-global `random` stream.

`track_camera_for_many_images` is the per-frame loop of the reference, kept for callers
that need its exact call pattern; `slam_amd.pipeline` / `slam_amd.vo` hold the batched
fast path.
"""
import math
import random

import gc

import numpy as np
import torch

from . import ops, pnp
from .constants import (LEFT, PIXEL_THRESH, RIGHT, SEED, STEREO_ROW_THRESH, VERTICAL_REPRESENTATION,
                        VO_MAX_ITERATIONS)
from .matcher import Matcher, _default_device
from .track_database import TrackDatabase
from .types import MatchLists, keypoints_to_array

random.seed(SEED)   # utils/utils.py:13 seeds the global stream at import

_CAMERAS = None
_NUM_FRAMES = 2560          # the frame count the reference hard-codes (utils/utils.py:241)


def set_sequence_length(num_of_frames=2560):
    """How many frames track_camera_for_many_images() walks when its caller passes none (the
    reference's callers pass nothing: models/TrajectorySolver.py:140)."""
    global _NUM_FRAMES
    _NUM_FRAMES = int(num_of_frames)


def set_cameras(k, m1, m2):
    """Install the calibration `read_cameras()` returns (the reference parses
    VAN_ex/dataset/sequences/05/calib.txt, utils/utils.py:75-86)."""
    global _CAMERAS
    _CAMERAS = (np.asarray(k, dtype=np.float64), np.asarray(m1, dtype=np.float64), np.asarray(m2, dtype=np.float64))


def read_cameras(path=None):
    """K, M1 = K^-1 P0, M2 = K^-1 P1 (utils/utils.py:75-86).  Reads `path` if given, else
    the calibration installed with set_cameras, else the KITTI odometry constants the
    reference's dataset ships."""
    if path is not None:
        with open(path) as f:
            l1 = [float(i) for i in f.readline().split()[1:]]
            l2 = [float(i) for i in f.readline().split()[1:]]
        m1 = np.array(l1).reshape(3, 4)
        m2 = np.array(l2).reshape(3, 4)
        k = m1[:, :3]
        return k, np.linalg.inv(k) @ m1, np.linalg.inv(k) @ m2
    if _CAMERAS is not None:
        return _CAMERAS
    from .synthetic import read_cameras as kitti
    return kitti()


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(_default_device())


# --------------------------------------------------------------------------- matching glue
def coords_from_kps(matches, kp1, kp2):
    """(x1, y1, x2, y2, indices_mapping[2,M]) of a k=1 match list (utils/utils.py:42-72)."""
    if isinstance(matches, MatchLists) and len(matches) and (matches.train[:, 0] >= 0).all():
        query_idxs, train_idxs = matches.query.astype(np.int64), matches.train[:, 0].astype(np.int64)
    else:
        query_idxs = np.array([match[0].queryIdx for match in matches])
        train_idxs = np.array([match[0].trainIdx for match in matches])
    indices_mapping = np.stack((query_idxs, train_idxs))
    x1, y1 = np.array([kp1[idx].pt for idx in query_idxs]).T
    x2, y2 = np.array([kp2[idx].pt for idx in train_idxs]).T
    return x1, y1, x2, y2, indices_mapping


def _rectified_stereo_classifier(y1, y2, thresh=1):
    """inliers / outliers of |y1 - y2| <= thresh, evaluated on the GPU in fp64 on the
    float32-origin coordinates (utils/utils.py:89-100)."""
    y1 = np.asarray(y1)
    y2 = np.asarray(y2)
    if y1.size == 0:
        e = np.zeros(0, dtype=np.int64)
        return e, e
    f1, f2 = y1.astype(np.float32), y2.astype(np.float32)
    if not (np.array_equal(f1.astype(np.float64), y1) and np.array_equal(f2.astype(np.float64), y2)):
        raise ValueError("row filter: coordinates must be float32-representable (cv2.KeyPoint.pt values)")
    keep = ops.row_filter(_t(f1, np.float32), _t(f2, np.float32), float(thresh)).cpu().numpy().astype(bool)
    return np.flatnonzero(keep), np.flatnonzero(~keep)


def rectificatied_stereo_pattern(y1, y2, indices_mapping, thresh=1):
    """(img1in, img2in, img1out, img2out, inlier_indices_mapping) (utils/utils.py:103-116)."""
    inliers_idx, outliers_idx = _rectified_stereo_classifier(y1, y2, thresh)
    inlier_indices_mapping = indices_mapping[:, inliers_idx]
    img1in, img2in = inlier_indices_mapping[0], inlier_indices_mapping[1]
    img1out, img2out = indices_mapping[0, outliers_idx], indices_mapping[1, outliers_idx]
    return img1in, img2in, img1out, img2out, inlier_indices_mapping


def array_to_dict(arr):
    """{arr[0,i]: arr[1,i]}; later duplicates of a key overwrite earlier ones
    (utils/utils.py:156-175)."""
    arr = np.asarray(arr)
    return dict(zip(arr[0].tolist(), arr[1].tolist()))       # (Python ints: they hash and compare like the numpy scalars)


def find_stereo_matches(matcher, file_index):
    """k=1 left->right matching + 2-pixel row filter of one frame; returns the inlier
    index mapping [2,M] and leaves the filtered match list in the matcher
    (utils/utils.py:302-315)."""
    matcher.detect_and_compute(file_index)
    fused = isinstance(matcher, Matcher) and hasattr(matcher.matcher, "knn_arrays")
    if fused:
        matcher.find_matching_features(with_significance_test=False, row_thresh=STEREO_ROW_THRESH)
    else:
        matcher.find_matching_features(with_significance_test=False)
    filtered_matches = matcher.get_filtered_matches(file_index)
    if fused and file_index in matcher._stereo_keep:
        # the row filter was evaluated inside the matching kernel
        _, train, keep = matcher._stereo_keep[file_index]
        q = np.flatnonzero(keep & (train >= 0))
        inlier_indices_mapping = np.stack((q, train[q].astype(np.int64)))
        img1in = inlier_indices_mapping[0]
    else:
        kp1, kp2 = matcher.get_kp(file_index)
        x1, y1, x2, y2, indices_mapping = coords_from_kps(filtered_matches, kp1, kp2)
        img1in, _, _, _, inlier_indices_mapping = rectificatied_stereo_pattern(y1, y2, indices_mapping,
                                                                               thresh=STEREO_ROW_THRESH)
    matcher.filter_matches(img1in, file_index)
    return inlier_indices_mapping


def match_next_pair(cur_file, matcher):
    return find_stereo_matches(matcher, cur_file)


# --------------------------------------------------------------------------- triangulation
def least_squares(p1, p2, Pmat, Qmat):
    """DLT triangulation of one correspondence (utils/utils.py:527-560)."""
    x1, y1 = p1
    x2, y2 = p2
    f = lambda v: _t([v], np.float32)
    X, Y, Z, _ = ops.triangulate_dlt(Pmat, Qmat, f(x1), f(y1), f(x2), f(y2))
    return np.array([X.item(), Y.item(), Z.item()])


def triangulate_points(k, m1, m2, xy1, xy2):
    """Batched form: xy1, xy2 [M,2] float32 -> (points [M,3] f64, valid [M] bool)."""
    xy1 = np.ascontiguousarray(xy1, dtype=np.float32).reshape(-1, 2)
    xy2 = np.ascontiguousarray(xy2, dtype=np.float32).reshape(-1, 2)
    if xy1.shape[0] == 0:
        return np.zeros((0, 3)), np.zeros(0, dtype=bool)
    X, Y, Z, valid = ops.triangulate_dlt(k @ m1, k @ m2, _t(xy1[:, 0], np.float32), _t(xy1[:, 1], np.float32),
                                         _t(xy2[:, 0], np.float32), _t(xy2[:, 1], np.float32))
    pts = torch.stack([X, Y, Z], dim=1).cpu().numpy()
    return pts, valid.cpu().numpy().astype(bool)


def get_3d_points_cloud(inlier_indices_mapping, k, m1, m2, matcher, file_index=0, debug=False):
    """Triangulate every stereo inlier of a frame; points with Z <= 0 or xL < xR are
    dropped.  Returns (list of points, {left keypoint index: point}) (utils/utils.py:318-335)."""
    kp1, kp2 = matcher.get_kp(file_index)
    ind1 = np.asarray(inlier_indices_mapping[0], dtype=np.int64)
    ind2 = np.asarray(inlier_indices_mapping[1], dtype=np.int64)
    xy1 = keypoints_to_array(kp1)[ind1] if len(ind1) else np.zeros((0, 2), np.float32)
    xy2 = keypoints_to_array(kp2)[ind2] if len(ind2) else np.zeros((0, 2), np.float32)
    pts, valid = triangulate_points(np.asarray(k, dtype=np.float64), np.asarray(m1, dtype=np.float64),
                                    np.asarray(m2, dtype=np.float64), xy1, xy2)
    kept = np.flatnonzero(valid)
    inlier_points_in_3d = list(pts[kept])                    # one [3] array per point, as the reference collects them
    ind_to_3d_point_dict = dict(zip(ind1[kept].tolist(), inlier_points_in_3d))
    return inlier_points_in_3d, ind_to_3d_point_dict


# --------------------------------------------------------------------------- consensus
def get_matches_that_appear_in_all_4_images(consecutive_matches, prev_indices_mapping, cur_indices_mapping,
                                            ind_to_3d_point_prev):
    """Temporal matches whose two ends both have a stereo partner and whose earlier end
    has a 3-D point, as (prev_left, prev_right, cur_left, cur_right, X) tuples in match
    order (utils/utils.py:338-386).  Dictionary glue, host side like the reference."""
    concensus_matces = []
    if isinstance(consecutive_matches, MatchLists) and (len(consecutive_matches) == 0 or
                                                        (consecutive_matches.train[:, 0] >= 0).all()):
        # the same walk over plain ints; the surviving match lists are made only if someone reads them
        taken = []
        for pos, (prev_left_kp, cur_left_kp) in enumerate(zip(consecutive_matches.query.tolist(),
                                                              consecutive_matches.train[:, 0].tolist())):
            if (prev_left_kp in prev_indices_mapping and cur_left_kp in cur_indices_mapping
                    and prev_left_kp in ind_to_3d_point_prev):
                concensus_matces.append((prev_left_kp, prev_indices_mapping[prev_left_kp], cur_left_kp,
                                         cur_indices_mapping[cur_left_kp], ind_to_3d_point_prev[prev_left_kp]))
                taken.append(pos)
        return concensus_matces, consecutive_matches.subset(taken)
    filtered_matches = []
    for matches_list in consecutive_matches:
        m = matches_list[0]
        prev_left_kp, cur_left_kp = m.queryIdx, m.trainIdx
        if (prev_left_kp in prev_indices_mapping and cur_left_kp in cur_indices_mapping
                and prev_left_kp in ind_to_3d_point_prev):
            concensus_matces.append((prev_left_kp, prev_indices_mapping[prev_left_kp], cur_left_kp,
                                     cur_indices_mapping[cur_left_kp], ind_to_3d_point_prev[prev_left_kp]))
            filtered_matches.append(matches_list)
    return concensus_matces, filtered_matches


def _consensus_arrays(consensus_matches, kp_left, kp_right):
    n = len(consensus_matches)
    X = np.empty((n, 3))
    cl = np.empty(n, dtype=np.int64)
    cr = np.empty(n, dtype=np.int64)
    for i, m in enumerate(consensus_matches):
        cl[i], cr[i], X[i] = m[2], m[3], m[4]
    uvL = keypoints_to_array(kp_left)[cl] if n else np.zeros((0, 2), np.float32)
    uvR = keypoints_to_array(kp_right)[cr] if n else np.zeros((0, 2), np.float32)
    return X, uvL, uvR


class _ScoreSet:
    """Consensus points of one frame pair on the device, scored against many poses."""

    def __init__(self, X, uvL, uvR, k, m2, thresh):
        """X [n,3] f64 points in the earlier left camera frame; uvL, uvR [n,2] float32
        pixels in the later left / right images."""
        self.n = X.shape[0]
        self.k = np.asarray(k, dtype=np.float64)
        self.tr = np.asarray(m2, dtype=np.float64)[:, -1]
        self.thresh = float(thresh)
        cap = max(self.n, 1)
        pad64 = lambda a: _t(np.concatenate([a, np.zeros(cap - self.n)]), np.float64)
        pad32 = lambda a: _t(np.concatenate([a.astype(np.float32), np.zeros(cap - self.n, np.float32)]), np.float32)
        self.cap = cap
        self._n_dev = None
        self.pts = (pad64(X[:, 0]), pad64(X[:, 1]), pad64(X[:, 2]), pad32(uvL[:, 0]), pad32(uvL[:, 1]),
                    pad32(uvR[:, 0]), pad32(uvR[:, 1]))

    @classmethod
    def from_matches(cls, consensus_matches, k, m2, kp_left, kp_right, thresh):
        X, uvL, uvR = _consensus_arrays(consensus_matches, kp_left, kp_right)
        return cls(X, uvL, uvR, k, m2, thresh)

    def counts(self, Rts):
        """Supporter counts of poses [H,3,4] (numpy or a device tensor [H,12]) -> numpy [H]."""
        if not torch.is_tensor(Rts):
            Rts = _t(np.ascontiguousarray(Rts, dtype=np.float64).reshape(-1, 12), np.float64)
        c = ops.score_hypotheses(Rts.reshape(1, -1, 3, 4), self.k, self.tr, *self.pts, thresh=self.thresh, cap=self.cap,
                                 n_host=self.n)
        return c.cpu().numpy()[0]

    def solve(self, samples):
        """Minimal solves on the device (slam_p3p_hypotheses: P3P on the first three correspondences of
        each 4-sample, the fourth disambiguates): samples [H,4] positions -> device tensor [H,12];
        a failed solve is a NaN pose (the reference's None, utils/utils.py:429-432)."""
        if self._n_dev is None:
            self._n_dev = torch.tensor([self.n], dtype=torch.int32, device=self.pts[0].device)
        smp = torch.from_numpy(np.ascontiguousarray(samples, dtype=np.int32).reshape(1, -1, 4)).to(self._n_dev.device)
        X, Y, Z, uL, vL = self.pts[:5]
        return ops.p3p_hypotheses(self.k, X, Y, Z, uL, vL, self.cap, self._n_dev, smp)[0]

    def solve_and_count(self, samples):
        """(poses numpy [H,3,4] with NaN rows for failed solves, supporter counts numpy [H]) of the
        minimal samples [H,4]: one solve launch, one scoring launch, two small copies back."""
        Rt = self.solve(samples)
        c = ops.score_hypotheses(Rt.reshape(1, -1, 3, 4), self.k, self.tr, *self.pts, thresh=self.thresh, cap=self.cap,
                                 n_host=self.n)
        return Rt.cpu().numpy().reshape(-1, 3, 4), c.cpu().numpy()[0]

    def refine(self, Rt):
        """The tail of ransac_for_pnp on the device (slam_refine_winners): supporters of Rt, Levenberg-
        Marquardt on their left-image residuals (utils/utils.py:450-454), -> refined pose [3,4] (Rt
        itself where the reference's `refined_Rt is None` branch applies)."""
        if self._n_dev is None:
            self._n_dev = torch.tensor([self.n], dtype=torch.int32, device=self.pts[0].device)
        Rt0 = _t(np.ascontiguousarray(Rt, dtype=np.float64).reshape(1, 12), np.float64)
        out, ok, _ = ops.refine_winners(self.k, self.tr, *self.pts, self.cap, self._n_dev, Rt0, thresh=self.thresh)
        return out.cpu().numpy().reshape(3, 4)

    def mask(self, Rt):
        """(float64 0/1 array [n], count) for one pose -- find_supporters' return value."""
        Rt = np.ascontiguousarray(Rt, dtype=np.float64).reshape(1, 3, 4)
        c, m = ops.score_hypotheses(_t(Rt, np.float64), self.k, self.tr, *self.pts, thresh=self.thresh, cap=self.cap,
                                    n_host=self.n, want_masks=True)
        words = m.cpu().numpy()[0, 0].view(np.uint64)
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:self.n]
        return bits.astype(np.float64), int(c.item())


def find_supporters(Rt, m2, consensus_matches, k, kp_left, kp_right, thresh=2, debug=True, file_index=0):
    """Which consensus matches reproject within `thresh` pixels in both current images
    under pose Rt (utils/utils.py:389-410).  Returns (float64 0/1 array, count)."""
    if len(consensus_matches) == 0:
        return np.zeros(0), 0
    good, num = _ScoreSet.from_matches(consensus_matches, k, m2, kp_left, kp_right, thresh).mask(Rt)
    if debug:
        print(f"out of {len(consensus_matches)} matches, {num} are supporters for this Rt choice")
    return good, num


def ransac_num_of_iterations(epsilon=0.99, p=0.99, s=4):
    """utils/utils.py:475-478."""
    if epsilon == 0:
        return 0
    return int(np.ceil(np.log(1 - p) / np.log(1 - (1 - epsilon) ** s)))


def solvePnP(kp, corresponding_points, camera_intrinsic, flags=0):
    """[3,4] camera pose from consensus tuples, or None (utils/utils.py:481-492).
    flags = pnp.SOLVEPNP_AP3P: minimal solve on four points; otherwise iterative
    refinement over all points (started from a minimal solve on a spread subset)."""
    points_3d = np.array([m[4] for m in corresponding_points], dtype=np.float64).reshape(-1, 3)
    pts = keypoints_to_array(kp)
    points_2d = np.array([pts[m[2]] for m in corresponding_points], dtype=np.float64).reshape(-1, 2)
    try:
        if flags == pnp.SOLVEPNP_AP3P:
            return pnp.p3p_4pt(points_3d[:4], points_2d[:4], camera_intrinsic)
        n = points_3d.shape[0]
        if n < 4:
            return None
        sel = np.linspace(0, n - 1, 4).astype(int)
        init = pnp.p3p_4pt(points_3d[sel], points_2d[sel], camera_intrinsic)
        if init is None:
            return None
        return pnp.refine_pnp(points_3d, points_2d, camera_intrinsic, init)
    except Exception:
        return None


def rodriguez_to_mat(rvec, tvec):
    """[R(rvec) | tvec] (utils/utils.py:495-497)."""
    from .synthetic import rodrigues
    return np.hstack((rodrigues(np.asarray(rvec, dtype=np.float64).reshape(3)),
                      np.asarray(tvec, dtype=np.float64).reshape(3, 1)))


# Hypotheses drawn, solved and scored per GPU call: the first chunk is small because the stop
# rule usually fires within a handful of iterations (every solve past it is wasted host time),
# later chunks double up to the cap.
_RANSAC_FIRST_CHUNK = 8
_RANSAC_CHUNK = 32


def ransac_for_pnp(points_to_choose_from, intrinsic_matrix, kp_left, kp_right, right_camera_matrix, thresh=2,
                   debug=False, max_iterations=100, return_supporters=False):
    """PnP-RANSAC over consensus tuples (utils/utils.py:413-464); see ransac_on_arrays."""
    if len(points_to_choose_from) < 4:
        print("too small population")
        return (None, None, None) if return_supporters else (None, None)
    X, uvL, uvR = _consensus_arrays(points_to_choose_from, kp_left, kp_right)
    Rt, ratio, mask = ransac_on_arrays(X, uvL, uvR, intrinsic_matrix, right_camera_matrix, thresh=thresh, debug=debug,
                                       max_iterations=max_iterations)
    if return_supporters:
        return Rt, ratio, mask
    return Rt, ratio


def ransac_on_arrays(X, uvL, uvR, intrinsic_matrix, right_camera_matrix, thresh=2, debug=False, max_iterations=100):
    """ransac_for_pnp on arrays: X [n,3] f64, uvL/uvR [n,2] float32.  Returns
    (Rt, inliers_ratio, supporter mask) or (None, None, None).

    Same control flow and the same consumption of the global `random` stream as the
    reference (utils/utils.py:413-464): one random.sample of 4 positions per iteration
    (sampling range(n) draws the same positions as sampling the list of n tuples), strict
    '>' best update, epsilon / iteration-count update after every scored hypothesis,
    refinement on the winner's supporters, and the quirk that the returned mask and ratio
    always belong to the refined pose.  Hypotheses are drawn in chunks; a chunk is solved in ONE
    GPU call (slam_p3p_hypotheses in place of cv2's AP3P) and scored in one more; the stop rule is
    then replayed and the random stream rewound to exactly the draws the reference would have made.
    The refinement (cv2's SOLVEPNP_ITERATIVE in the reference) is slam_refine_winners.
    """
    n_pts = X.shape[0]
    if n_pts < 4:
        print("too small population")
        return None, None, None
    K = np.asarray(intrinsic_matrix, dtype=np.float64)
    scorer = _ScoreSet(X, uvL, uvR, K, right_camera_matrix, thresh)
    population = range(n_pts)
    best_num_of_supporters = 0
    best_Rt_candidate = None
    epsilon = 0.99
    I = max_iterations
    i = 0
    chunk_size = _RANSAC_FIRST_CHUNK
    while i < I and epsilon != 0:
        chunk = min(chunk_size, I - i)
        chunk_size = min(2 * chunk_size, _RANSAC_CHUNK)
        states, samples = [], []
        for _ in range(chunk):
            states.append(random.getstate())
            samples.append(random.sample(population, k=4))
        poses, counts = scorer.solve_and_count(samples)
        solved = np.isfinite(poses).all(axis=(1, 2))
        consumed = 0
        for c in range(chunk):
            if not (i < I and epsilon != 0):
                break
            consumed += 1
            if not solved[c]:
                i += 1
                continue
            num_good_matches = int(counts[c])
            if num_good_matches > best_num_of_supporters:
                best_Rt_candidate = poses[c]
                best_num_of_supporters = num_good_matches
            epsilon = min(epsilon, 1 - (num_good_matches / n_pts))
            I = min(ransac_num_of_iterations(epsilon, 0.999), max_iterations)
            i += 1
        if consumed < chunk:
            random.setstate(states[consumed])   # the reference would not have drawn past the stop
    if best_Rt_candidate is None:
        return None, None, None
    refined_Rt = scorer.refine(best_Rt_candidate)
    are_supporters_boolean_array, num_good_matches = scorer.mask(refined_Rt)
    if num_good_matches >= best_num_of_supporters:
        best_Rt_candidate = refined_Rt
        if debug:
            print(f"after refinement: {num_good_matches} supporters out of {n_pts} matches")
    return best_Rt_candidate, num_good_matches / n_pts, are_supporters_boolean_array


# --------------------------------------------------------------------------- VO loop
def track_camera_for_many_images(thresh=1, matcher=None, num_of_frames=None, cameras=None):
    """Frame-by-frame visual odometry building the track database (utils/utils.py:236-294).

    The reference constructs its own Matcher on the KITTI images and hard-codes 2560
    frames; here both can be supplied (matcher=..., num_of_frames=...), the defaults keep
    the reference's values.  Returns (camera_positions [F,3], TrackDatabase)."""
    # The loop allocates tens of thousands of small acyclic objects per frame (DMatch, KeyPoint,
    # tuples) on top of a heap that keeps growing (the matcher's cache, the track store): the
    # cyclic collector would rescan all of it again and again for nothing to free.
    if num_of_frames is None:
        num_of_frames = _NUM_FRAMES
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        return _track_camera_for_many_images(thresh, matcher, num_of_frames, cameras)
    finally:
        if gc_was_on:
            gc.enable()


def _track_camera_for_many_images(thresh, matcher, num_of_frames, cameras):
    k, m1, m2 = cameras if cameras is not None else read_cameras()
    if matcher is None:
        matcher = Matcher(display=VERTICAL_REPRESENTATION)
    extrinsic_matrices = np.zeros(shape=(num_of_frames, 3, 4))
    camera_positions = np.zeros(shape=(num_of_frames, 3))
    left_camera_extrinsic_mat = m1
    extrinsic_matrices[0] = left_camera_extrinsic_mat
    track_db = TrackDatabase()

    frameId = 0
    matcher.read_images(frameId)
    prev_inlier_indices_mapping = match_next_pair(frameId, matcher)
    _, prev_ind_to_3d_point_dict = get_3d_points_cloud(prev_inlier_indices_mapping, k, left_camera_extrinsic_mat, m2,
                                                       matcher, file_index=frameId, debug=False)
    prev_indices_mapping = array_to_dict(prev_inlier_indices_mapping)

    for frameId in range(num_of_frames - 1):
        cur_inlier_indices_mapping = match_next_pair(frameId + 1, matcher)
        cur_indices_mapping = array_to_dict(cur_inlier_indices_mapping)
        _, cur_ind_to_3d_point_dict = get_3d_points_cloud(cur_inlier_indices_mapping, k, left_camera_extrinsic_mat, m2,
                                                          matcher, file_index=frameId + 1, debug=False)
        consecutive_matches = matcher.match_between_consecutive_frames(frameId, frameId + 1, thresh=thresh)
        matches_in_4, _ = get_matches_that_appear_in_all_4_images(consecutive_matches, prev_indices_mapping,
                                                                  cur_indices_mapping, prev_ind_to_3d_point_dict)
        kp1, kp2 = matcher.get_kp(idx=frameId + 1)
        Rt, inliers_ratio, are_supporters = ransac_for_pnp(matches_in_4, k, kp1, kp2, m2, thresh=PIXEL_THRESH,
                                                           debug=False, max_iterations=VO_MAX_ITERATIONS,
                                                           return_supporters=True)
        assert are_supporters is not None, \
            f"RANSAC Couldn't find a transformation from frame {frameId} to frame {frameId + 1}..."
        Rt_inliers = [p for ind, p in enumerate(matches_in_4) if are_supporters[ind]]
        num_inliers = np.sum(are_supporters)
        track_db.update_tracks_from_frame(matcher, frameId, Rt_inliers)
        track_db.add_inliers_ratio(frameId, inliers_ratio)
        track_db.add_num_matches(frameId, num_matches=num_inliers)
        R, t = Rt[:, :-1], Rt[:, -1]
        new_R = R @ extrinsic_matrices[frameId][:, :-1]
        new_t = R @ extrinsic_matrices[frameId][:, -1] + t
        extrinsic_matrices[frameId + 1] = np.hstack((new_R, new_t[:, None]))
        camera_positions[frameId + 1] = -new_R.T @ new_t
        prev_ind_to_3d_point_dict = cur_ind_to_3d_point_dict
        prev_indices_mapping = cur_indices_mapping
        track_db.set_ex_camera_positions(camera_positions)
        track_db.set_ex_matrices(extrinsic_matrices)
    return camera_positions, track_db


# --------------------------------------------------------------------------- loop closure
def find_4_images_matches(consecutive_matches, prev_indices_mapping, cur_indices_mapping, ind_to_3d_point_prev,
                          matcher, reference_kf, candidate_fk):
    """The join above plus the (xl, xr, yl) track pair of every consensus match
    (models/LoopClosure.py:71-109)."""
    consensus, filtered = get_matches_that_appear_in_all_4_images(consecutive_matches, prev_indices_mapping,
                                                                  cur_indices_mapping, ind_to_3d_point_prev)
    tracks = []
    for prev_left, prev_right, cur_left, cur_right, _ in consensus:
        xl, yl = matcher.get_feature_location_frame(reference_kf, kp=prev_left, loc=LEFT)
        xr, _ = matcher.get_feature_location_frame(reference_kf, kp=prev_right, loc=RIGHT)
        xl_n, yl_n = matcher.get_feature_location_frame(candidate_fk, kp=cur_left, loc=LEFT)
        xr_n, _ = matcher.get_feature_location_frame(candidate_fk, kp=cur_right, loc=RIGHT)
        tracks.append([(xl, xr, yl), (xl_n, xr_n, yl_n)])
    return consensus, tracks, filtered


def consensus_matching_of_general_two_frames(reference_kf, candidate_kf, matcher, thresh=0.6, cameras=None):
    """Consensus matching of two arbitrary keyframes for loop closure
    (models/LoopClosure.py:112-139): stereo matching + triangulation of both, left-left
    2-NN matching at ratio `thresh`, four-image join, PnP-RANSAC with 500 iterations.
    Returns (Rt, filtered_tracks, inliers_ratio, filtered_matches, supporter_mask,
    n_consensus) or six Nones."""
    from .constants import LC_MAX_ITERATIONS, LC_MIN_CONSENSUS
    k, m1, m2 = cameras if cameras is not None else read_cameras()
    matcher.read_images(reference_kf)
    prev_inlier_indices_mapping = match_next_pair(reference_kf, matcher)
    _, prev_ind_to_3d_point_dict = get_3d_points_cloud(prev_inlier_indices_mapping, k, m1, m2, matcher,
                                                       file_index=reference_kf, debug=False)
    prev_indices_mapping = array_to_dict(prev_inlier_indices_mapping)
    cur_inlier_indices_mapping = match_next_pair(candidate_kf, matcher)
    cur_indices_mapping = array_to_dict(cur_inlier_indices_mapping)
    consecutive_matches = matcher.match_between_any_frames(reference_kf, candidate_kf, thresh=thresh)
    matches_on_all_images, tracks, filtered_matches = find_4_images_matches(
        consecutive_matches, prev_indices_mapping, cur_indices_mapping, prev_ind_to_3d_point_dict, matcher,
        reference_kf, candidate_kf)
    if len(matches_on_all_images) < LC_MIN_CONSENSUS:
        return None, None, None, None, None, None
    kp1, kp2 = matcher.get_kp(idx=candidate_kf)
    Rt, inliers_ratio, supporter_indices = ransac_for_pnp(matches_on_all_images, k, kp1, kp2, m2, thresh=PIXEL_THRESH,
                                                          debug=False, max_iterations=LC_MAX_ITERATIONS,
                                                          return_supporters=True)
    if Rt is None:
        return None, None, None, None, None, None
    filtered_tracks = [track for i, track in enumerate(tracks) if supporter_indices[i]]
    return Rt, filtered_tracks, inliers_ratio, filtered_matches, supporter_indices, len(matches_on_all_images)
