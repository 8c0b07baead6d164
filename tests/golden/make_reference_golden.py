#!/usr/bin/env python3
"""Generates tests/golden/reference_golden.npz by EXECUTING THE REFERENCE'S OWN FUNCTION BODIES.

The reference's front-end modules cannot be imported here (utils/utils.py, models/Matcher.py and
models/LoopClosure.py import cv2 / gtsam at the top), but most of the functions on the hot path
need numpy only.  In the build container -- where /root/reference exists -- this script parses
those three files with `ast`, takes the listed function definitions UNCHANGED (no edit of any
node, no stand-in for cv2 or gtsam, no module import), compiles each on its own and runs it on
a synthetic mini-sequence.  Only arrays are stored: inputs and the outputs the reference's code
produced from them.  No source text of the reference is kept anywhere.

Functions taken from utils/utils.py: coords_from_kps (:42), read_cameras (:75, on the reference's
own calib.txt), _rectified_stereo_classifier (:89), rectificatied_stereo_pattern (:103),
array_to_dict (:156), match_next_pair (:297), find_stereo_matches (:302), get_3d_points_cloud
(:318), get_matches_that_appear_in_all_4_images (:338), find_supporters (:389),
ransac_num_of_iterations (:475), least_squares (:527).
From models/LoopClosure.py: find_4_images_matches (:71).
Methods of models/Matcher.py's class, bound to a carrier object: get_matches (:55),
get_filtered_matches (:60), set_matches (:65), get_kp (:68), get_feature_location_frame (:73),
detect_and_compute (:111), find_matching_features (:121), apply_threshold (:148), filter_matches
(:162), apply_thresholds (:178), match_between_consecutive_frames (:187),
match_between_any_frames (:206); the module constant FILTERED_MATCHES (:10) and the cache keys of
models/Constants.py (imported as it is: it needs nothing).

What the carrier supplies itself, because the reference's versions call cv2: the constructor
and read_images (cv2.imread, utils/utils.py:30-33) -- here an "image" is a (frame, side) token;
`algo.detectAndCompute(token)` -- the reference's algo= plug point, models/Matcher.py:15,116 --
answers with the synthetic keypoints and descriptors of that frame; `matcher.knnMatch` -- the
matcher= plug point, :29,127,140,199,219 -- is the numpy restatement of cv2's scan
(oracle/numpy_twin.py).  So the brute-force matching itself (cv2) stays unpinned: its lists are
INPUTS of this fixture, everything computed from them is the reference's own arithmetic.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_golden.py
"""
import ast
import os
import sys

sys.dont_write_bytecode = True          # nothing may be written next to the reference's files

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import numpy_twin as twin                                   # noqa: E402
from slam_amd.synthetic import SyntheticSequence, seeded_hypotheses     # noqa: E402

UTILS_FUNCS = ["coords_from_kps", "read_cameras", "_rectified_stereo_classifier", "rectificatied_stereo_pattern",
               "array_to_dict", "match_next_pair", "find_stereo_matches", "get_3d_points_cloud",
               "get_matches_that_appear_in_all_4_images", "find_supporters", "ransac_num_of_iterations",
               "least_squares"]
LC_FUNCS = ["find_4_images_matches"]
MATCHER_METHODS = ["get_matches", "get_filtered_matches", "set_matches", "get_kp", "get_feature_location_frame",
                   "detect_and_compute", "find_matching_features", "apply_threshold", "filter_matches",
                   "apply_thresholds", "match_between_consecutive_frames", "match_between_any_frames"]

N_KP, N_FRAMES, N_HYP = 256, 8, 24
LC_PAIRS = [(0, 5), (2, 7), (6, 1)]


# --------------------------------------------------------------------------- extraction
def _defs(path, names, cls=None):
    """{name: FunctionDef node} of the named module-level functions (or methods of class `cls`)."""
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    found = {n.name: n for n in body if isinstance(n, ast.FunctionDef) and n.name in names}
    missing = set(names) - set(found)
    if missing:
        raise SystemExit(f"{path}: no definition of {sorted(missing)}")
    return found, tree


def _run_defs(path, nodes, namespace):
    """Compile each definition on its own, unchanged, and execute it in `namespace`."""
    for node in nodes.values():
        code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
        exec(code, namespace)


def reference_functions():
    sys.path.insert(0, REFERENCE)
    try:
        import models.Constants as C                                   # the reference's constants, unmodified
    finally:
        sys.path.remove(REFERENCE)
    consts = {k: v for k, v in vars(C).items() if not k.startswith("_")}

    upath = os.path.join(REFERENCE, "utils", "utils.py")
    u_nodes, _ = _defs(upath, UTILS_FUNCS)
    utils_ns = {"np": np, "DATA_PATH": os.path.join(REFERENCE, "VAN_ex", "dataset", "sequences", "05") + os.sep}
    _run_defs(upath, u_nodes, utils_ns)

    mpath = os.path.join(REFERENCE, "models", "Matcher.py")
    m_nodes, m_tree = _defs(mpath, MATCHER_METHODS, cls="Matcher")
    matcher_ns = {"np": np, **consts}
    for node in m_tree.body:                                           # FILTERED_MATCHES = "FILTERED_MATCHES" (:10)
        if isinstance(node, ast.Assign) and any(getattr(t, "id", None) == "FILTERED_MATCHES" for t in node.targets):
            exec(compile(ast.Module(body=[node], type_ignores=[]), mpath, "exec"), matcher_ns)
    _run_defs(mpath, m_nodes, matcher_ns)

    lpath = os.path.join(REFERENCE, "models", "LoopClosure.py")
    l_nodes, _ = _defs(lpath, LC_FUNCS)
    lc_ns = {"np": np, **consts}
    _run_defs(lpath, l_nodes, lc_ns)
    return utils_ns, matcher_ns, lc_ns, consts


# --------------------------------------------------------------------------- data carriers
class KeyPointData:
    """What the reference reads of a cv2.KeyPoint on this path: .pt, two Python floats holding
    float32 values."""
    __slots__ = ("pt",)

    def __init__(self, x, y):
        self.pt = (float(np.float32(x)), float(np.float32(y)))


class FeatureStore:
    """algo= plug: detectAndCompute((frame, side), None) -> (tuple of keypoints, descriptors)."""

    def __init__(self, frames):
        self.frames = frames

    def detectAndCompute(self, image, mask):
        frame, side = image
        f = self.frames[frame]
        kp, dsc = (f.kpL, f.descL) if side == 0 else (f.kpR, f.descR)
        return tuple(KeyPointData(x, y) for x, y in kp), dsc


class ScanMatcher:
    """matcher= plug: cv2.BFMatcher(NORM_HAMMING).knnMatch restated (oracle/numpy_twin.py); remembers
    every result so that the lists handed to the reference's code are stored as the fixture's inputs."""

    def __init__(self):
        self.calls = []

    def knnMatch(self, q, t, k):
        res = twin.knn_match(q, t, k)
        idx = np.full((len(res), 2), -1, np.int32)
        dist = np.full((len(res), 2), np.iinfo(np.int32).max, np.int32)
        for i, row in enumerate(res):
            for c, m in enumerate(row):
                idx[i, c], dist[i, c] = m.trainIdx, int(m.distance)
        self.calls.append((k, idx, dist))
        return res


def carrier_class(matcher_ns, consts):
    """A class named Matcher (match_between_* call Matcher.apply_thresholds by that name) whose
    matching / cache methods are the reference's; constructor and read_images are this script's."""
    FRAMES, LEFT, RIGHT = consts["FRAMES"], consts["LEFT"], consts["RIGHT"]

    def __init__(self, algo, matcher, threshold=.6):
        # the fields models/Matcher.py:25-37 sets, without reading an image
        self.cache = {}
        self._file_index = None
        self.algo, self.matcher, self.threshold = algo, matcher, threshold
        self._img1 = self._img2 = None
        self._filtered_matches = self._matches = None
        self._img1_kp = self._img1_dsc = self._img2_kp = self._img2_dsc = None

    def read_images(self, idx):
        # models/Matcher.py:39-53 with (frame, side) tokens where cv2.imread would give pixels
        self._file_index = idx
        self._img1, self._img2 = (idx, 0), (idx, 1)
        if idx not in self.cache:
            self.cache[idx] = {FRAMES: (self._img1, self._img2), LEFT: None, RIGHT: None}
        else:
            self.cache[idx][FRAMES] = (self._img1, self._img2)

    members = {name: matcher_ns[name] for name in MATCHER_METHODS}
    members.update(__init__=__init__, read_images=read_images)
    cls = type("Matcher", (), members)
    matcher_ns["Matcher"] = cls
    return cls


# --------------------------------------------------------------------------- helpers
def _qt(match_lists):
    """[(queryIdx, trainIdx)] of a list of 1-element match lists."""
    return np.array([(m[0].queryIdx, m[0].trainIdx) for m in match_lists], dtype=np.int32).reshape(-1, 2)


def _rows(consensus):
    return np.array([c[:4] for c in consensus], dtype=np.int32).reshape(-1, 4)


def _points(consensus):
    return np.array([c[4] for c in consensus], dtype=np.float64).reshape(-1, 3)


def knife_edge_case(utils_ns, k, m2, rng):
    """Consensus tuples whose reprojection distance in one image is planted within 1e-7..1e-2 px of the
    2-pixel threshold (as near as float32 pixel coordinates allow), scored by the reference's find_supporters."""
    n = 96
    X = np.stack([rng.uniform(-20, 20, n), rng.uniform(-3, 2, n), np.exp(rng.uniform(np.log(5), np.log(90), n))], axis=1)
    Rt = seeded_hypotheses(np.hstack([np.eye(3), np.array([[0.02], [-0.01], [-0.8]])]), 3, seed=11)[0]
    left = X @ Rt[:, :3].T + Rt[:, 3]
    right = left + m2[:, -1]
    pl = (left @ k.T)[:, :2] / (left @ k.T)[:, 2:]
    pr = (right @ k.T)[:, :2] / (right @ k.T)[:, 2:]
    ang = rng.uniform(0, 2 * np.pi, n)
    delta = np.tile([0.0, 1e-7, -1e-7, 1e-5, -1e-5, 1e-4, -1e-4, 1e-3, -1e-3, 1e-2, -1e-2, -0.5], n // 12)
    off = (2.0 + delta)[:, None] * np.stack([np.cos(ang), np.sin(ang)], axis=1)
    off[:8] = [[2, 0], [0, 2], [-2, 0], [0, -2], [1.2, 1.6], [1.6, -1.2], [-1.2, 1.6], [-1.6, -1.2]]
    which = np.arange(n) % 2 == 0                       # plant on the left image for even rows, right for odd
    uvL = (pl + np.where(which[:, None], off, 0.3)).astype(np.float32)
    uvR = (pr + np.where(which[:, None], 0.3, off)).astype(np.float32)
    kpl = [KeyPointData(x, y) for x, y in uvL]
    kpr = [KeyPointData(x, y) for x, y in uvR]
    cons = [(i, i, i, i, X[i]) for i in range(n)]
    good, num = utils_ns["find_supporters"](Rt, m2, cons, k, kpl, kpr, thresh=2, debug=False)
    return dict(knife_X=X, knife_Rt=Rt, knife_uvL=uvL, knife_uvR=uvR, knife_mask=good.astype(bool), knife_count=np.int64(num))


# --------------------------------------------------------------------------- main
def main():
    utils_ns, matcher_ns, lc_ns, consts = reference_functions()
    U = utils_ns
    RefMatcher = carrier_class(matcher_ns, consts)
    MATCHES, CONSECUTIVE = consts["MATCHES"], consts["CONSECUTIVE"]
    out = {}

    # read_cameras on the reference's calib.txt (utils/utils.py:75-86)
    k, m1, m2 = U["read_cameras"]()
    out["cam_k"], out["cam_m1"], out["cam_m2"] = k, m1, m2

    seq = SyntheticSequence(N_KP, N_FRAMES)
    out["in_n_kp"], out["in_n_frames"], out["in_n_hyp"] = np.int64(N_KP), np.int64(N_FRAMES), np.int64(N_HYP)
    out["in_descL"] = np.stack([f.descL for f in seq.frames])
    out["in_descR"] = np.stack([f.descR for f in seq.frames])
    out["in_kpL"] = np.stack([f.kpL for f in seq.frames])
    out["in_kpR"] = np.stack([f.kpR for f in seq.frames])

    scan = ScanMatcher()
    matcher = RefMatcher(FeatureStore(seq.frames), scan)

    # ---- the VO loop's per-frame and per-pair steps, in the order utils/utils.py:251-290 makes them
    def stereo(f):
        """match_next_pair + get_3d_points_cloud + array_to_dict of frame f; records every intermediate."""
        matcher.read_images(f) if f == 0 else None
        mapping = U["match_next_pair"](f, matcher)
        k_, idx, dist = scan.calls[-1]
        assert k_ == 1
        out[f"in_st_idx{f}"], out[f"in_st_dist{f}"] = idx[:, 0].copy(), dist[:, 0].copy()
        # the pieces find_stereo_matches is made of, once more on their own for their outputs
        filtered = matcher.get_filtered_matches(f)
        kp1, kp2 = matcher.get_kp(f)
        x1, y1, x2, y2, im = U["coords_from_kps"](filtered, kp1, kp2)
        out[f"st_x1_{f}"], out[f"st_y1_{f}"], out[f"st_x2_{f}"], out[f"st_y2_{f}"] = x1, y1, x2, y2
        out[f"st_all{f}"] = im.astype(np.int32)
        for thr in (1, 2):
            i1, i2, o1, o2, imap = U["rectificatied_stereo_pattern"](y1, y2, im, thresh=thr)
            out[f"st{thr}_in1_{f}"], out[f"st{thr}_in2_{f}"] = i1.astype(np.int32), i2.astype(np.int32)
            out[f"st{thr}_out1_{f}"], out[f"st{thr}_out2_{f}"] = o1.astype(np.int32), o2.astype(np.int32)
            out[f"st{thr}_map{f}"] = imap.astype(np.int32)
        assert np.array_equal(out[f"st2_map{f}"], mapping)
        out[f"st_kept{f}"] = _qt(matcher.get_matches(f))               # MATCHES after filter_matches (:162-172)
        assert matcher.cache[f][MATCHES] is matcher.get_matches()
        pts, d3 = U["get_3d_points_cloud"](mapping, k, m1, m2, matcher, file_index=f, debug=False)
        assert len(pts) == len(d3) and all(a is b for a, b in zip(pts, d3.values()))
        out[f"pc_keys{f}"] = np.array(list(d3.keys()), dtype=np.int32)
        out[f"pc_pts{f}"] = np.array(list(d3.values()), dtype=np.float64).reshape(-1, 3)
        smap = U["array_to_dict"](mapping)
        out[f"map_keys{f}"] = np.array(list(smap.keys()), dtype=np.int32)
        out[f"map_vals{f}"] = np.array(list(smap.values()), dtype=np.int32)
        return smap, d3

    state = [stereo(0)]
    for p in range(N_FRAMES - 1):
        state.append(stereo(p + 1))
        consecutive = matcher.match_between_consecutive_frames(p, p + 1, thresh=1)
        k_, idx, dist = scan.calls[-1]
        assert k_ == 2 and matcher.cache[p + 1][CONSECUTIVE] is consecutive
        out[f"in_tm_idx{p}"], out[f"in_tm_dist{p}"] = idx, dist
        out[f"tm_kept{p}"] = _qt(consecutive)
        cons, filtered = U["get_matches_that_appear_in_all_4_images"](consecutive, state[p][0], state[p + 1][0],
                                                                    state[p][1])
        out[f"cons_rows{p}"], out[f"cons_X{p}"], out[f"cons_filtered{p}"] = _rows(cons), _points(cons), _qt(filtered)
        kp1, kp2 = matcher.get_kp(idx=p + 1)
        Rt = seeded_hypotheses(seq.relative_pose(p, p + 1), N_HYP, seed=6 + p)
        masks = np.zeros((N_HYP, len(cons)), dtype=bool)
        counts = np.zeros(N_HYP, dtype=np.int64)
        for h in range(N_HYP):
            good, num = U["find_supporters"](Rt[h], m2, cons, k, kp1, kp2, thresh=2, debug=False)
            masks[h], counts[h] = good.astype(bool), num
        out[f"sup_Rt{p}"], out[f"sup_masks{p}"], out[f"sup_counts{p}"] = Rt, masks, counts

    # ---- ratio test at the other thresholds the reference uses (models/Matcher.py:15,178-184; exercise_1.py:123)
    for p in (0, 3):
        lists = twin.knn_match(seq.frames[p].descL, seq.frames[p + 1].descL, 2)
        for thr in (0.2, 0.6):
            out[f"ratio{int(thr * 10)}_kept{p}"] = _qt(matcher_ns["apply_thresholds"](lists, thr))
    # the instance form with its significance test (models/Matcher.py:121-130,148-160)
    m2nd = RefMatcher(FeatureStore(seq.frames), ScanMatcher(), threshold=0.6)
    m2nd.read_images(1)
    m2nd.detect_and_compute(1)
    m2nd.find_matching_features(with_significance_test=True)
    out["sig_kept1"] = _qt(m2nd.get_filtered_matches(1))

    # ---- loop-closure consensus matching of arbitrary frames (models/LoopClosure.py:112-127, its pure part)
    out["lc_pairs"] = np.array(LC_PAIRS, dtype=np.int32)
    for n, (ref, cand) in enumerate(LC_PAIRS):
        lcm = RefMatcher(FeatureStore(seq.frames), ScanMatcher())
        lcm.read_images(ref)
        pmap = U["match_next_pair"](ref, lcm)
        _, p3d = U["get_3d_points_cloud"](pmap, k, m1, m2, lcm, file_index=ref, debug=False)
        pdict = U["array_to_dict"](pmap)
        cdict = U["array_to_dict"](U["match_next_pair"](cand, lcm))
        consecutive = lcm.match_between_any_frames(ref, cand, thresh=0.6)
        cons, tracks, filtered = lc_ns["find_4_images_matches"](consecutive, pdict, cdict, p3d, lcm, ref, cand)
        out[f"lc_tm_kept{n}"] = _qt(consecutive)
        out[f"lc_rows{n}"], out[f"lc_X{n}"], out[f"lc_filtered{n}"] = _rows(cons), _points(cons), _qt(filtered)
        out[f"lc_tracks{n}"] = np.array(tracks, dtype=np.float64).reshape(-1, 2, 3)

    # ---- least_squares on the DLT inputs of frontend_golden.npz (disparity 0.05..200 px), P = K@M1, Q = K@M2
    G = np.load(os.path.join(HERE, "frontend_golden.npz"))
    P, Q = k @ m1, k @ m2
    x1, y1, x2, y2 = (G[n_] for n_ in ("dlt_x1", "dlt_y1", "dlt_x2", "dlt_y2"))
    out["dlt_x1"], out["dlt_y1"], out["dlt_x2"], out["dlt_y2"] = x1, y1, x2, y2
    out["dlt_X"] = np.array([U["least_squares"]((float(x1[i]), float(y1[i])), (float(x2[i]), float(y2[i])), P, Q)
                             for i in range(len(x1))])

    # ---- supporters at the edge of the threshold
    out.update(knife_edge_case(U, k, m2, np.random.default_rng(61)))

    # ---- ransac_num_of_iterations over the epsilons ransac_for_pnp can reach (utils/utils.py:443-444)
    eps = sorted({min(0.99, 1 - n / N) for N in (10, 137, 300, 2000) for n in range(N + 1)})
    eps = np.array(eps, dtype=np.float64)
    out["iters_eps"] = eps
    for name, p in (("iters_p999", 0.999), ("iters_p99", 0.99)):
        out[name] = np.array([U["ransac_num_of_iterations"](float(e), p) for e in eps], dtype=np.int64)
    out["iters_default"] = np.int64(U["ransac_num_of_iterations"]())

    path = os.path.join(HERE, "reference_golden.npz")
    np.savez_compressed(path, **out)
    n_cons = [len(out[f"cons_rows{p}"]) for p in range(N_FRAMES - 1)]
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays; consensus per pair", n_cons,
          "; lc", [len(out[f"lc_rows{n}"]) for n in range(len(LC_PAIRS))], "; knife", int(out["knife_count"]), "of 96")


if __name__ == "__main__":
    main()
