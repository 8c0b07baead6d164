#!/usr/bin/env python3
"""Generates tests/golden/trackdb_golden.npz from the REFERENCE's own track database.

models/TrackDatabase.py needs numpy only, so -- in the build container, where /root/reference
exists -- it is imported as is and driven exactly as the visual-odometry loop drives it
(utils/utils.py:277-282: update_tracks_from_frame with the inlier consensus tuples, then
add_inliers_ratio and add_num_matches), through a stand-in matcher that answers
get_feature_location_frame (models/Matcher.py:73-76) from keypoint arrays.  The resulting state
and the answers of its getters are stored as plain arrays: inputs and expected outputs only,
nothing of the reference's source.

Two scenarios:
  mini   the 8-frame, 256-keypoint synthetic sequence end to end: consensus tuples of every pair
         (the numpy restatement of the front-end), 24 seeded hypotheses per pair, the first best
         one's supporters as the inliers -- what the GPU pipeline's winner tables must reproduce;
  rand   40 pairs of random inlier tables with track continuations, repeated keypoint indices
         inside a pair (two rows continuing one track, two rows ending on one keypoint) and
         pairs without any inlier.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_trackdb_golden.py
"""
import os
import sys

sys.dont_write_bytecode = True          # nothing may be written next to the reference's files

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import numpy_twin as twin                                   # noqa: E402
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses  # noqa: E402
from trackdb_state import query_arrays, state_arrays                    # noqa: E402


def reference_class():
    sys.path.insert(0, REFERENCE)
    try:
        from models.TrackDatabase import TrackDatabase                  # the reference's class, unmodified
    finally:
        sys.path.remove(REFERENCE)
    return TrackDatabase


class ArrayMatcher:
    """get_feature_location_frame over per-frame keypoint arrays: kp[frame] = (left [N,2], right [M,2])."""

    def __init__(self, kp):
        self.kp = kp

    def get_feature_location_frame(self, frameId, kp, loc):
        x, y = self.kp[frameId][loc][kp]
        return float(x), float(y)


def drive(RefDB, matcher, first_frame, rows, n_inl, n_cons):
    db = RefDB()
    off = 0
    for p, m in enumerate(n_inl):
        frame = first_frame + p
        tuples = [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), None) for r in rows[off:off + m]]
        db.update_tracks_from_frame(matcher, frame, tuples)
        db.add_inliers_ratio(frame, (m / n_cons[p]) if n_cons[p] else 0.0)
        db.add_num_matches(frame, num_matches=np.float64(m))     # np.sum of a float64 0/1 array (utils/utils.py:279,390)
        off += m
    return db


def mini_sequence():
    n_kp, n_frames, H = 256, 8, 24
    k, m1, m2 = read_cameras()
    seq = SyntheticSequence(n_kp, n_frames)
    state = []
    for fr in seq.frames:
        matches = twin.knn_match(fr.descL, fr.descR, 1)
        _, y1, _, y2, im = twin.coords_from_kps(matches, fr.kpL, fr.kpR)
        mapping = twin.rectified_stereo_pattern(y1, y2, im, thresh=2)
        _, d3 = twin.get_3d_points_cloud(mapping, k, m1, m2, fr.kpL, fr.kpR)
        state.append((twin.array_to_dict(mapping), d3))
    rows, n_inl, n_cons, best = [], [], [], []
    for p in range(n_frames - 1):
        f0, f1 = seq.frames[p], seq.frames[p + 1]
        consecutive = twin.apply_thresholds(twin.knn_match(f0.descL, f1.descL, 2), 1)
        cons, _ = twin.matches_in_all_4_images(consecutive, state[p][0], state[p + 1][0], state[p][1])
        Rt = seeded_hypotheses(seq.relative_pose(p, p + 1), H, seed=6 + p)
        top, top_mask, top_h = 0, None, -1
        for h in range(H):
            good, num = twin.find_supporters(Rt[h], m2, cons, k, f1.kpL, f1.kpR, thresh=2)
            if num > top:                                   # strict '>': the first best wins (utils/utils.py:438)
                top, top_mask, top_h = num, good.astype(bool), h
        sel = [c[:4] for c, g in zip(cons, top_mask) if g]
        rows.append(np.array(sel, dtype=np.int32).reshape(-1, 4))
        n_inl.append(len(sel))
        n_cons.append(len(cons))
        best.append(top_h)
    kp = {f: (fr.kpL, fr.kpR) for f, fr in enumerate(seq.frames)}
    return dict(n_kp=n_kp, n_frames=n_frames, H=H, rows=np.concatenate(rows), n_inl=np.array(n_inl, np.int64),
                n_cons=np.array(n_cons, np.int64), best=np.array(best, np.int64)), kp


def random_tables(n_pairs, n_kp, seed):
    rng = np.random.default_rng(seed)
    rows, n_inl = [], []
    prev_cur = rng.permutation(n_kp)[: n_kp // 2]
    for p in range(n_pairs):
        m = int(rng.integers(1, n_kp // 2)) if p % 9 != 4 else 0
        cont = prev_cur[rng.random(prev_cur.shape[0]) < 0.7][:m]
        fresh = rng.integers(0, n_kp, size=m - cont.shape[0])
        prev_left = np.concatenate([cont, fresh]).astype(np.int32)
        rng.shuffle(prev_left)
        cur_left = rng.integers(0, n_kp, size=m).astype(np.int32)           # repeated indices on purpose
        if m > 6:
            prev_left[3] = prev_left[1]                                     # two rows continue the same track
            cur_left[5] = cur_left[2]                                       # two rows end on the same keypoint
        r = np.stack([prev_left, rng.integers(0, n_kp, m), cur_left, rng.integers(0, n_kp, m)], axis=1)
        rows.append(r.astype(np.int32).reshape(-1, 4))
        n_inl.append(m)
        prev_cur = cur_left if m else prev_cur
    kp = {f: (rng.uniform(0, 1226, (n_kp, 2)).astype(np.float32), rng.uniform(0, 1226, (n_kp, 2)).astype(np.float32))
          for f in range(3, 3 + n_pairs + 1)}
    n_inl = np.array(n_inl, np.int64)
    return dict(first_frame=3, n_kp=n_kp, rows=np.concatenate(rows), n_inl=n_inl,
                n_cons=n_inl + rng.integers(0, 20, n_pairs)), kp


def locations(rows, n_inl, kp, first_frame):
    """[T,6] float32 (xl, xr, yl of the earlier frame, then of the later one) of every inlier row:
    the table form of what the stand-in matcher answers one keypoint at a time."""
    out, off = [], 0
    for p, m in enumerate(n_inl):
        r = rows[off:off + m]
        (pl, pr), (cl, cr) = kp[first_frame + p], kp[first_frame + p + 1]
        out.append(np.stack([pl[r[:, 0], 0], pr[r[:, 1], 0], pl[r[:, 0], 1], cl[r[:, 2], 0], cr[r[:, 3], 0],
                             cl[r[:, 2], 1]], axis=1).astype(np.float32).reshape(-1, 6))
        off += m
    return np.concatenate(out)


def main():
    RefDB = reference_class()
    out = {}
    mini, kp = mini_sequence()
    db = drive(RefDB, ArrayMatcher(kp), 0, mini["rows"], mini["n_inl"], mini["n_cons"])
    for k_, v in mini.items():
        out[f"mini_{k_}"] = np.asarray(v)
    out["mini_loc"] = locations(mini["rows"], mini["n_inl"], kp, 0)
    for k_, v in {**state_arrays(db), **query_arrays(db)}.items():
        out[f"mini_state_{k_}"] = v
    rnd, kp = random_tables(40, 120, seed=17)
    db = drive(RefDB, ArrayMatcher(kp), rnd["first_frame"], rnd["rows"], rnd["n_inl"], rnd["n_cons"])
    for k_, v in rnd.items():
        out[f"rand_{k_}"] = np.asarray(v)
    out["rand_loc"] = locations(rnd["rows"], rnd["n_inl"], kp, rnd["first_frame"])
    out["rand_kpL"] = np.stack([kp[f][0] for f in sorted(kp)])
    out["rand_kpR"] = np.stack([kp[f][1] for f in sorted(kp)])
    for k_, v in {**state_arrays(db), **query_arrays(db)}.items():
        out[f"rand_state_{k_}"] = v
    path = os.path.join(HERE, "trackdb_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", "mini tracks", int(out["mini_state_counters"][1]),
          "rand tracks", int(out["rand_state_counters"][1]))


if __name__ == "__main__":
    main()
