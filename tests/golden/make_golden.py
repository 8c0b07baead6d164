#!/usr/bin/env python3
"""Generates tests/golden/frontend_golden.npz.

The reference holds no fixtures for this path and cannot be imported here (it needs cv2 /
gtsam), so the vectors are produced by oracle/numpy_twin.py -- the statement-by-statement
restatement of the reference's Python, which calls the same numpy routines the reference
calls (np.linalg.svd, '@', np.linalg.norm) and restates cv2's knnMatch scan -- run in
this container (numpy 2.2.6).  Nothing from the reference is executed.  The C oracle and
the HIP kernels are then both checked against this file.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import numpy_twin as twin                      # noqa: E402
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses  # noqa: E402


def knn_arrays(q, t, k=2):
    res = twin.knn_match(q, t, k)
    idx = np.full((len(q), 2), -1, np.int32)
    dist = np.full((len(q), 2), np.iinfo(np.int32).max, np.int32)
    for i, row in enumerate(res):
        for c, m in enumerate(row):
            idx[i, c], dist[i, c] = m.trainIdx, int(m.distance)
    return idx, dist


def main():
    rng = np.random.default_rng(6)
    out = {}

    # (i) k-NN with ties, duplicates, tiny train sets, D in {32, 61, 64}
    cases = []
    for D in (32, 61, 64):
        base = rng.integers(0, 256, (12, D), dtype=np.uint8)
        cases.append((np.concatenate([base, rng.integers(0, 256, (9, D), dtype=np.uint8)]),
                      np.concatenate([base, base[::-1], base])))                       # duplicates
        cases.append(((rng.integers(0, 256, (30, D), dtype=np.uint8) & 1).astype(np.uint8),
                      (rng.integers(0, 256, (70, D), dtype=np.uint8) & 1).astype(np.uint8)))  # heavy collisions
        cases.append((base, np.repeat(base[:1], 5, axis=0)))                                 # all equal
        cases.append((base, base[:1]))                                                       # Nt = 1
        cases.append((base, base[:2]))                                                       # Nt = 2
        cases.append((base, np.zeros((0, D), np.uint8)))                                     # Nt = 0
    out["knn_n"] = np.array(len(cases))
    for n, (q, t) in enumerate(cases):
        idx, dist = knn_arrays(q, t)
        out[f"knn_q{n}"], out[f"knn_t{n}"], out[f"knn_idx{n}"], out[f"knn_dist{n}"] = q, t, idx, dist

    # (ii) ratio test: Python float compare m.distance < thr * n.distance (models/Matcher.py:182)
    d = np.arange(0, 489)
    thrs = np.array([0.2, 0.4, 0.6, 1.0])
    grid = np.zeros((len(thrs), len(d), len(d)), dtype=bool)
    for ti, thr in enumerate(thrs):
        for a in d:
            fa = float(np.float32(a))
            for b in d:
                grid[ti, a, b] = fa < float(thr) * float(np.float32(b))
    out["ratio_thr"], out["ratio_grid"] = thrs, np.packbits(grid, axis=None)

    # (iii) row filter at the threshold edge: np.abs(y1 - y2) <= thresh on float32-origin floats
    y1 = np.float32([10.0, 10.0, 10.0, 10.0, 100.25, 100.25, 0.0, 369.5])
    y2 = np.float32([12.0, np.nextafter(np.float32(12), np.float32(13)), np.nextafter(np.float32(12), np.float32(11)),
                     8.0, 102.25, 98.2499, 2.0000002, 367.5])
    out["row_y1"], out["row_y2"] = y1, y2
    for thr in (1, 2):
        out[f"row_keep{thr}"] = (np.abs(y1.astype(np.float64) - y2.astype(np.float64)) <= thr)

    # (iv) DLT: np.linalg.svd per point (utils/utils.py:553), disparity 0.05 .. 200 px
    k, m1, m2 = read_cameras()
    P, Q = k @ m1, k @ m2
    n = 400
    disp = np.exp(rng.uniform(np.log(0.05), np.log(200.0), n))
    x1 = rng.uniform(250, 1200, n).astype(np.float32)
    yy = rng.uniform(0, 370, n).astype(np.float32)
    x2 = (x1 - disp).astype(np.float32)
    y2d = (yy + rng.normal(0, 0.7, n)).astype(np.float32)
    x2[:10] = x1[:10] + 2.5
    sol = np.array([twin.least_squares((float(x1[i]), float(yy[i])), (float(x2[i]), float(y2d[i])), P, Q)
                    for i in range(n)])
    out["dlt_P"], out["dlt_Q"] = P, Q
    out["dlt_x1"], out["dlt_y1"], out["dlt_x2"], out["dlt_y2"], out["dlt_X"] = x1, yy, x2, y2d, sol
    out["dlt_valid"] = ~((sol[:, 2] <= 0) | (x1.astype(np.float64) < x2.astype(np.float64)))

    # (vi) mini sequence end to end through the twin: 4 frames, 160 keypoints
    seq = SyntheticSequence(160, 4)
    out["seq_n_kp"], out["seq_n_frames"] = np.array(160), np.array(4)
    frame_state = []
    for f, fr in enumerate(seq.frames):
        matches = twin.knn_match(fr.descL, fr.descR, 1)
        x1_, y1_, x2_, y2_, im = twin.coords_from_kps(matches, fr.kpL, fr.kpR)
        mapping = twin.rectified_stereo_pattern(y1_, y2_, im, thresh=2)
        pts, d3 = twin.get_3d_points_cloud(mapping, k, m1, m2, fr.kpL, fr.kpR)
        frame_state.append((twin.array_to_dict(mapping), d3))
        out[f"seq_map{f}"] = mapping.astype(np.int32)
        out[f"seq_pt_keys{f}"] = np.array(list(d3.keys()), dtype=np.int32)
        out[f"seq_pts{f}"] = np.array(list(d3.values())).reshape(-1, 3)
    for p in range(3):
        f0, f1 = seq.frames[p], seq.frames[p + 1]
        consecutive = twin.apply_thresholds(twin.knn_match(f0.descL, f1.descL, 2), 1)
        cons, _ = twin.matches_in_all_4_images(consecutive, frame_state[p][0], frame_state[p + 1][0],
                                               frame_state[p][1])
        out[f"seq_cons{p}"] = np.array([c[:4] for c in cons], dtype=np.int32).reshape(-1, 4)
        # (v) supporters of seeded poses under the reference's numpy arithmetic
        Rt = seeded_hypotheses(seq.relative_pose(p, p + 1), 24, seed=6 + p)
        masks = np.zeros((24, len(cons)), dtype=bool)
        for h in range(24):
            good, _ = twin.find_supporters(Rt[h], m2, cons, k, f1.kpL, f1.kpR, thresh=2)
            masks[h] = good.astype(bool)
        out[f"seq_Rt{p}"], out[f"seq_masks{p}"] = Rt, masks

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frontend_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
