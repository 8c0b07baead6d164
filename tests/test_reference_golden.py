"""CPU tests against tests/golden/reference_golden.npz -- outputs of the REFERENCE'S OWN function
bodies (utils/utils.py, models/Matcher.py, models/LoopClosure.py), executed in the build container
by tests/golden/make_reference_golden.py.  Held against it here: the numpy restatement
(oracle/numpy_twin.py), the C oracle (oracle/slam_oracle.c) and the host-side mirrors in slam_amd
that need no GPU.  The k-NN lists (cv2's knnMatch) are inputs of the fixture, not pinned by it."""
import os

import numpy as np
import pytest

from oracle import cpu as oracle
from oracle import numpy_twin as twin

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_golden.npz"))
F = int(G["in_n_frames"])
P = F - 1
N = int(G["in_n_kp"])
K, M1, M2 = G["cam_k"], G["cam_m1"], G["cam_m2"]


def rel_err(got, ref):
    return (np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)).max() if len(ref) else 0.0


def stereo_lists(f):
    return [[twin.DMatchLike(i, t, d)] for i, (t, d) in enumerate(zip(G[f"in_st_idx{f}"], G[f"in_st_dist{f}"]))]


def temporal_lists(idx, dist):
    return [[twin.DMatchLike(i, idx[i, c], dist[i, c]) for c in range(2)] for i in range(len(idx))]


def kept(lists):
    return np.array([(m[0].queryIdx, m[0].trainIdx) for m in lists], dtype=np.int32).reshape(-1, 2)


# --------------------------------------------------------------------------- inputs
def test_fixture_inputs_are_the_oracles_knn():
    """The k-NN lists the reference's code was fed are what the C oracle computes (a1 itself is
    restated from cv2's published behaviour: parity unpinned)."""
    for f in range(F):
        idx, dist = oracle.hamming_knn(G["in_descL"][f], G["in_descR"][f], 1)
        assert np.array_equal(idx[:, 0], G[f"in_st_idx{f}"]) and np.array_equal(dist[:, 0], G[f"in_st_dist{f}"])
    for p in range(P):
        idx, dist = oracle.hamming_knn(G["in_descL"][p], G["in_descL"][p + 1], 2)
        assert np.array_equal(idx, G[f"in_tm_idx{p}"]) and np.array_equal(dist, G[f"in_tm_dist{p}"])


def test_cameras_are_the_reference_read_cameras(tmp_path):
    """K, K^-1 P0, K^-1 P1 as utils/utils.py:75-86 derives them from the reference's calib.txt."""
    from slam_amd import frontend
    from slam_amd.synthetic import read_cameras
    for k, m1, m2 in (read_cameras(), frontend.read_cameras()):
        assert np.array_equal(k, K)
        assert np.allclose(m1, M1, rtol=0, atol=2e-16) and np.allclose(m2, M2, rtol=0, atol=2e-16)
    calib = tmp_path / "calib.txt"
    rows = [[707.0912, 0, 601.8873, 0, 0, 707.0912, 183.1104, 0, 0, 0, 1, 0],
            [707.0912, 0, 601.8873, -379.8145, 0, 707.0912, 183.1104, 0, 0, 0, 1, 0]]
    calib.write_text("".join(f"P{i}: " + " ".join(f"{v:.12e}" for v in r) + "\n" for i, r in enumerate(rows)))
    k, m1, m2 = frontend.read_cameras(str(calib))
    assert np.array_equal(k, K) and np.allclose(m1, M1, rtol=0, atol=2e-16) and np.allclose(m2, M2, rtol=0, atol=2e-16)
    assert abs(M2[0, 3] + 0.53715) < 1e-5 and np.allclose(M1[:, :3], np.eye(3), rtol=0, atol=1e-15)   # K^-1 K: not exactly I


# --------------------------------------------------------------------------- the numpy restatement
def test_twin_stereo_stage_equals_the_reference():
    for f in range(F):
        x1, y1, x2, y2, im = twin.coords_from_kps(stereo_lists(f), G["in_kpL"][f], G["in_kpR"][f])
        for got, name in ((x1, "st_x1_"), (y1, "st_y1_"), (x2, "st_x2_"), (y2, "st_y2_")):
            assert np.array_equal(got, G[f"{name}{f}"])
        assert np.array_equal(im, G[f"st_all{f}"])
        for thr in (1, 2):
            assert np.array_equal(twin.rectified_stereo_pattern(y1, y2, im, thresh=thr), G[f"st{thr}_map{f}"])
        mapping = G[f"st2_map{f}"]
        pts, d3 = twin.get_3d_points_cloud(mapping, K, M1, M2, G["in_kpL"][f], G["in_kpR"][f])
        assert np.array_equal(np.array(list(d3.keys())), G[f"pc_keys{f}"])
        assert rel_err(np.array(list(d3.values())).reshape(-1, 3), G[f"pc_pts{f}"]) < 1e-9
        d = twin.array_to_dict(mapping)
        assert np.array_equal(list(d.keys()), G[f"map_keys{f}"]) and np.array_equal(list(d.values()), G[f"map_vals{f}"])


def test_twin_ratio_join_and_supporters_equal_the_reference():
    for p in range(P):
        lists = temporal_lists(G[f"in_tm_idx{p}"], G[f"in_tm_dist{p}"])
        consecutive = twin.apply_thresholds(lists, 1)
        assert np.array_equal(kept(consecutive), G[f"tm_kept{p}"])
        pm = dict(zip(G[f"map_keys{p}"].tolist(), G[f"map_vals{p}"].tolist()))
        cm = dict(zip(G[f"map_keys{p + 1}"].tolist(), G[f"map_vals{p + 1}"].tolist()))
        d3 = dict(zip(G[f"pc_keys{p}"].tolist(), G[f"pc_pts{p}"]))
        cons, filtered = twin.matches_in_all_4_images(consecutive, pm, cm, d3)
        assert np.array_equal(np.array([c[:4] for c in cons]).reshape(-1, 4), G[f"cons_rows{p}"])
        assert np.array_equal(np.array([c[4] for c in cons]).reshape(-1, 3), G[f"cons_X{p}"])
        assert np.array_equal(kept(filtered), G[f"cons_filtered{p}"])
        for h in range(0, int(G["in_n_hyp"]), 5):
            good, num = twin.find_supporters(G[f"sup_Rt{p}"][h], M2, cons, K, G["in_kpL"][p + 1], G["in_kpR"][p + 1], 2)
            assert np.array_equal(good.astype(bool), G[f"sup_masks{p}"][h]) and num == G[f"sup_counts{p}"][h]
    for p in (0, 3):
        idx, dist = oracle.hamming_knn(G["in_descL"][p], G["in_descL"][p + 1], 2)
        for thr in (0.2, 0.6):
            assert np.array_equal(kept(twin.apply_thresholds(temporal_lists(idx, dist), thr)),
                                  G[f"ratio{int(thr * 10)}_kept{p}"])
    idx, dist = oracle.hamming_knn(G["in_descL"][1], G["in_descR"][1], 2)
    assert np.array_equal(kept(twin.apply_thresholds(temporal_lists(idx, dist), 0.6)), G["sig_kept1"])


def test_twin_iteration_count_dlt_and_knife_edge_equal_the_reference():
    for name, p in (("iters_p999", 0.999), ("iters_p99", 0.99)):
        got = [twin.ransac_num_of_iterations(float(e), p) for e in G["iters_eps"]]
        assert np.array_equal(got, G[name])
    assert twin.ransac_num_of_iterations() == int(G["iters_default"])
    Pm, Qm = K @ M1, K @ M2
    got = np.array([twin.least_squares((float(a), float(b)), (float(c), float(d)), Pm, Qm)
                    for a, b, c, d in zip(G["dlt_x1"], G["dlt_y1"], G["dlt_x2"], G["dlt_y2"])])
    assert rel_err(got, G["dlt_X"]) < 1e-9
    n = len(G["knife_X"])
    cons = [(i, i, i, i, G["knife_X"][i]) for i in range(n)]
    good, num = twin.find_supporters(G["knife_Rt"], M2, cons, K, G["knife_uvL"], G["knife_uvR"], 2)
    assert np.array_equal(good.astype(bool), G["knife_mask"]) and num == int(G["knife_count"])


# --------------------------------------------------------------------------- the C oracle
def test_c_oracle_equals_the_reference():
    Pm, Qm = K @ M1, K @ M2
    state = []
    for f in range(F):
        kpL, kpR = G["in_kpL"][f], G["in_kpR"][f]
        idx = G[f"in_st_idx{f}"][:, None].astype(np.int32)
        for thr in (1, 2):
            q = np.flatnonzero(oracle.row_filter(idx, kpL[:, 1], kpR[:, 1], float(thr)))
            assert np.array_equal(np.stack([q, idx[q, 0]]), G[f"st{thr}_map{f}"])
        X, Y, Z, valid = oracle.triangulate_dlt(Pm, Qm, kpL[q, 0], kpL[q, 1], kpR[idx[q, 0], 0], kpR[idx[q, 0], 1])
        assert np.array_equal(q[valid > 0], G[f"pc_keys{f}"])
        assert rel_err(np.stack([X, Y, Z], axis=1)[valid > 0], G[f"pc_pts{f}"]) < 1e-4          # the contract
        smap = np.full(N, -1, np.int32)
        smap[q] = idx[q, 0]
        v3 = np.zeros(N, np.uint8)
        v3[q[valid > 0]] = 1
        state.append((smap, v3))
    for p in range(P):
        tidx, tdist = G[f"in_tm_idx{p}"], G[f"in_tm_dist{p}"]
        tkeep = oracle.ratio_filter(tidx, tdist, 1.0)
        assert np.array_equal(np.flatnonzero(tkeep), G[f"tm_kept{p}"][:, 0])
        rows = oracle.join(tidx, tkeep, state[p][0], state[p][1], state[p + 1][0])
        assert np.array_equal(rows, G[f"cons_rows{p}"])
        Xp = G[f"cons_X{p}"]                                    # the reference's own triangulated points
        kpL, kpR = G["in_kpL"][p + 1], G["in_kpR"][p + 1]
        counts, masks = oracle.score_hypotheses(G[f"sup_Rt{p}"], K, M2[:, 3], Xp[:, 0], Xp[:, 1], Xp[:, 2],
                                                kpL[rows[:, 2], 0], kpL[rows[:, 2], 1], kpR[rows[:, 3], 0],
                                                kpR[rows[:, 3], 1], 2.0, want_masks=True)
        gm = G[f"sup_masks{p}"]
        bits = np.unpackbits(masks.view(np.uint8), axis=1, bitorder="little")[:, :gm.shape[1]].astype(bool)
        assert np.array_equal(bits, gm) and np.array_equal(counts, G[f"sup_counts{p}"])
    for p in (0, 3):
        idx, dist = oracle.hamming_knn(G["in_descL"][p], G["in_descL"][p + 1], 2)
        for thr in (0.2, 0.6):
            assert np.array_equal(np.flatnonzero(oracle.ratio_filter(idx, dist, thr)),
                                  G[f"ratio{int(thr * 10)}_kept{p}"][:, 0])
    X, Y, Z, _ = oracle.triangulate_dlt(Pm, Qm, G["dlt_x1"], G["dlt_y1"], G["dlt_x2"], G["dlt_y2"])
    rel = np.abs(np.stack([X, Y, Z], axis=1) - G["dlt_X"]).max(axis=1) / np.abs(G["dlt_X"]).max(axis=1)
    assert rel.max() < 1e-4 and np.median(rel) < 1e-10
    kx = G["knife_X"]
    counts, masks = oracle.score_hypotheses(G["knife_Rt"][None], K, M2[:, 3], kx[:, 0], kx[:, 1], kx[:, 2],
                                            G["knife_uvL"][:, 0], G["knife_uvL"][:, 1], G["knife_uvR"][:, 0],
                                            G["knife_uvR"][:, 1], 2.0, want_masks=True)
    bits = np.unpackbits(masks.view(np.uint8), axis=1, bitorder="little")[0, :len(kx)].astype(bool)
    assert np.array_equal(bits, G["knife_mask"]) and counts[0] == int(G["knife_count"])


# --------------------------------------------------------------------------- host mirrors (no GPU needed)
def test_host_mirrors_equal_the_reference():
    from slam_amd import frontend
    from slam_amd.constants import LEFT, RIGHT
    from slam_amd.matcher import Matcher
    from slam_amd.types import DMatch, keypoints_from_array
    for f in (0, 4):
        lists = [[DMatch(i, t, d)] for i, (t, d) in enumerate(zip(G[f"in_st_idx{f}"], G[f"in_st_dist{f}"]))]
        x1, y1, x2, y2, im = frontend.coords_from_kps(lists, keypoints_from_array(G["in_kpL"][f]),
                                                      keypoints_from_array(G["in_kpR"][f]))
        assert all(np.array_equal(a, G[f"{n}{f}"]) for a, n in ((x1, "st_x1_"), (y1, "st_y1_"), (x2, "st_x2_"),
                                                                (y2, "st_y2_"), (im, "st_all")))
        d = frontend.array_to_dict(G[f"st2_map{f}"])
        assert np.array_equal(list(d.keys()), G[f"map_keys{f}"]) and np.array_equal(list(d.values()), G[f"map_vals{f}"])
    for name, p_ in (("iters_p999", 0.999), ("iters_p99", 0.99)):
        assert np.array_equal([frontend.ransac_num_of_iterations(float(e), p_) for e in G["iters_eps"]], G[name])
    # ratio test on match lists, as a machine without a GPU evaluates it (models/Matcher.py:178-184)
    for p in (0, 3):
        idx, dist = oracle.hamming_knn(G["in_descL"][p], G["in_descL"][p + 1], 2)
        lists = [[DMatch(i, idx[i, c], dist[i, c]) for c in range(2)] for i in range(len(idx))]
        import torch
        if not torch.cuda.is_available():
            for thr in (0.2, 0.6):
                assert np.array_equal(kept(Matcher.apply_thresholds(lists, thr)), G[f"ratio{int(thr * 10)}_kept{p}"])
    # the joins over dicts, and the loop-closure twin with its (xl, xr, yl) track pairs
    for n, (ref, cand) in enumerate(G["lc_pairs"].tolist()):
        m = Matcher(algo=None, matcher=object())
        for f in (ref, cand):
            m.set_features(f, keypoints_from_array(G["in_kpL"][f]), G["in_descL"][f],
                           keypoints_from_array(G["in_kpR"][f]), G["in_descR"][f])
        consecutive = [[DMatch(q, t, 0)] for q, t in G[f"lc_tm_kept{n}"].tolist()]
        pm = dict(zip(G[f"map_keys{ref}"].tolist(), G[f"map_vals{ref}"].tolist()))
        cm = dict(zip(G[f"map_keys{cand}"].tolist(), G[f"map_vals{cand}"].tolist()))
        d3 = dict(zip(G[f"pc_keys{ref}"].tolist(), G[f"pc_pts{ref}"]))
        cons, tracks, filtered = frontend.find_4_images_matches(consecutive, pm, cm, d3, m, ref, cand)
        assert np.array_equal(np.array([c[:4] for c in cons]).reshape(-1, 4), G[f"lc_rows{n}"])
        assert np.array_equal(np.array([c[4] for c in cons]).reshape(-1, 3), G[f"lc_X{n}"])
        assert np.array_equal(np.array(tracks, dtype=np.float64).reshape(-1, 2, 3), G[f"lc_tracks{n}"])
        assert np.array_equal(kept(filtered), G[f"lc_filtered{n}"])
        assert m.get_feature_location_frame(ref, 3, LEFT) == tuple(float(v) for v in G["in_kpL"][ref][3])
        assert m.get_feature_location_frame(cand, 5, RIGHT) == tuple(float(v) for v in G["in_kpR"][cand][5])
