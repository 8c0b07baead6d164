"""CPU tests: the C-ABI library loads and exports everything include/slam_amd.h
declares (no compute calls without a GPU), argument validation that never reaches a
launch, and the host-side logic (track store, pose chaining, solvers, generator)."""
import ctypes
import os
import pickle
import re

import numpy as np
import pytest

from slam_amd import _lib, pnp, sharding
from slam_amd.synthetic import SyntheticSequence, read_cameras, rodrigues, seeded_hypotheses
from slam_amd.track_database import TrackDatabase
from slam_amd.types import DMatch, KeyPoint, keypoints_from_array, keypoints_to_array

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "slam_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slam_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared_symbols()
    assert len(names) >= 12
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/slam_amd.h but not exported"
    assert sorted(_lib.EXPORTS) == names, "slam_amd/_lib.py must bind exactly the declared symbols"
    assert _lib.load().slam_version() >= 1


def test_argument_validation_needs_no_gpu():
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    # bad stride
    st = lib.slam_hamming_match(null, null, null, 1, 10, 10, 61, 61, 2, 0, 0.0, null, null, null, null, null, null)
    assert st == -3 and b"stride_bytes" in lib.slam_last_error()
    # ratio test needs k = 2
    st = lib.slam_hamming_match(null, null, null, 1, 10, 10, 61, 64, 1, 1, 0.6, null, null, null, null, null, null)
    assert st == -1 and b"k = 2" in lib.slam_last_error()
    # too many train rows for the packed key
    st = lib.slam_hamming_match(null, null, null, 1, 10, (1 << 22) + 1, 61, 64, 2, 0, 0.0, null, null, null, null,
                                null, null)
    assert st == -3
    # null buffers are refused before any launch
    st = lib.slam_hamming_match(null, null, null, 1, 10, 10, 61, 64, 2, 0, 0.0, null, null, null, null, null, null)
    assert st == -1 and b"null" in lib.slam_last_error()
    assert lib.slam_triangulate_dlt(null, null, null, null, null, null, 5, null, null, null, null, null) == -1
    assert lib.slam_score_hypotheses(null, 1, 1, null, null, null, null, null, null, null, null, null, 4, null, 4,
                                     2.0, null, null, 0, null, null) == -1
    # the fused entry points of the batched path
    K9 = (ctypes.c_double * 9)(700.0, 0.0, 600.0, 0.0, 700.0, 180.0, 0.0, 0.0, 1.0)
    t3 = (ctypes.c_double * 3)(-0.5, 0.0, 0.0)
    assert lib.slam_winner_tables(*([null, 3, null, null, 16, K9, t3, 2.0] + [null] * 16)) == -1
    assert b"null" in lib.slam_last_error()
    assert lib.slam_winner_tables(*([null, -1, null, null, 16, K9, t3, 2.0] + [null] * 16)) == -1
    assert lib.slam_winner_tables(*([null, 0, null, null, 16, K9, t3, 2.0] + [null] * 16)) == 0
    refine = [K9, t3, 2.0] + [null] * 7 + [16, null, 16, null, 2, 30, null, null, null, null]
    assert lib.slam_refine_winners(*refine) == -1 and b"null" in lib.slam_last_error()
    refine[14] = 0                                              # no sets: nothing to do
    assert lib.slam_refine_winners(*refine) == 0
    bad_K = (ctypes.c_double * 9)(700.0, 0.1, 600.0, 0.0, 700.0, 180.0, 0.0, 0.0, 1.0)
    assert lib.slam_refine_poses(bad_K, null, null, null, null, null, 16, null, 16, ctypes.c_void_p(4096), 1, null, 2,
                                 30, null, null, null) == -1 and b"pinhole" in lib.slam_last_error()
    join = [null, 2, 100] + [null] * 11 + [16] + [null] * 9
    assert lib.slam_join_consensus_prepared(*(join + [null, 2.0, ctypes.c_void_p(4096), null])) == -1
    assert b"intrinsics" in lib.slam_last_error()               # a workspace without the camera
    assert lib.slam_join_consensus_prepared(*(join + [K9, 2.0, ctypes.c_void_p(4104), null])) == -1
    assert b"aligned" in lib.slam_last_error()
    assert lib.slam_join_consensus_prepared(*(join + [K9, 2.0, null, null])) == -1     # null tables
    assert lib.slam_score_hypotheses_range(null, 1, 8, 4, 8, null, K9, t3, *([null] * 7), 4, null, 4, 2.0, null, null,
                                           0, null, 0, null) == -1
    assert b"exceeds" in lib.slam_last_error()
    # empty work is a no-op success
    assert lib.slam_hamming_match(null, null, null, 0, 0, 0, 61, 64, 2, 0, 0.0, null, null, null, null, null,
                                  null) == 0
    assert lib.slam_pack_descriptors(null, 0, 61, 61, null, 64, null) == 0
    assert lib.slam_pack_descriptors(null, 5, 61, 61, null, 48, null) == -1


def test_ops_refuse_cpu_tensors():
    import torch
    from slam_amd import ops
    with pytest.raises(_lib.SlamError):
        ops.pack_descriptors(torch.zeros((4, 61), dtype=torch.uint8), 61)


# ------------------------------------------------------------------ track store
class _FakeMatcher:
    def __init__(self, kp):
        self.kp = kp

    def get_feature_location_frame(self, frameId, kp, loc):
        return self.kp[(frameId, loc)][kp]


def test_track_database_continuation_and_schema(tmp_path):
    kp = {(f, s): {i: (float(10 * f + i), float(100 * s + i)) for i in range(8)} for f in range(4) for s in (0, 1)}
    m = _FakeMatcher(kp)
    db = TrackDatabase()
    X = np.zeros(3)
    db.update_tracks_from_frame(m, 0, [(0, 1, 2, 3, X), (4, 5, 6, 7, X)])
    assert db.get_num_tracks() == 2 and db.get_track_ids_for_frame(0) == [0, 1]
    # frame-1 left keypoint 2 continues track 0; keypoint 5 starts a new one; keypoint 6's track ends
    db.update_tracks_from_frame(m, 1, [(2, 0, 1, 1, X), (5, 5, 5, 5, X)])
    assert db.get_frame_ids_for_track(0) == [0, 1, 2] and db.get_frame_ids_for_track(1) == [0, 1]
    assert db.get_frame_ids_for_track(2) == [1, 2]
    # parity table: a keypoint index seen two pairs ago must not resurrect its track
    db.update_tracks_from_frame(m, 2, [(6, 0, 0, 0, X), (1, 1, 3, 3, X)])
    assert db.get_frame_ids_for_track(0) == [0, 1, 2, 3]
    assert db.get_frame_ids_for_track(3) == [2, 3] and db.get_num_tracks() == 4
    assert db.get_feature_locations(0, 0) == (kp[(0, 0)][0][0], kp[(0, 1)][1][0], kp[(0, 0)][0][1])
    kp_idx, (xl, xr, yl), fid = db.get_track_data(0)[0]
    assert (kp_idx, fid) == (0, 0) and (xl, yl) == kp[(0, 0)][0] and xr == kp[(0, 1)][1][0]
    assert db.get_max_track() == 4 and db.get_min_track() == 2 and db.get_num_frames() == 4
    assert abs(db.get_mean_track_length() - (4 + 2 + 2 + 2) / 4) < 1e-12
    db.add_inliers_ratio(0, 0.9)
    db.add_num_matches(0, 2.0)
    with pytest.raises(AssertionError):
        db.add_inliers_ratio(17, 0.5)
    # table entry point builds the same store
    db2 = TrackDatabase()
    for f, rows in ((0, [(0, 1, 2, 3), (4, 5, 6, 7)]), (1, [(2, 0, 1, 1), (5, 5, 5, 5)]), (2, [(6, 0, 0, 0), (1, 1, 3, 3)])):
        rows = np.array(rows)
        loc = np.array([[kp[(f, 0)][r[0]][0], kp[(f, 1)][r[1]][0], kp[(f, 0)][r[0]][1],
                         kp[(f + 1, 0)][r[2]][0], kp[(f + 1, 1)][r[3]][0], kp[(f + 1, 0)][r[2]][1]] for r in rows],
                       dtype=np.float32)
        db2.update_tracks_from_tables(f, rows, loc)
    db2.add_inliers_ratio(0, 0.9)
    db2.add_num_matches(0, 2.0)
    db2.get_mean_track_length()   # the getter caches its result in the stored state
    assert db == db2
    # pickle schema: the reference's key names (models/TrackDatabase.py:121-137)
    path = str(tmp_path / "tracks.pkl")
    db.serialize(path)
    data = pickle.load(open(path, "rb"))
    assert set(data) == {"tracks", "frame_ids", "next_free_trackId", "last_insertions", "num_tracks", "num_frames",
                         "mean_track_length", "max_length", "min_length", "mean_frame_links",
                         "frame_id_to_inliers_ratio", "ex_camera_positions", "ex_matrices",
                         "frame_id_to_num_matches"}
    db3 = TrackDatabase(path)
    assert db3 == db
    assert TrackDatabase().deserialize(str(tmp_path / "missing.pkl")) == 1
    x, y = db.calculate_connectivity_data()
    assert list(y) == [2, 2, 2, 0]


def test_pose_chaining_matches_reference_formula():
    rng = np.random.default_rng(0)
    rel = np.stack([np.hstack([rodrigues(rng.normal(size=3) * 0.05), rng.normal(size=(3, 1))]) for _ in range(5)])
    ex, cam = sharding.chain_poses(rel)
    R, t = np.eye(3), np.zeros(3)
    for i in range(5):
        R, t = rel[i][:, :3] @ R, rel[i][:, :3] @ t + rel[i][:, 3]
        assert np.allclose(ex[i + 1], np.hstack([R, t[:, None]]))
        assert np.allclose(cam[i + 1], -R.T @ t)


def test_every_route_chains_poses_bit_for_bit_like_numpy():
    """chain_poses is the one place poses are chained (host replay, device track arrays, batched VO).  It
    runs in the C helper only where that reproduces numpy's `@` products -- the reference's arithmetic
    (utils/utils.py:283-288) -- bit for bit; over a 4541-frame sequence any other summation order shows."""
    rng = np.random.default_rng(5)
    P = 4540
    rel = np.empty((P, 3, 4))
    rel[:, :, :3] = np.stack([rodrigues(v) for v in rng.normal(size=(P, 3)) * 0.01])
    rel[:, :, 3] = rng.normal(size=(P, 3)) * 0.1 + [0, 0, -0.8]
    k, m1, m2 = read_cameras()
    ex, cam = sharding.chain_poses(rel, first=m1)
    ex_np, cam_np = sharding._chain_poses_numpy(rel, first=m1)
    assert np.array_equal(ex, ex_np) and np.array_equal(cam, cam_np)
    if sharding.native_chain_matches_numpy():
        ex_c, cam_c = sharding._chain_poses_native(rel, first=m1)
        assert np.array_equal(ex_c, ex_np) and np.array_equal(cam_c, cam_np)


def test_replayed_frame_records_have_the_reference_types():
    """num_matches is np.sum of find_supporters' float64 0/1 array in the reference (utils/utils.py:279,282,390)."""
    import torch
    sh = dict(n_inl=torch.tensor([2, 0], dtype=torch.int32), n_cons=torch.tensor([3, 0], dtype=torch.int32),
              pose=torch.eye(3, 4, dtype=torch.float64).reshape(1, 12).repeat(2, 1),
              rows=torch.tensor([[0, 1, 2, 3], [4, 5, 6, 7]], dtype=torch.int32), loc=torch.zeros((2, 6)))
    db = sharding.replay_tracks([sh])
    nm = db.get_num_matches_per_frame()
    assert [type(v) for v in nm.values()] == [np.float64, np.float64] and list(nm.values()) == [2.0, 0.0]
    assert db.get_inliers_ratio_per_frame() == {0: 2 / 3, 1: 0.0}


def test_shard_pairs_cover_the_sequence():
    for n_frames, world in ((4541, 8), (2560, 3), (5, 8), (1, 2), (2, 2)):
        sh = sharding.shard_pairs(n_frames, world)
        assert len(sh) == world and sh[0][0] == 0 and sh[-1][1] == max(n_frames - 1, 0)
        assert all(sh[i][1] == sh[i + 1][0] for i in range(world - 1))
        sizes = [e - s for s, e in sh]
        assert max(sizes) - min(sizes) <= 1


def test_p3p_and_refinement_recover_known_poses():
    k, m1, m2 = read_cameras()
    rng = np.random.default_rng(3)
    for _ in range(50):
        R, t = rodrigues(rng.normal(size=3) * 0.1), rng.normal(size=3) * 0.5
        X = np.stack([rng.uniform(-20, 20, 60), rng.uniform(-3, 2, 60), rng.uniform(5, 80, 60)], axis=1)
        Y = X @ R.T + t
        uv = (k @ Y.T).T
        uv = uv[:, :2] / uv[:, 2:]
        Rt = pnp.p3p_4pt(X[:4], uv[:4], k)
        d3 = [np.linalg.norm(X[i] - X[j]) for i in range(3) for j in range(i)]
        if min(d3) > 0.1 * max(d3):       # a thin triangle may be refused (None), like a failed cv2 solve
            assert Rt is not None
        if Rt is not None:
            proj = (k @ (X @ Rt[:, :3].T + Rt[:, 3]).T).T
            assert np.abs(proj[:, :2] / proj[:, 2:] - uv).max() < 1e-3
        noisy = uv + rng.normal(0, 0.3, uv.shape)
        start = np.hstack([rodrigues(np.array([0.01, -0.01, 0.02])) @ R, (t + 0.05)[:, None]])
        ref = pnp.refine_pnp(X, noisy, k, start)
        assert np.abs(ref - np.hstack([R, t[:, None]])).max() < 0.02
    assert pnp.p3p_4pt(X[:3], uv[:3], k) is None                      # needs four points
    assert pnp.p3p_4pt(np.repeat(X[:1], 4, axis=0), uv[:4], k) is None  # degenerate sample


def test_host_refinement_finds_the_minimum_an_independent_solver_finds():
    """pnp.refine_pnp stands in for cv2.solvePnP(SOLVEPNP_ITERATIVE) (utils/utils.py:452), which cannot run
    here.  What can be pinned without it: the minimum of that cost is a property of the data, so a
    Levenberg-Marquardt that shares no code with this repository (MINPACK through scipy) must arrive at the same
    pose from the same start -- with the left image's residuals alone and with both images'."""
    from helpers import scipy_pnp_minimum
    k, m1, m2 = read_cameras()
    t_right = m2[:, 3]
    rng = np.random.default_rng(12)
    for trial in range(12):
        n = int(rng.integers(6, 400))
        R, t = rodrigues(rng.normal(size=3) * 0.05), rng.normal(size=3) * 0.5
        X = np.stack([rng.uniform(-20, 20, n), rng.uniform(-3, 2, n), np.exp(rng.uniform(np.log(4), np.log(80), n))], axis=1)
        Y = X @ R.T + t
        uvL = np.stack([k[0, 0] * Y[:, 0] / Y[:, 2] + k[0, 2], k[1, 1] * Y[:, 1] / Y[:, 2] + k[1, 2]], axis=1)
        Yr = Y + t_right
        uvR = np.stack([k[0, 0] * Yr[:, 0] / Yr[:, 2] + k[0, 2], k[1, 1] * Yr[:, 1] / Yr[:, 2] + k[1, 2]], axis=1)
        uvL, uvR = uvL + rng.normal(0, 0.4, uvL.shape), uvR + rng.normal(0, 0.4, uvR.shape)
        start = np.hstack([rodrigues(rng.normal(size=3) * 5e-3) @ R, (t + rng.normal(size=3) * 0.05)[:, None]])
        for right in (False, True):
            kw = dict(points_2d_right=uvR, t_right=t_right) if right else {}
            mine = pnp.refine_pnp(X, uvL, k, start, **kw)
            theirs, cost = scipy_pnp_minimum(X, uvL, k, start, uvR if right else None, t_right if right else None)
            assert np.abs(mine - theirs).max() < 2e-7, (trial, right)
            Ym = X @ mine[:, :3].T + mine[:, 3]
            r = [k[0, 0] * Ym[:, 0] / Ym[:, 2] + k[0, 2] - uvL[:, 0], k[1, 1] * Ym[:, 1] / Ym[:, 2] + k[1, 2] - uvL[:, 1]]
            if right:
                Yq = Ym + t_right
                r += [k[0, 0] * Yq[:, 0] / Yq[:, 2] + k[0, 2] - uvR[:, 0], k[1, 1] * Yq[:, 1] / Yq[:, 2] + k[1, 2] - uvR[:, 1]]
            assert abs(float((np.concatenate(r) ** 2).sum()) - cost) <= 1e-9 * cost


def test_p3p_branch_is_the_best_one_a_multistart_search_finds():
    """pnp.p3p_4pt stands in for cv2.solvePnP(flags=SOLVEPNP_AP3P) on four points (utils/utils.py:428,481-492):
    every pose that reprojects the first three points exactly is a candidate, the fourth point chooses.  Checked
    against a search that shares nothing with the quartic solver: scipy's least squares from many random starts
    collects poses with zero residual on the three points and all depths positive; none of them may fit the
    fourth point better than the pose returned."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation
    k, m1, m2 = read_cameras()
    fx, fy, cx, cy = k[0, 0], k[1, 1], k[0, 2], k[1, 2]
    rng = np.random.default_rng(21)

    def project(p, X):
        Y = X @ Rotation.from_rotvec(p[:3]).as_matrix().T + p[3:]
        return np.stack([fx * Y[:, 0] / Y[:, 2] + cx, fy * Y[:, 1] / Y[:, 2] + cy], axis=1), Y[:, 2]

    checked = 0
    for trial in range(12):
        R, t = rodrigues(rng.normal(size=3) * 0.05), np.array([0.05, -0.02, -0.8]) + rng.normal(size=3) * 0.05
        X = np.stack([rng.uniform(-15, 15, 4), rng.uniform(-3, 2, 4), np.exp(rng.uniform(np.log(5), np.log(60), 4))], axis=1)
        Y = X @ R.T + t
        uv = np.stack([fx * Y[:, 0] / Y[:, 2] + cx, fy * Y[:, 1] / Y[:, 2] + cy], axis=1) + rng.normal(0, 0.3, (4, 2))
        mine = pnp.p3p_4pt(X, uv, k)
        if mine is None:
            continue
        Ym = X @ mine[:, :3].T + mine[:, 3]
        pm = np.stack([fx * Ym[:, 0] / Ym[:, 2] + cx, fy * Ym[:, 1] / Ym[:, 2] + cy], axis=1)
        assert np.abs(pm[:3] - uv[:3]).max() < 1e-6
        my_err = float(np.linalg.norm(pm[3] - uv[3]))
        found = []
        truth = np.concatenate([Rotation.from_matrix(R).as_rotvec(), t])
        starts = [truth] + [np.concatenate([rng.normal(size=3) * s, t + rng.normal(size=3) * 10 * s])
                            for s in (0.05, 0.3, 1.0) for _ in range(14)]
        for p0 in starts:
            sol = least_squares(lambda p: (project(p, X[:3])[0] - uv[:3]).ravel(), p0, method="lm", xtol=1e-14, ftol=1e-14,
                                gtol=1e-14, max_nfev=400)
            pr, depth = project(sol.x, X)
            if np.abs(pr[:3] - uv[:3]).max() < 1e-7 and (depth > 0).all():
                found.append(float(np.linalg.norm(pr[3] - uv[3])))
        assert found, trial                                  # the start at the true pose converges at least
        assert my_err <= min(found) + 1e-5, (trial, my_err, sorted(found)[:3])
        checked += 1
    assert checked >= 8


def test_synthetic_sequence_shape_and_reproducibility():
    a = SyntheticSequence(300, 3)
    b = SyntheticSequence(300, 3)
    for fa, fb in zip(a.frames, b.frames):
        assert np.array_equal(fa.descL, fb.descL) and np.array_equal(fa.kpR, fb.kpR)
    f = a.frames[1]
    assert f.kpL.dtype == np.float32 and f.descL.dtype == np.uint8 and f.descL.shape == (300, 61)
    assert (f.descL[:, 60] & 0xC0 == 0).all()                       # 486 used bits
    assert (f.idL >= 0).sum() == 180                                # 60 % true points
    # persistent points reappear in the next frame
    assert len(set(a.frames[0].idL[a.frames[0].idL >= 0]) & set(f.idL[f.idL >= 0])) > 120
    Rt = a.relative_pose(0, 1)
    assert abs(np.linalg.det(Rt[:, :3]) - 1) < 1e-12 and abs(Rt[2, 3] + 0.8) < 0.05
    H = seeded_hypotheses(Rt, 9)
    assert H.shape == (9, 3, 4) and np.array_equal(H, seeded_hypotheses(Rt, 9))


def test_type_shims():
    kp = KeyPoint(1.1, 2.2)
    assert kp.pt == (float(np.float32(1.1)), float(np.float32(2.2)))
    arr = keypoints_to_array(keypoints_from_array(np.array([[1.5, 2.5], [3.25, 4.0]])))
    assert arr.dtype == np.float32 and arr.tolist() == [[1.5, 2.5], [3.25, 4.0]]
    m = DMatch(3, 7, 12)
    assert (m.queryIdx, m.trainIdx, m.imgIdx, m.distance) == (3, 7, 0, 12.0)
