"""GPU tests of the host-side mirror of the reference interface (Matcher, the utils
functions, RANSAC control flow, the visual-odometry loop) against the statement-by-
statement numpy twin and the C oracle."""
import random

import numpy as np
import pytest

from oracle import cpu as oracle
from oracle import numpy_twin as twin
import torch

from slam_amd import frontend, pipeline, pnp, vo
from slam_amd.constants import CONSECUTIVE, FILTERED_MATCHES, LEFT, MATCHES, RIGHT
from slam_amd.matcher import HipBFMatcher, Matcher, PrecomputedFeatures
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seq():
    return SyntheticSequence(350, 6)


def test_hip_bf_matcher_is_a_knnmatch_drop_in(device):
    rng = np.random.default_rng(1)
    q = (rng.integers(0, 256, (40, 32), dtype=np.uint8) & 7).astype(np.uint8)
    t = (rng.integers(0, 256, (55, 32), dtype=np.uint8) & 7).astype(np.uint8)
    m = HipBFMatcher()
    for k in (1, 2):
        got = m.knnMatch(q, t, k=k)
        ref = twin.knn_match(q, t, k)
        assert len(got) == len(ref)
        for g, r in zip(got, ref):
            assert [(x.queryIdx, x.trainIdx, x.distance, x.imgIdx) for x in g] == \
                   [(x.queryIdx, x.trainIdx, x.distance, x.imgIdx) for x in r]
    # fewer train rows than k: shorter lists, like cv2
    got = m.knnMatch(q, t[:1], k=2)
    assert all(len(g) == 1 for g in got)
    with pytest.raises(ValueError):
        Matcher.apply_thresholds(got, 0.6)


def test_matcher_cache_and_stereo_functions(device, seq):
    k, m1, m2 = read_cameras()
    matcher = Matcher(algo=PrecomputedFeatures(seq.frames))
    f0 = seq.frames[0]
    mapping = frontend.find_stereo_matches(matcher, 0)
    # statement-by-statement reference flow on the twin
    ref_matches = twin.knn_match(f0.descL, f0.descR, 1)
    x1, y1, x2, y2, im = twin.coords_from_kps(ref_matches, f0.kpL, f0.kpR)
    ref_mapping = twin.rectified_stereo_pattern(y1, y2, im, thresh=2)
    assert np.array_equal(mapping, ref_mapping)
    # cache layout (models/Matcher.py:118-119,144-145,172)
    entry = matcher.cache[0]
    assert len(entry[FILTERED_MATCHES]) == len(f0.descL)           # unfiltered k=1 list
    assert len(entry[MATCHES]) == mapping.shape[1]                  # row-filtered list
    assert [m[0].queryIdx for m in entry[MATCHES]] == list(mapping[0])
    kpl, kpr = matcher.get_kp(0)
    assert kpl[3].pt == (float(f0.kpL[3, 0]), float(f0.kpL[3, 1]))
    assert matcher.get_feature_location_frame(0, 5, RIGHT) == (float(f0.kpR[5, 0]), float(f0.kpR[5, 1]))
    # the stand-alone filter functions agree with the fused kernel path
    x1g, y1g, x2g, y2g, img = frontend.coords_from_kps(entry[FILTERED_MATCHES], kpl, kpr)
    assert np.array_equal(img, im) and np.array_equal(y1g, y1) and np.array_equal(x2g, x2)
    a, b, c, d, inl = frontend.rectificatied_stereo_pattern(y1g, y2g, img, thresh=2)
    assert np.array_equal(inl, ref_mapping)
    assert len(c) + len(a) == len(f0.descL)

    # triangulation: same keys, same points as np.linalg.svd per point
    pts, d3 = frontend.get_3d_points_cloud(mapping, k, m1, m2, matcher, file_index=0)
    rpts, rd3 = twin.get_3d_points_cloud(ref_mapping, k, m1, m2, f0.kpL, f0.kpR)
    assert list(d3.keys()) == list(rd3.keys()) and len(pts) == len(rpts)
    for key in rd3:
        assert np.abs(d3[key] - rd3[key]).max() / np.abs(rd3[key]).max() < 1e-4
    one = frontend.least_squares(tuple(f0.kpL[mapping[0, 0]]), tuple(f0.kpR[mapping[1, 0]]), k @ m1, k @ m2)
    assert np.allclose(one, twin.least_squares(tuple(map(float, f0.kpL[mapping[0, 0]])),
                                               tuple(map(float, f0.kpR[mapping[1, 0]])), k @ m1, k @ m2), rtol=1e-6)


def test_temporal_match_join_and_supporters(device, seq):
    k, m1, m2 = read_cameras()
    matcher = Matcher(algo=PrecomputedFeatures(seq.frames))
    f0, f1 = seq.frames[0], seq.frames[1]
    map0 = frontend.match_next_pair(0, matcher)
    _, d3 = frontend.get_3d_points_cloud(map0, k, m1, m2, matcher, file_index=0)
    map1 = frontend.match_next_pair(1, matcher)
    for thr in (1, 0.6):
        got = matcher.match_between_consecutive_frames(0, 1, thresh=thr)
        ref = twin.apply_thresholds(twin.knn_match(f0.descL, f1.descL, 2), thr)
        assert [(m[0].queryIdx, m[0].trainIdx, m[0].distance) for m in got] == \
               [(m[0].queryIdx, m[0].trainIdx, m[0].distance) for m in ref]
        assert matcher.cache[1][CONSECUTIVE] is got
        assert [(m[0].queryIdx, m[0].trainIdx) for m in matcher.match_between_any_frames(0, 1, thresh=thr)] == \
               [(m[0].queryIdx, m[0].trainIdx) for m in ref]
    consecutive = matcher.match_between_consecutive_frames(0, 1, thresh=1)
    pm, cm = frontend.array_to_dict(map0), frontend.array_to_dict(map1)
    cons, filt = frontend.get_matches_that_appear_in_all_4_images(consecutive, pm, cm, d3)
    rcons, rfilt = twin.matches_in_all_4_images(consecutive, twin.array_to_dict(map0), twin.array_to_dict(map1),
                                                {int(a): b for a, b in d3.items()})
    assert [c[:4] for c in cons] == [c[:4] for c in rcons] and len(filt) == len(rfilt) and len(cons) > 50
    kp1, kp2 = matcher.get_kp(1)
    Rt = seq.relative_pose(0, 1)
    for pose in (Rt, Rt + 0.002, np.hstack([np.eye(3), np.zeros((3, 1))])):
        good, num = frontend.find_supporters(pose, m2, cons, k, kp1, kp2, thresh=2, debug=False)
        rgood, rnum = twin.find_supporters(pose, m2, cons, k, f1.kpL, f1.kpR, thresh=2)
        assert num == rnum and np.array_equal(good, rgood) and good.dtype == np.float64


def _reference_order_ransac(X, uvL, uvR, k, m2, thresh, max_iterations):
    """ransac_for_pnp's loop exactly as written in the reference (one draw, one solve, one
    score per iteration), with the package's device P3P / refinement called one pose at a time
    and the C oracle as scorer."""
    n = X.shape[0]
    solver = frontend._ScoreSet(X, uvL, uvR, k, m2, thresh)
    tr = m2[:, 3]
    pts = (X[:, 0].copy(), X[:, 1].copy(), X[:, 2].copy(), uvL[:, 0].copy(), uvL[:, 1].copy(), uvR[:, 0].copy(),
           uvR[:, 1].copy())
    score = lambda Rt: oracle.score_hypotheses(Rt[None], k, tr, *pts, thresh, want_masks=True)
    best, best_Rt = 0, None
    eps, I, i = 0.99, max_iterations, 0
    while i < I and eps != 0:
        sel = random.sample(range(n), k=4)
        Rt = solver.solve([sel]).cpu().numpy().reshape(3, 4)
        if not np.isfinite(Rt).all():            # a failed solve: the reference's None
            i += 1
            continue
        num = int(score(Rt)[0][0])
        if num > best:
            best, best_Rt = num, Rt
        eps = min(eps, 1 - (num / n))
        I = min(twin.ransac_num_of_iterations(eps, 0.999), max_iterations)
        i += 1
    if best_Rt is None:
        return None, None, None, i
    c, m = score(best_Rt)
    refined = solver.refine(best_Rt)
    c, m = score(refined)
    mask = np.unpackbits(m[0].view(np.uint8), bitorder="little")[:n].astype(np.float64)
    if int(c[0]) >= best:
        best_Rt = refined
    return best_Rt, int(c[0]) / n, mask, i


def test_ransac_control_flow_and_random_stream(device, seq):
    k, m1, m2 = read_cameras()
    ws, pairs = vo.front_end_tables(seq.frames[:3], (k, m1, m2))
    for p, max_it in ((0, 1000), (1, 40), (0, 3)):
        tab = pairs[p]
        random.seed(1234 + p)
        ref = _reference_order_ransac(tab["X"], tab["uvL"], tab["uvR"], k, m2, 2, max_it)
        ref_state = random.getstate()
        random.seed(1234 + p)
        Rt, ratio, mask = frontend.ransac_on_arrays(tab["X"], tab["uvL"], tab["uvR"], k, m2, thresh=2,
                                                    max_iterations=max_it)
        assert random.getstate() == ref_state, "the global random stream must advance exactly as in the reference"
        assert np.array_equal(Rt, ref[0]) and ratio == ref[1] and np.array_equal(mask, ref[2])
        assert ratio > 0.8
    # populations below four return Nones (utils/utils.py:417-419)
    tab = pairs[0]
    assert frontend.ransac_on_arrays(tab["X"][:3], tab["uvL"][:3], tab["uvR"][:3], k, m2) == (None, None, None)
    assert frontend.ransac_for_pnp([], k, [], [], m2, return_supporters=True) == (None, None, None)
    assert frontend.ransac_for_pnp([], k, [], [], m2) == (None, None)


def test_vo_loop_matches_batched_pipeline_and_ground_truth(device, seq):
    k, m1, m2 = read_cameras()
    frames = seq.frames
    random.seed(6)
    matcher = Matcher(algo=PrecomputedFeatures(frames))
    cam_a, db_a = frontend.track_camera_for_many_images(thresh=1, matcher=matcher, num_of_frames=len(frames),
                                                        cameras=(k, m1, m2))
    state_a = random.getstate()
    random.seed(6)
    cam_b, db_b = vo.track_camera_batched(frames, (k, m1, m2), thresh=1)
    assert random.getstate() == state_a
    assert np.array_equal(cam_a, cam_b)
    assert db_a == db_b
    assert db_a.get_num_frames() == len(frames) and db_a.get_num_tracks() > 100
    assert db_a.get_max_track() == len(frames)          # persistent points are tracked through the whole clip
    assert min(db_a.get_inliers_ratio_per_frame().values()) > 0.8
    # the estimated trajectory follows the synthetic ground truth
    for f in range(1, len(frames)):
        Rt = frames[f].Rt
        gt_centre = -Rt[:, :3].T @ Rt[:, 3]
        assert np.linalg.norm(cam_a[f] - gt_centre) < 0.05 * f + 0.02
    # a track's stored locations are the float32 keypoint coordinates (xl, xr, yl)
    tid, track = next(iter(db_a.get_trackIds().items()))
    kp_idx, (xl, xr, yl), fid = track[0]
    assert (xl, yl) == (float(frames[fid].kpL[kp_idx, 0]), float(frames[fid].kpL[kp_idx, 1]))


def test_loop_closure_consensus_matching(device, seq):
    k, m1, m2 = read_cameras()
    matcher = Matcher(algo=PrecomputedFeatures(seq.frames))
    random.seed(6)
    Rt, tracks, ratio, filt, sup, n_cons = frontend.consensus_matching_of_general_two_frames(0, 3, matcher, thresh=0.6,
                                                                                              cameras=(k, m1, m2))
    assert Rt is not None and n_cons >= 10 and len(sup) == n_cons and len(tracks) == int(np.sum(sup))
    gt = seq.relative_pose(0, 3)
    assert np.abs(Rt - gt).max() < 0.05
    # too few consensus matches -> six Nones (models/LoopClosure.py:129-130)
    tiny = SyntheticSequence(12, 2)
    m2_ = Matcher(algo=PrecomputedFeatures(tiny.frames))
    assert frontend.consensus_matching_of_general_two_frames(0, 1, m2_, cameras=(k, m1, m2)) == (None,) * 6


def test_graph_replay_matches_plain_launches(device):
    """The whole front-end recorded into a HIP graph gives the same tables as plain launches,
    also after the inputs were changed in place."""
    k, m1, m2 = read_cameras()
    seq = SyntheticSequence(600, 5, seed=11)
    P, H = 4, 300
    Rt_host = np.stack([seeded_hypotheses(seq.relative_pose(p, p + 1), H, seed=p) for p in range(P)])
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seq.frames, device)
    ws = fe.prepare(tb, n_hyp=H)
    Rt = torch.from_numpy(Rt_host).to(device)
    graph = pipeline.capture(fe, ws, Rt)
    names = ("s_idx", "s_keep", "t_idx", "t_keep", "counts", "best", "n_inl", "inl_rows", "inl_loc")
    for trial in range(2):
        fe.run(ws, Rt)
        torch.cuda.synchronize()
        plain = {n: getattr(ws, n).clone() for n in names}
        for n in names:
            getattr(ws, n).fill_(0)
        graph.replay()
        torch.cuda.synchronize()
        n_t = int(tb.offL[-2])           # temporal outputs exist for the queries of all but the last frame
        n_i = int(ws.n_inl.sum())        # dense inlier tables are filled up to the total count
        for n in names:
            lim = n_t if n.startswith("t_") else (n_i if n.startswith("inl_") else None)
            assert torch.equal(plain[n][:lim], getattr(ws, n)[:lim]), (trial, n)
        assert n_i > 0
        # second trial: other hypotheses written into the same tensor
        Rt.copy_(torch.from_numpy(np.ascontiguousarray(Rt_host[:, ::-1])).to(device))


def test_graph_replay_of_the_device_ransac(device):
    """run_ransac recorded into a HIP graph: same poses, counts and inlier tables as plain launches, also
    after other frames were written into the same tables, and with the stereo refinement option."""
    k, m1, m2 = read_cameras()
    seqs = [SyntheticSequence(700, 4, seed=s) for s in (5, 9)]
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seqs[0].frames, device)
    names = ("pose", "ratio", "found", "iters", "best_count", "n_inl", "inl_rows", "inl_loc", "counts")
    for kw in ({}, {"stereo_refine": True}):
        ws = fe.prepare(tb, n_hyp=256)
        graph = pipeline.capture_ransac(fe, ws, 6, **kw)
        for seq in seqs:
            other = pipeline.FrameTables.from_frames(seq.frames, device)
            for name in ("descL", "descR", "xL", "yL", "xR", "yR"):
                getattr(tb, name).copy_(getattr(other, name))
            fe.run_ransac(ws, 6, **kw)
            torch.cuda.synchronize()
            plain = {n: getattr(ws, n).clone() for n in names}
            for n in names:
                getattr(ws, n).fill_(0)
            graph.replay()
            torch.cuda.synchronize()
            n_i = int(ws.n_inl.sum())
            assert n_i > 0 and bool(ws.found[:ws.P].all())
            first = 128                                  # counts beyond the first slice exist only for pairs still running
            for n in names:
                a, b = plain[n], getattr(ws, n)
                if n.startswith("inl_"):
                    a, b = a[:n_i], b[:n_i]
                if n == "counts":
                    a, b = a[:, :first], b[:, :first]
                assert torch.equal(a, b), (kw, n)
            gt = np.stack([seq.relative_pose(p, p + 1) for p in range(ws.P)])
            assert np.abs(ws.pose.cpu().numpy()[:ws.P].reshape(-1, 3, 4) - gt).max() < 0.05


def test_winner_tables_match_the_mask_path(device):
    """slam_winner_tables (flags of the winner recomputed beside the compaction) against the
    longer route through the C ABI: one-hypothesis scoring with a supporter mask, prefix sum,
    slam_build_inlier_tables.  Ragged pairs, one of them empty."""
    k, m1, m2 = read_cameras()
    seq = SyntheticSequence(500, 6, seed=23)
    f = seq.frames[3]                                    # pairs 2 and 3 have no consensus matches
    seq.frames[3] = type(f)(f.kpL[:0], f.kpR[:0], f.descL[:0], f.descR[:0], f.idL[:0], f.idR[:0], f.Rt)
    P, H = 5, 200
    Rt_host = np.stack([seeded_hypotheses(seq.relative_pose(p, p + 1), H, seed=40 + p) for p in range(P)])
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seq.frames, device)
    ws = fe.prepare(tb, n_hyp=H)
    Rt = torch.from_numpy(Rt_host).to(device)
    fe.run(ws, Rt)
    torch.cuda.synchronize()
    fused = {n: getattr(ws, n).clone() for n in ("n_inl", "inl_rows", "inl_loc")}
    total = int(fused["n_inl"].sum())
    assert total > 0 and int(fused["n_inl"][2]) == 0 and int(fused["n_inl"][3]) == 0
    assert torch.equal(fused["n_inl"].view(-1)[:P], ws.best_count.view(-1)[:P])
    for n in fused:
        getattr(ws, n).fill_(0)
    stream = torch.cuda.current_stream().cuda_stream
    fe._score_one(ws, ws.best_Rt, stream)
    fe._tables(ws, stream, dense=True)
    torch.cuda.synchronize()
    assert torch.equal(ws.n_inl, fused["n_inl"])
    assert torch.equal(ws.inl_rows[:total], fused["inl_rows"][:total])
    assert torch.equal(ws.inl_loc[:total], fused["inl_loc"][:total])


def test_join_prepares_the_scoring_workspace(device):
    """The workspace the join fills on the way (slam_join_consensus_prepared) is byte for byte
    the one the scoring call prepares for itself, and the counts do not depend on who did it."""
    k, m1, m2 = read_cameras()
    seq = SyntheticSequence(700, 5, seed=31)
    f = seq.frames[2]                                    # pairs 1 and 2 end up without consensus matches
    seq.frames[2] = type(f)(f.kpL[:0], f.kpR[:0], f.descL[:0], f.descR[:0], f.idL[:0], f.idR[:0], f.Rt)
    P, H = 4, 130
    Rt = torch.from_numpy(np.stack([seeded_hypotheses(seq.relative_pose(p, p + 1), H, seed=p) for p in range(P)])).to(device)
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seq.frames, device)
    ws = fe.prepare(tb, n_hyp=H)
    stream = torch.cuda.current_stream().cuda_stream
    fe.stereo_match(ws, stream)
    fe.triangulate(ws, stream)
    fe.temporal_match(ws, stream)
    ws.score_work.zero_()
    fe.join(ws, stream)
    assert ws.work_ready == 1
    by_join = ws.score_work.clone()
    fe.score(ws, Rt, stream)
    counts_a = ws.counts.clone()
    ws.score_work.zero_()
    ws.work_ready = 0
    fe.score(ws, Rt, stream)                             # prepares the workspace itself
    torch.cuda.synchronize()
    assert torch.equal(by_join.view(torch.int64), ws.score_work.view(torch.int64))
    assert torch.equal(counts_a, ws.counts)
    n = ws.cons.count.cpu().numpy()[:P]
    assert n[0] > 50 and n[1] == 0 and n[2] == 0 and int(counts_a[0].max()) > 0


def test_two_batches_in_flight_on_two_streams(device):
    """The library keeps no state between calls: two workspaces driven from two streams at the
    same time (as bench.py does) give what each gives alone, for the given-hypotheses path and
    for the device-side RANSAC."""
    k, m1, m2 = read_cameras()
    seqs = [SyntheticSequence(900, 6, seed=41), SyntheticSequence(700, 6, seed=42)]
    P, H = 5, 600
    fe = pipeline.FrontEnd(k, m1, m2)
    lanes = []
    for s_, seq in enumerate(seqs):
        tb = pipeline.FrameTables.from_frames(seq.frames, device)
        ws = fe.prepare(tb, n_hyp=H)
        fe.prepare_ransac(ws)
        Rt = torch.from_numpy(np.stack([seeded_hypotheses(seq.relative_pose(p, p + 1), H, seed=7 * s_ + p)
                                        for p in range(P)])).to(device)
        lanes.append((ws, Rt, torch.cuda.Stream()))
    names = ("counts", "best", "n_inl", "inl_rows", "inl_loc")
    rnames = ("iters", "best_count", "n_inl", "inl_rows", "pose")

    def snap(ws, which):
        return {n: getattr(ws, n).clone() for n in which}

    alone, alone_r = [], []
    for ws, Rt, _ in lanes:
        fe.run(ws, Rt)
        torch.cuda.synchronize()
        alone.append(snap(ws, names))
        fe.run_ransac(ws, seed=9)
        torch.cuda.synchronize()
        alone_r.append(snap(ws, rnames))
    for mode, want, which in (("run", alone, names), ("ransac", alone_r, rnames)):
        for ws, _, _ in lanes:
            for n in which:
                getattr(ws, n).zero_()
        torch.cuda.synchronize()
        for _ in range(3):                                  # several rounds, the streams free to overlap
            for ws, Rt, st in lanes:
                with torch.cuda.stream(st):
                    if mode == "run":
                        fe.run(ws, Rt)
                    else:
                        fe.run_ransac(ws, seed=9)
        torch.cuda.synchronize()
        for (ws, _, _), ref in zip(lanes, want):
            total = int(ref["n_inl"].sum())
            for n in which:
                a, b = ref[n], getattr(ws, n)
                if n.startswith("inl_"):
                    a, b = a[:total], b[:total]
                assert np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True), (mode, n)


def test_frames_without_left_keypoints(device):
    """A batch whose reference frames have no left keypoints at all: no slot per pair (cap = 0), eight
    pairs -- every stage runs, nothing is written out of bounds, every count is zero."""
    k, m1, m2 = read_cameras()
    seq = SyntheticSequence(60, 9, seed=2)
    frames = []
    for f in seq.frames:
        frames.append(type(f)(f.kpL[:0], f.kpR, f.descL[:0], f.descR, f.idL[:0], f.idR, f.Rt))
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(frames, device)
    H = 16
    ws = fe.prepare(tb, n_hyp=H)
    assert ws.cap == 0 and ws.P == 8 and ws.score_work.numel() >= 16 * ws.P
    guard = ws.score_work.clone().fill_(7.0)
    ws.score_work.copy_(guard)
    Rt = torch.zeros((ws.P, H, 12), dtype=torch.float64, device=device)
    fe.run(ws, Rt)
    fe.run_ransac(ws, seed=1)
    torch.cuda.synchronize()
    assert torch.equal(ws.score_work, guard)                   # nothing to lay out, nothing touched
    assert int(ws.cons.count.abs().sum()) == 0 and int(ws.counts.abs().sum()) == 0 and int(ws.n_inl.abs().sum()) == 0
    assert int(ws.found.sum()) == 0
    # only the later frame of every pair is empty: slots exist, still no consensus
    mixed = [seq.frames[0]] + frames[1:3]
    tb2 = pipeline.FrameTables.from_frames(mixed, device)
    ws2 = fe.prepare(tb2, n_hyp=H)
    fe.run(ws2, torch.zeros((ws2.P, H, 12), dtype=torch.float64, device=device))
    torch.cuda.synchronize()
    assert ws2.cap == 60 and int(ws2.cons.count.abs().sum()) == 0
