"""GPU tests against the committed golden vectors (tests/golden/frontend_golden.npz, made
by the statement-by-statement numpy twin), and of the whole device pipeline -- including
winner selection, inlier tables, the gather and the track-store replay -- against the
oracle-built tables."""
import os

import numpy as np
import pytest
import torch

from slam_amd import ops, pipeline, sharding
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses

from helpers import gpu_match

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_golden.npz"))


def test_knn_golden(device):
    items = [(GOLD[f"knn_q{n}"], GOLD[f"knn_t{n}"]) for n in range(int(GOLD["knn_n"]))]
    by_d = {}
    for n, (q, t) in enumerate(items):
        by_d.setdefault(q.shape[1], []).append(n)
    for D, ns in by_d.items():
        stride = 32 if D <= 32 else 64
        idx, dist, _ = gpu_match([items[n] for n in ns], D, stride, 2, device)
        assert np.array_equal(idx, np.concatenate([GOLD[f"knn_idx{n}"] for n in ns]))
        assert np.array_equal(dist, np.concatenate([GOLD[f"knn_dist{n}"] for n in ns]))


def test_ratio_and_row_filter_golden(device):
    thrs = GOLD["ratio_thr"]
    grid = np.unpackbits(GOLD["ratio_grid"])[:len(thrs) * 489 * 489].reshape(len(thrs), 489, 489).astype(bool)
    a, b = np.meshgrid(np.arange(489), np.arange(489), indexing="ij")
    idx = np.zeros((489 * 489, 2), np.int32)
    idx[:, 1] = 1
    dist = np.stack([a.ravel(), b.ravel()], axis=1).astype(np.int32)
    tidx, tdist = torch.from_numpy(idx).to(device), torch.from_numpy(dist).to(device)
    for ti, thr in enumerate(thrs):
        keep = ops.ratio_filter(tidx, tdist, float(thr)).cpu().numpy().astype(bool)
        assert np.array_equal(keep, grid[ti].ravel()), thr
    y1, y2 = torch.from_numpy(GOLD["row_y1"]).to(device), torch.from_numpy(GOLD["row_y2"]).to(device)
    for thr in (1, 2):
        assert np.array_equal(ops.row_filter(y1, y2, thr).cpu().numpy().astype(bool), GOLD[f"row_keep{thr}"])


def test_dlt_golden(device):
    t = lambda n: torch.from_numpy(GOLD[n]).to(device)
    X, Y, Z, valid = ops.triangulate_dlt(GOLD["dlt_P"], GOLD["dlt_Q"], t("dlt_x1"), t("dlt_y1"), t("dlt_x2"),
                                         t("dlt_y2"))
    got = torch.stack([X, Y, Z], dim=1).cpu().numpy()
    ref = GOLD["dlt_X"]
    assert (np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)).max() < 1e-4   # the contract vs LAPACK
    assert np.array_equal(valid.cpu().numpy().astype(bool), GOLD["dlt_valid"])


def test_mini_sequence_golden_through_the_pipeline(device):
    k, m1, m2 = read_cameras()
    seq = SyntheticSequence(int(GOLD["seq_n_kp"]), int(GOLD["seq_n_frames"]))
    H = GOLD["seq_Rt0"].shape[0]
    P = len(seq.frames) - 1
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seq.frames, device)
    ws = fe.prepare(tb, n_hyp=H, want_masks=True)
    Rt = torch.from_numpy(np.stack([GOLD[f"seq_Rt{p}"] for p in range(P)])).to(device)
    fe.run(ws, Rt)
    torch.cuda.synchronize()
    smap = ws.smap.cpu().numpy()
    valid = ws.valid.cpu().numpy()
    pts = torch.stack([ws.X, ws.Y, ws.Z], dim=1).cpu().numpy()
    for f in range(len(seq.frames)):
        s = slice(int(tb.offL[f]), int(tb.offL[f + 1]))
        q = np.flatnonzero(smap[s] >= 0)
        assert np.array_equal(np.stack([q, smap[s][q]]), GOLD[f"seq_map{f}"])
        keys = np.flatnonzero(valid[s])
        assert np.array_equal(keys, GOLD[f"seq_pt_keys{f}"])
        ref = GOLD[f"seq_pts{f}"]
        assert (np.abs(pts[s][keys] - ref).max(axis=1) / np.abs(ref).max(axis=1)).max() < 1e-4
    counts = ws.counts.cpu().numpy()
    masks = ws.masks.cpu().numpy()
    cons = pipeline.consensus_to_host(ws)
    for p in range(P):
        assert np.array_equal(cons[p]["rows"], GOLD[f"seq_cons{p}"])
        gm = GOLD[f"seq_masks{p}"]
        bits = np.unpackbits(masks[p].view(np.uint8), axis=1, bitorder="little")[:, :gm.shape[1]].astype(bool)
        assert np.array_equal(bits, gm)
        assert np.array_equal(counts[p], gm.sum(axis=1))


def test_pipeline_select_gather_replay_equals_oracle_tables(device):
    from test_sharding_gloo import shard_tables
    k, m1, m2 = read_cameras()
    n_kp, n_frames, H = 220, 8, 40
    seq = SyntheticSequence(n_kp, n_frames)
    gt = [seq.relative_pose(p, p + 1) for p in range(n_frames - 1)]
    Rt_host = np.stack([seeded_hypotheses(gt[p], H, seed=6 + p) for p in range(n_frames - 1)])
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seq.frames, device)
    ws = fe.prepare(tb, n_hyp=H)
    fe.run(ws, torch.from_numpy(Rt_host).to(device))
    torch.cuda.synchronize()
    ref = shard_tables(seq.frames, gt, 0)
    P = n_frames - 1
    assert torch.equal(ws.n_inl[:P].cpu(), ref["n_inl"])
    assert torch.equal(ws.cons.count[:P].cpu(), ref["n_cons"])
    T = int(ref["n_inl"].sum())
    assert torch.equal(ws.inl_rows[:T].cpu(), ref["rows"])
    assert torch.equal(ws.inl_loc[:T].cpu(), ref["loc"])
    assert torch.equal(ws.best_Rt[:P].cpu(), ref["pose"])
    shards = sharding.TableGather(ws, 1, 0).run()
    db = sharding.replay_tracks(shards)
    assert db == sharding.replay_tracks([ref])
    assert db.get_max_track() == n_frames


def test_winner_tables_and_replay_equal_the_reference_track_store(device):
    """GPU winner tables -> gather -> sequential replay against the state the REFERENCE's own
    models/TrackDatabase.py reached on the same mini-sequence (tests/golden/trackdb_golden.npz,
    generated by importing the reference class: tests/golden/make_trackdb_golden.py)."""
    from trackdb_state import assert_same, query_arrays, state_arrays
    T = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trackdb_golden.npz"))
    k, m1, m2 = read_cameras()
    n_kp, n_frames, H = int(T["mini_n_kp"]), int(T["mini_n_frames"]), int(T["mini_H"])
    seq = SyntheticSequence(n_kp, n_frames)
    P = n_frames - 1
    Rt_host = np.stack([seeded_hypotheses(seq.relative_pose(p, p + 1), H, seed=6 + p) for p in range(P)])
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seq.frames, device)
    ws = fe.prepare(tb, n_hyp=H)
    fe.run(ws, torch.from_numpy(Rt_host).to(device))
    torch.cuda.synchronize()
    assert np.array_equal(ws.best.cpu().numpy()[:P], T["mini_best"])
    assert np.array_equal(ws.n_inl.cpu().numpy()[:P], T["mini_n_inl"])
    assert np.array_equal(ws.cons.count.cpu().numpy()[:P], T["mini_n_cons"])
    n = int(T["mini_n_inl"].sum())
    assert np.array_equal(ws.inl_rows.cpu().numpy()[:n], T["mini_rows"])
    assert np.array_equal(ws.inl_loc.cpu().numpy()[:n], T["mini_loc"])
    db = sharding.replay_tracks(sharding.TableGather(ws, 1, 0).run(), first=m1)
    want = {key[len("mini_state_"):]: T[key] for key in T.files if key.startswith("mini_state_")}
    assert_same({**state_arrays(db), **query_arrays(db)}, want, "gpu tables -> replay")
