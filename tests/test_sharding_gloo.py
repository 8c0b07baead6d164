"""CPU test of the multi-rank path: two `gloo` ranks each build the inlier tables of
their own frame range (here with the CPU oracle standing in for the kernels), the tables
are gathered to rank 0 exactly as bench.py / the GPU pipeline gathers them over RCCL, and
the replayed track database must equal the single-process one."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu as oracle
from slam_amd import sharding
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses

N_KP, N_FRAMES, H = 220, 8, 40


def shard_tables(frames, poses_gt, first_pair):
    """Inlier tables of the pairs inside `frames` (a shard incl. its halo frame): winner =
    first-best of a seeded hypothesis set, as the device pipeline's select stage does."""
    k, m1, m2 = read_cameras()
    P, Q = k @ m1, k @ m2
    st = []
    for fr in frames:
        idx, _ = oracle.hamming_knn(fr.descL, fr.descR, 1)
        keep = oracle.row_filter(idx, fr.kpL[:, 1], fr.kpR[:, 1], 2.0)
        j = np.where(keep > 0, idx[:, 0], 0)
        X, Y, Z, valid = oracle.triangulate_dlt(P, Q, fr.kpL[:, 0], fr.kpL[:, 1], fr.kpR[j, 0], fr.kpR[j, 1])
        st.append((np.where(keep > 0, idx[:, 0], -1).astype(np.int32), (valid * (keep > 0)).astype(np.uint8), X, Y, Z))
    n_inl, n_cons, rows_all, loc_all, pose = [], [], [], [], []
    for p in range(len(frames) - 1):
        f0, f1 = frames[p], frames[p + 1]
        tidx, tdist = oracle.hamming_knn(f0.descL, f1.descL, 2)
        rows = oracle.join(tidx, oracle.ratio_filter(tidx, tdist, 1.0), st[p][0], st[p][1], st[p + 1][0])
        Rt = seeded_hypotheses(poses_gt[p], H, seed=6 + first_pair + p)
        pts = (st[p][2][rows[:, 0]], st[p][3][rows[:, 0]], st[p][4][rows[:, 0]], f1.kpL[rows[:, 2], 0],
               f1.kpL[rows[:, 2], 1], f1.kpR[rows[:, 3], 0], f1.kpR[rows[:, 3], 1])
        counts = oracle.score_hypotheses(Rt, k, m2[:, 3], *pts, 2.0)
        best = int(np.argmax(counts))
        _, masks = oracle.score_hypotheses(Rt[best:best + 1], k, m2[:, 3], *pts, 2.0, want_masks=True)
        bits = np.unpackbits(masks[0].view(np.uint8), bitorder="little")[:len(rows)].astype(bool)
        r = rows[bits]
        loc = np.stack([f0.kpL[r[:, 0], 0], f0.kpR[r[:, 1], 0], f0.kpL[r[:, 0], 1], f1.kpL[r[:, 2], 0],
                        f1.kpR[r[:, 3], 0], f1.kpL[r[:, 2], 1]], axis=1).astype(np.float32)
        n_inl.append(len(r)); n_cons.append(len(rows)); rows_all.append(r); loc_all.append(loc); pose.append(Rt[best])
    return dict(n_inl=torch.tensor(n_inl, dtype=torch.int32), n_cons=torch.tensor(n_cons, dtype=torch.int32),
                rows=torch.from_numpy(np.concatenate(rows_all).astype(np.int32)),
                loc=torch.from_numpy(np.concatenate(loc_all)),
                pose=torch.from_numpy(np.stack(pose).reshape(-1, 12)))


def _sequence():
    seq = SyntheticSequence(N_KP, N_FRAMES)
    gt = [seq.relative_pose(p, p + 1) for p in range(N_FRAMES - 1)]
    return seq, gt


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seq, gt = _sequence()
        s, e = sharding.shard_pairs(N_FRAMES, world)[rank]
        mine = shard_tables(seq.frames[s:e + 1], gt[s:e], s)          # frames s..e: one-frame halo
        g = sharding.TableGather(world=world, rank=rank)
        shards = g.gather(mine["n_inl"], mine["rows"], mine["loc"], mine["pose"], mine["n_cons"])
        if rank == 0:
            db = sharding.replay_tracks(shards)
            db.serialize(out_path)
        else:
            assert shards is None
    finally:
        dist.destroy_process_group()


class _Obj:
    pass


def _fake_workspace(tables, pairs_cap, rows_cap):
    """A workspace as pipeline.FrontEnd.prepare lays it out, filled from host tables."""
    ws = _Obj()
    ws.P = tables["n_inl"].shape[0]
    ws.layout = sharding.TableLayout(pairs_cap, rows_cap)
    ws.arena = torch.zeros(ws.layout.nbytes, dtype=torch.uint8)
    ws.table_views = v = ws.layout.views(ws.arena)
    T = tables["rows"].shape[0]
    v["n_inl"][:ws.P] = tables["n_inl"]
    v["rows"][:T] = tables["rows"]
    v["loc"][:T] = tables["loc"]
    v["pose_a"][:ws.P] = tables["pose"]
    v["rows"][T:] = -7                      # capacity behind the valid rows must never be read
    ws.cons = _Obj()
    ws.cons.count = tables["n_cons"]
    return ws


def _arena_worker(rank, world, port, out_path, common_shape):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seq, gt = _sequence()
        s, e = sharding.shard_pairs(N_FRAMES, world)[rank]
        mine = shard_tables(seq.frames[s:e + 1], gt[s:e], s)
        pairs, rows = e - s, (e - s) * N_KP
        if common_shape:
            pairs, rows = sharding.agree_table_shape(pairs, rows)
        g = sharding.TableGather(_fake_workspace(mine, pairs, rows), world, rank)
        assert g.fixed == common_shape
        for _ in range(2):                   # the receive buffers are reused from step to step
            shards = g.run()
        if rank == 0:
            sharding.replay_tracks(shards).serialize(out_path)
        else:
            assert shards is None
    finally:
        dist.destroy_process_group()


def test_fixed_arena_gather_equals_single_process(tmp_path):
    """The one-collective arena gather (equal arena shapes on all ranks) and its fallback
    (shapes differ: 3 ranks over 7 pairs) both reproduce the single-process track database."""
    seq, gt = _sequence()
    ref = sharding.replay_tracks([shard_tables(seq.frames, gt, 0)])
    from slam_amd.track_database import TrackDatabase
    for world, common in ((2, True), (3, True), (3, False)):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out_path = str(tmp_path / f"db_{world}_{common}.pkl")
        mp.spawn(_arena_worker, args=(world, port, out_path, common), nprocs=world, join=True)
        assert TrackDatabase(out_path) == ref


def test_table_layout_views():
    lay = sharding.TableLayout(5, 33)
    arena = torch.zeros(lay.nbytes, dtype=torch.uint8)
    v = lay.views(arena)
    assert v["pose_a"].shape == (5, 12) and v["pose_b"].shape == (5, 12) and v["rows"].shape == (33, 4)
    assert v["loc"].shape == (33, 6) and v["n_inl"].shape == (5,) and v["n_cons"].shape == (5,)
    for i, (name, t) in enumerate(v.items()):
        t.fill_(i + 1)
    for i, (name, t) in enumerate(v.items()):        # no two fields overlap
        assert (t == i + 1).all(), name
        assert lay.offset[name][0] % 16 == 0


def test_two_rank_gather_and_replay_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / "db.pkl")
    mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
    from slam_amd.track_database import TrackDatabase
    got = TrackDatabase(out_path)
    seq, gt = _sequence()
    whole = shard_tables(seq.frames, gt, 0)
    ref = sharding.replay_tracks([whole])
    assert got == ref
    assert got.get_num_frames() == N_FRAMES and got.get_num_tracks() > 50
    assert got.get_max_track() == N_FRAMES      # tracks continue across the shard boundary
    assert len(got.get_inliers_ratio_per_frame()) == N_FRAMES - 1
    # uneven split (3 ranks' worth of ranges, replayed in order) gives the same store too
    parts = [shard_tables(seq.frames[s:e + 1], gt[s:e], s) for s, e in sharding.shard_pairs(N_FRAMES, 3)]
    assert sharding.replay_tracks(parts) == ref


def test_single_rank_gather_is_identity():
    seq, gt = _sequence()
    t = shard_tables(seq.frames[:3], gt[:2], 0)
    res = sharding.TableGather(world=1, rank=0).gather(t["n_inl"], t["rows"], t["loc"], t["pose"], t["n_cons"])
    assert len(res) == 1 and torch.equal(res[0]["rows"], t["rows"]) and torch.equal(res[0]["n_inl"], t["n_inl"])
