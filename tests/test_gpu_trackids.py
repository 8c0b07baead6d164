"""Device-side track-id assignment (slam_track_assign) and the array-backed track store built
from it: the arrays equal the host row walk's, the store equals the reference's own TrackDatabase
state (tests/golden/trackdb_golden.npz) and the sequential host replay."""
import os

import numpy as np
import pytest
import torch

from slam_amd import ops, pipeline, sharding
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses
from slam_amd.track_database import ArrayTrackDatabase, TrackDatabase
from trackdb_state import arrays_from_rows, assert_same, chain_tables, query_arrays, state_arrays

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trackdb_golden.npz"))


def _assign(device, rows, loc, n_inl, kp_cap, first=0, pad=13):
    """Through the C ABI with unused capacity behind the valid rows."""
    r = np.concatenate([rows, np.full((pad, 4), -5, np.int32)])
    l = np.concatenate([loc, np.full((pad, 6), np.nan, np.float32)])
    out = ops.track_assign(torch.from_numpy(r).to(device), torch.from_numpy(l).to(device),
                           torch.from_numpy(np.asarray(n_inl, np.int32)).to(device), kp_cap, first)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items() if k != "work"}


@pytest.mark.parametrize("first,n_pairs,n_kp,seed", [(0, 40, 90, 1), (5, 33, 90, 2), (0, 1, 50, 3), (0, 300, 400, 4),
                                                     (2, 2500, 64, 5)])
def test_track_assign_equals_the_row_walk(device, first, n_pairs, n_kp, seed):
    long_chains = n_pairs >= 300                  # tracks hundreds of frames long: pointer jumping goes all the way
    rows, loc, n_inl = chain_tables(n_pairs, n_kp, seed, empty_every=0 if long_chains else 7,
                                    keep=0.995 if long_chains else 0.8, full=long_chains)
    want = arrays_from_rows(first, rows, loc, n_inl)
    got = _assign(device, rows, loc, n_inl, n_kp, first)
    T, nt, N = rows.shape[0], want["n_tracks"], want["n_points"]
    assert got["summary"][:5].tolist() == [T, nt, N, want["longest"], 0]
    assert np.array_equal(got["row_track"][:T], want["row_track"]) and np.array_equal(got["row_depth"][:T], want["row_depth"])
    assert np.array_equal(got["track_start"][:nt + 1], want["track_start"])
    assert np.array_equal(got["pt_kp"][:N], want["pt_kp"]) and np.array_equal(got["pt_frame"][:N], want["pt_frame"])
    assert np.array_equal(got["pt_loc"][:N], want["pt_loc"])
    if long_chains:
        assert want["longest"] > 100


def test_track_assign_reports_what_it_does_not_cover(device):
    # the reference-generated random scenario has two rows of a pair continuing one track: flag 1
    rows, loc, n_inl = GOLD["rand_rows"], GOLD["rand_loc"], GOLD["rand_n_inl"]
    got = _assign(device, rows, loc, n_inl, int(GOLD["rand_n_kp"]), int(GOLD["rand_first_frame"]))
    assert got["summary"][4] & 1
    # keypoint index beyond kp_cap: flag 2; more rows than the tables hold: flag 4 and nothing done
    rows, loc, n_inl = chain_tables(6, 40, 9, empty_every=0)
    assert _assign(device, rows, loc, n_inl, 20)["summary"][4] & 2
    got = _assign(device, rows, loc, n_inl + 100, 40)
    assert got["summary"][4] == 4 and got["summary"][0] == 0
    # nothing to do
    lib = ops._lib.load()
    s = torch.full((8,), 7, dtype=torch.int32, device=device)
    import ctypes
    null = ctypes.c_void_p(0)
    assert lib.slam_track_assign(null, null, null, 0, 0, 10, 0, null, null, null, null, null, null,
                                 ctypes.c_void_p(s.data_ptr()), null, 0, null) == 0
    torch.cuda.synchronize()
    assert s.cpu().tolist() == [0] * 8


def test_device_replay_equals_the_reference_track_store(device):
    """GPU winner tables -> gather -> slam_track_assign -> ArrayTrackDatabase against the state of the
    reference's own class on the same mini-sequence; the fallback (rows the kernels do not cover)
    against the reference-generated random scenario."""
    k, m1, m2 = read_cameras()
    n_kp, n_frames, H = int(GOLD["mini_n_kp"]), int(GOLD["mini_n_frames"]), int(GOLD["mini_H"])
    seq = SyntheticSequence(n_kp, n_frames)
    P = n_frames - 1
    Rt_host = np.stack([seeded_hypotheses(seq.relative_pose(p, p + 1), H, seed=6 + p) for p in range(P)])
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seq.frames, device)
    ws = fe.prepare(tb, n_hyp=H)
    fe.run(ws, torch.from_numpy(Rt_host).to(device))
    shards = sharding.TableGather(ws, 1, 0).run()
    db = sharding.replay_tracks_device(shards, first=m1)
    assert isinstance(db, ArrayTrackDatabase) and db._arrays is not None
    want = {key[len("mini_state_"):]: GOLD[key] for key in GOLD.files if key.startswith("mini_state_")}
    assert_same(query_arrays(db), {a: b for a, b in want.items() if a.startswith("q_")}, "device replay getters")
    assert db._arrays is not None
    host = sharding.replay_tracks(shards, first=m1)
    assert np.allclose(db.ex_matrices, host.ex_matrices, rtol=0, atol=1e-12)
    assert np.allclose(db.camera_positions, host.camera_positions, rtol=0, atol=1e-12)
    assert_same(state_arrays(db), {a: b for a, b in want.items() if not a.startswith("q_")}, "device replay state")
    db.set_ex_matrices(host.ex_matrices)
    db.set_ex_camera_positions(host.camera_positions)
    host.get_mean_track_length(), host.get_mean_frame_links()     # (the getters cache their answer in the state)
    assert db == host
    # three shards (as gathered from three ranks), one of them on the host
    n_inl = ws.n_inl[:P].cpu()
    cuts, parts = [0, 2, 5, P], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        r0, r1 = int(n_inl[:a].sum()), int(n_inl[:b].sum())
        sh = {kk: shards[0][kk][a:b] for kk in ("n_inl", "n_cons", "pose")}
        sh["rows"] = torch.cat([shards[0]["rows"][r0:r1], torch.full((4, 4), -3, dtype=torch.int32, device=device)])
        sh["loc"] = torch.cat([shards[0]["loc"][r0:r1], torch.zeros((4, 6), device=device)])
        parts.append(sh)
    parts[1] = {kk: v.cpu() for kk, v in parts[1].items()}
    db3 = sharding.replay_tracks_device(parts, first=m1, device=device)
    assert isinstance(db3, ArrayTrackDatabase)
    db3.set_ex_matrices(host.ex_matrices)
    db3.set_ex_camera_positions(host.camera_positions)
    db3.get_mean_track_length(), db3.get_mean_frame_links()
    assert db3 == host
    # fallback
    first = int(GOLD["rand_first_frame"])
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(device)
    Pr = len(GOLD["rand_n_inl"])
    sh = dict(n_inl=t(GOLD["rand_n_inl"], torch.int32), n_cons=t(GOLD["rand_n_cons"], torch.int32),
              rows=t(GOLD["rand_rows"], torch.int32), loc=t(GOLD["rand_loc"], torch.float32),
              pose=torch.eye(3, 4, dtype=torch.float64, device=device).repeat(Pr, 1, 1).reshape(Pr, 12))
    dbr = sharding.replay_tracks_device([sh], first_frame=first)
    assert not isinstance(dbr, ArrayTrackDatabase)
    wantr = {key[len("rand_state_"):]: GOLD[key] for key in GOLD.files if key.startswith("rand_state_")}
    assert_same({**state_arrays(dbr), **query_arrays(dbr)}, wantr, "fallback")
