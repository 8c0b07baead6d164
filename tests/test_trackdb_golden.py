"""The track store against the reference's own class.

tests/golden/trackdb_golden.npz holds states and getter answers produced by the reference's
models/TrackDatabase.py itself (imported unmodified in the build container by
tests/golden/make_trackdb_golden.py and driven the way utils/utils.py:277-282 drives it).  Every
route this package has into its track store must end in exactly that state: the reference's
per-row entry point, the table form, the C helper, and the sequential replay of gathered tables."""
import os

import numpy as np
import pytest

from slam_amd import sharding, track_database
from slam_amd.track_database import TrackDatabase
from trackdb_state import assert_same, query_arrays, state_arrays

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trackdb_golden.npz"))


def _want(scn):
    pre = f"{scn}_state_"
    return {k[len(pre):]: GOLD[k] for k in GOLD.files if k.startswith(pre)}


def _inputs(scn):
    first = int(GOLD[f"{scn}_first_frame"]) if f"{scn}_first_frame" in GOLD.files else 0
    return first, GOLD[f"{scn}_rows"], GOLD[f"{scn}_loc"], GOLD[f"{scn}_n_inl"], GOLD[f"{scn}_n_cons"]


def _keypoints(scn, first, n_pairs):
    if scn == "rand":
        return {first + i: (GOLD["rand_kpL"][i], GOLD["rand_kpR"][i]) for i in range(n_pairs + 1)}
    from slam_amd.synthetic import SyntheticSequence
    seq = SyntheticSequence(int(GOLD["mini_n_kp"]), int(GOLD["mini_n_frames"]))
    return {f: (fr.kpL, fr.kpR) for f, fr in enumerate(seq.frames)}


class RowMatcher:
    """Only get_feature_location_frame, like the reference's Matcher seen from the track store."""

    def __init__(self, kp):
        self.kp = kp

    def get_feature_location_frame(self, frameId, kp, loc):
        x, y = self.kp[frameId][loc][kp]
        return float(x), float(y)


class ArrayMatcher(RowMatcher):
    def feature_location_arrays(self, frameId):
        return self.kp[frameId]


def _finish(db, first, n_inl, n_cons):
    for p, m in enumerate(n_inl.tolist()):
        db.add_inliers_ratio(first + p, (m / n_cons[p]) if n_cons[p] else 0.0)
        db.add_num_matches(first + p, num_matches=float(m))


def _check(db, scn, what):
    got = {**state_arrays(db), **query_arrays(db)}
    assert_same(got, _want(scn), f"{scn}/{what}")


@pytest.mark.parametrize("scn", ["mini", "rand"])
@pytest.mark.parametrize("matcher_cls", [RowMatcher, ArrayMatcher])
def test_per_row_entry_point_equals_the_reference(scn, matcher_cls):
    first, rows, loc, n_inl, n_cons = _inputs(scn)
    m = matcher_cls(_keypoints(scn, first, len(n_inl)))
    db = TrackDatabase()
    off = 0
    for p, cnt in enumerate(n_inl.tolist()):
        tuples = [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), None) for r in rows[off:off + cnt]]
        db.update_tracks_from_frame(m, first + p, tuples)
        db.add_inliers_ratio(first + p, (cnt / n_cons[p]) if n_cons[p] else 0.0)
        db.add_num_matches(first + p, num_matches=float(cnt))
        off += cnt
    _check(db, scn, matcher_cls.__name__)


@pytest.mark.parametrize("scn", ["mini", "rand"])
def test_table_form_equals_the_reference(scn):
    first, rows, loc, n_inl, n_cons = _inputs(scn)
    db = TrackDatabase()
    off = 0
    for p, cnt in enumerate(n_inl.tolist()):
        db.update_tracks_from_tables(first + p, rows[off:off + cnt], loc[off:off + cnt])
        off += cnt
    _finish(db, first, n_inl, n_cons)
    _check(db, scn, "tables")


@pytest.mark.skipif(track_database._native is None, reason="C helper not built")
@pytest.mark.parametrize("scn", ["mini", "rand"])
def test_c_helper_equals_the_reference(scn):
    first, rows, loc, n_inl, n_cons = _inputs(scn)
    db = TrackDatabase()
    cut = len(n_inl) // 3                              # in two calls: carried state must line up
    t0 = int(n_inl[:cut].sum())
    db.update_tracks_from_pair_tables(first, rows[:t0], loc[:t0], n_inl[:cut])
    db.update_tracks_from_pair_tables(first + cut, rows[t0:], loc[t0:], n_inl[cut:])
    _finish(db, first, n_inl, n_cons)
    _check(db, scn, "C helper")


@pytest.mark.parametrize("scn", ["mini", "rand"])
def test_replay_of_gathered_shards_equals_the_reference(scn):
    import torch
    first, rows, loc, n_inl, n_cons = _inputs(scn)
    P = len(n_inl)
    cuts = [0, P // 3, P // 3, (2 * P) // 3, P]        # three shards and an empty one
    shards = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        r0, r1 = int(n_inl[:a].sum()), int(n_inl[:b].sum())
        pad = 5                                        # unused capacity behind the valid rows, as in the arena
        shards.append(dict(n_inl=torch.from_numpy(n_inl[a:b].astype(np.int32)),
                           n_cons=torch.from_numpy(n_cons[a:b].astype(np.int32)),
                           pose=torch.eye(3, 4, dtype=torch.float64).repeat(b - a, 1, 1).reshape(b - a, 12),
                           rows=torch.from_numpy(np.concatenate([rows[r0:r1], np.full((pad, 4), -9, np.int32)])),
                           loc=torch.from_numpy(np.concatenate([loc[r0:r1], np.zeros((pad, 6), np.float32)]))))
    db = sharding.replay_tracks(shards, first_frame=first)
    _check(db, scn, "replay")


def test_pickle_written_here_has_the_reference_schema(tmp_path):
    """Keys of models/TrackDatabase.py:121-137; the state survives the round trip."""
    import pickle
    first, rows, loc, n_inl, n_cons = _inputs("mini")
    db = TrackDatabase()
    db.update_tracks_from_pair_tables(first, rows, loc, n_inl)
    _finish(db, first, n_inl, n_cons)
    path = str(tmp_path / "db.pkl")
    db.serialize(path)
    with open(path, "rb") as f:
        d = pickle.load(f)
    assert set(d) == {"tracks", "frame_ids", "next_free_trackId", "last_insertions", "num_tracks", "num_frames",
                      "mean_track_length", "max_length", "min_length", "mean_frame_links",
                      "frame_id_to_inliers_ratio", "ex_camera_positions", "ex_matrices", "frame_id_to_num_matches"}
    back = TrackDatabase(path)
    _check(back, "mini", "pickle")
