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


# ---- the array-backed store -----------------------------------------------------------------------
from slam_amd.track_database import ArrayTrackDatabase       # noqa: E402
from trackdb_state import arrays_from_rows, chain_tables      # noqa: E402


def _array_db(first, rows, loc, n_inl):
    a = arrays_from_rows(first, rows, loc, n_inl)
    return ArrayTrackDatabase(first, n_inl, a["row_track"], a["row_depth"], a["track_start"], a["pt_kp"], a["pt_frame"],
                              a["pt_loc"], max_length=a["longest"])


def test_array_backed_store_answers_like_the_reference():
    """Getters straight from the arrays (nothing materialised), then the dict form on demand."""
    first, rows, loc, n_inl, n_cons = _inputs("mini")
    db = _array_db(first, rows, loc, n_inl)
    db.set_frame_records(np.where(n_cons > 0, n_inl / np.maximum(n_cons, 1), 0.0), n_inl.astype(np.float64))
    want = _want("mini")
    got_q = query_arrays(db)
    assert db._arrays is not None, "a getter built the dict form"
    assert_same(got_q, {k: v for k, v in want.items() if k.startswith("q_")}, "mini/array getters")
    assert_same(state_arrays(db), {k: v for k, v in want.items() if not k.startswith("q_")}, "mini/array state")
    assert db._arrays is None                                  # the state comparison reads the dicts
    assert_same(query_arrays(db), {k: v for k, v in want.items() if k.startswith("q_")}, "mini/after materialising")


@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("first,n_pairs,seed", [(0, 40, 1), (5, 33, 2), (0, 1, 3), (0, 9, 4)])
def test_array_backed_store_equals_the_row_walk(first, n_pairs, seed, native, tmp_path, monkeypatch):
    if not native:
        monkeypatch.setattr(track_database, "_native", None)      # the pure-Python forms of the same containers
    elif track_database._native is None:
        pytest.skip("C helper not built")
    rows, loc, n_inl = chain_tables(n_pairs, 90, seed)
    want = TrackDatabase()
    off = 0
    for p, m in enumerate(n_inl.tolist()):
        want.update_tracks_from_tables(first + p, rows[off:off + m], loc[off:off + m])
        off += m
    db = _array_db(first, rows, loc, n_inl)
    # every getter, before anything is materialised
    for f in range(first - 1, first + n_pairs + 3):
        assert sorted(db.get_track_ids_for_frame(f)) == sorted(want.get_track_ids_for_frame(f))
    assert len(db.get_frameIds()) == len(want.get_frameIds()) and list(db.get_frameIds()) == list(want.get_frameIds())
    assert db.get_frameIds() == want.get_frameIds() and db.get_trackIds() == want.get_trackIds()
    for t in list(want.get_trackIds())[::7] + [10 ** 6]:
        if t in want.get_trackIds():
            assert db.get_track_data(t) == want.get_track_data(t)
            assert [type(x) for x in db.get_track_data(t)[0]] == [type(x) for x in want.get_track_data(t)[0]]
            assert db.get_all_points_in_track(t) == want.get_all_points_in_track(t)
        assert db.get_frame_ids_for_track(t) == want.get_frame_ids_for_track(t)
        assert db.get_feature_locations(first + 2, t) == want.get_feature_locations(first + 2, t)
    assert (db.get_num_tracks(), db.get_num_frames(), db.get_max_track(), db.get_min_track()) == \
        (want.get_num_tracks(), want.get_num_frames(), want.get_max_track(), want.get_min_track())
    assert db.get_mean_track_length() == want.get_mean_track_length()
    assert db.get_mean_frame_links() == want.get_mean_frame_links()
    assert np.array_equal(db.get_track_length_data(), want.get_track_length_data())
    assert db.get_tracks_in_bundle_window(first, first + 4) == want.get_tracks_in_bundle_window(first, first + 4)
    for L in (2, 3, 5):
        assert db.get_all_tracks_of_length(L) == want.get_all_tracks_of_length(L)
        import random
        random.seed(L)
        a = db.get_random_track_of_length(L)
        random.seed(L)
        assert a == want.get_random_track_of_length(L)
    if first == 0:
        assert np.array_equal(db.calculate_connectivity_data()[1], want.calculate_connectivity_data()[1])
    assert db._arrays is not None
    # the pickle is the plain schema, loadable by the plain class; the object itself has become a plain store
    path = str(tmp_path / "a.pkl")
    db.serialize(path)
    assert TrackDatabase(path) == want and db == want
    # ... and keeps working as one: more rows continue the tracks
    more_rows, more_loc, more_n = chain_tables(3, 90, seed + 100, empty_every=0)
    for d in (db, want):
        d.update_tracks_from_pair_tables(first + n_pairs, more_rows, more_loc, more_n)
    assert db == want
