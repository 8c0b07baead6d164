"""The C helper of the track store builds exactly the containers the Python loop builds."""
import time

import numpy as np
import pytest

from slam_amd import track_database
from slam_amd.track_database import TrackDatabase


def _tables(n_pairs, n_kp, seed, keep=0.7):
    """Inlier tables with real track structure: a later-frame keypoint index of pair p shows up
    as the earlier-frame index of pair p+1 for most rows; some indices repeat inside a pair."""
    rng = np.random.default_rng(seed)
    rows, loc, n_inl = [], [], []
    prev_cur = rng.permutation(n_kp)[: n_kp // 2]
    for _ in range(n_pairs):
        m = int(rng.integers(0, n_kp // 2)) if rng.random() < 0.9 else 0
        cont = prev_cur[rng.random(prev_cur.shape[0]) < keep][:m]
        fresh = rng.integers(0, n_kp, size=m - cont.shape[0])
        prev_left = np.concatenate([cont, fresh]).astype(np.int32)
        rng.shuffle(prev_left)
        cur_left = rng.integers(0, n_kp, size=m).astype(np.int32)      # collisions on purpose
        r = np.stack([prev_left, rng.integers(0, n_kp, m), cur_left, rng.integers(0, n_kp, m)], axis=1).astype(np.int32)
        rows.append(r)
        loc.append(rng.uniform(0, 1200, size=(m, 6)).astype(np.float32))
        n_inl.append(m)
        prev_cur = cur_left if m else prev_cur
    return np.concatenate(rows), np.concatenate(loc), np.asarray(n_inl, dtype=np.int64)


def _python_replay(first, rows, loc, n_inl):
    db = TrackDatabase()
    off = 0
    for p, m in enumerate(n_inl.tolist()):
        db.update_tracks_from_tables(first + p, rows[off:off + m], loc[off:off + m])
        off += m
    return db


@pytest.mark.skipif(track_database._native is None, reason="C helper not built")
@pytest.mark.parametrize("first,n_pairs", [(0, 60), (7, 60), (0, 64), (3, 47)])
def test_native_replay_equals_python_loop(first, n_pairs):
    rows, loc, n_inl = _tables(n_pairs, 300, seed=3 + first + n_pairs)
    if n_pairs == 64:
        n_inl = np.concatenate([n_inl, [0, 0]])          # trailing pairs without rows
    want = _python_replay(first, rows, loc, n_inl)
    got = TrackDatabase()
    # in two calls: the state carried between calls (parity slots, counters) must line up too
    cut = 25
    t0 = int(n_inl[:cut].sum())
    got.update_tracks_from_pair_tables(first, rows[:t0], loc[:t0], n_inl[:cut])
    got.update_tracks_from_pair_tables(first + cut, rows[t0:], loc[t0:], n_inl[cut:])
    assert got == want
    assert got.get_num_frames() == want.get_num_frames() and got.get_max_track() == want.get_max_track()
    assert got.get_min_track() == want.get_min_track()
    some = next(iter(want.get_trackIds()))
    p_got, p_want = got.get_track_data(some)[0], want.get_track_data(some)[0]
    assert [type(x) for x in p_got] == [type(x) for x in p_want] and type(p_got[1][0]) is float


@pytest.mark.skipif(track_database._native is None, reason="C helper not built")
def test_native_replay_empty_and_bad_input():
    db = TrackDatabase()
    db.update_tracks_from_pair_tables(0, np.zeros((0, 4), np.int32), np.zeros((0, 6), np.float32), np.zeros(3, np.int64))
    assert db.get_num_tracks() == 0 and db.get_min_track() == np.inf and len(db.get_frameIds()) == 4
    with pytest.raises(ValueError):
        db.update_tracks_from_pair_tables(0, np.zeros((2, 4), np.int32), np.zeros((2, 6), np.float32), np.array([5]))


@pytest.mark.skipif(track_database._native is None, reason="C helper not built")
def test_native_replay_is_faster():
    import gc
    rows, loc, n_inl = _tables(300, 2000, seed=11)
    gc.disable()
    t0 = time.perf_counter()
    want = _python_replay(0, rows, loc, n_inl)
    t1 = time.perf_counter()
    got = TrackDatabase()
    got.update_tracks_from_pair_tables(0, rows, loc, n_inl)
    t2 = time.perf_counter()
    gc.enable()
    assert got == want
    print(f"python {t1 - t0:.3f} s, native {t2 - t1:.3f} s for {int(n_inl.sum())} rows")
    assert (t2 - t1) < 2 * (t1 - t0)          # about 3x faster in practice; the slack is for a busy host


def test_frame_update_through_tables_equals_the_row_loop():
    """update_tracks_from_frame with a matcher that hands out coordinate arrays (the table
    route) builds the same store as the keypoint-by-keypoint route of the reference
    (models/TrackDatabase.py:34-53)."""
    from slam_amd.matcher import Matcher, PrecomputedFeatures
    from slam_amd.synthetic import SyntheticSequence

    seq = SyntheticSequence(300, 4, seed=5)
    matcher = Matcher(algo=PrecomputedFeatures(seq.frames))
    for i in range(4):
        matcher.read_images(i)
        matcher.detect_and_compute(i)

    class RowsOnly:                       # the reference's view of a matcher: one keypoint at a time
        def get_feature_location_frame(self, frameId, kp, loc):
            return matcher.get_feature_location_frame(frameId, kp, loc)

    rng = np.random.default_rng(2)
    a, b = TrackDatabase(), TrackDatabase()
    prev_cur = rng.permutation(300)[:150]
    for frame in range(3):
        m = 0 if frame == 1 else 120      # a pair without inliers in the middle
        prev_left = rng.permutation(prev_cur)[:m]
        cur_left = rng.integers(0, 300, m)
        rows = [(int(p), int(rng.integers(0, 300)), int(c), int(rng.integers(0, 300)), np.zeros(3))
                for p, c in zip(prev_left, cur_left)]
        a.update_tracks_from_frame(matcher, frame, rows)
        b.update_tracks_from_frame(RowsOnly(), frame, rows)
        prev_cur = cur_left if m else prev_cur
    assert a == b and a.get_num_tracks() > 0
    t = next(iter(a._tracks.values()))
    assert type(t[0][0]) is int and all(type(v) is float for v in t[0][1])


def test_keypoint_containers_remember_their_coordinates():
    """Lists / object arrays made from a coordinate array hand that array back instead of walking
    the KeyPoint objects; slices, copies and pickles fall back to the objects and agree."""
    import pickle

    from slam_amd import types

    xy = np.random.default_rng(0).uniform(0, 1200, (50, 2)).astype(np.float32)
    kps = types.keypoints_from_array(xy)
    assert isinstance(kps, list) and types.keypoints_to_array(kps) is kps.array
    assert kps[7].pt == (float(xy[7, 0]), float(xy[7, 1]))
    arr = types.keypoint_object_array(kps)
    assert arr.dtype == object and arr.shape == (50,) and arr[7] is kps[7]
    assert types.keypoints_to_array(arr) is kps.array
    for other in (arr[:10], arr.copy()[::2], pickle.loads(pickle.dumps(arr)), list(kps)[:10], kps + kps[:1]):
        got = types.keypoints_to_array(other)
        assert got is not kps.array and got.dtype == np.float32
        assert np.array_equal(got, np.array([k.pt for k in other], dtype=np.float32).reshape(-1, 2))
    kps.append(types.KeyPoint(1.0, 2.0))                    # a list that grew no longer matches its array
    assert types.keypoints_to_array(kps).shape == (51, 2)
