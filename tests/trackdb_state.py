"""Canonical array form of a track store's state, for comparing stores and for fixtures.

Works on any object with the attribute layout of the reference's track database
(models/TrackDatabase.py:17-32): `_tracks` {trackId: [(kp, (xl, xr, y), frameId), ...]},
`_frame_ids` {frameId: set(trackId)}, `_last_insertions` {0: {kp: trackId}, 1: {...}}, the
counters and the two per-frame dicts.  Everything comes out as plain numpy arrays (no pickle)."""
import numpy as np


def _next_free(db):
    for name in ("_TrackDatabase__next_free_trackId", "_next_free_trackId"):
        if hasattr(db, name):
            return int(getattr(db, name))
    raise AttributeError("track store has no next-free-track counter")


def state_arrays(db):
    tracks = db._tracks
    tids = list(tracks.keys())                       # dict order = order of creation
    lens = np.array([len(tracks[t]) for t in tids], dtype=np.int64)
    n = int(lens.sum())
    kp = np.empty(n, dtype=np.int64)
    fr = np.empty(n, dtype=np.int64)
    loc = np.empty((n, 3), dtype=np.float64)
    i = 0
    for t in tids:
        for k, l, f in tracks[t]:
            kp[i], fr[i], loc[i] = k, f, l
            i += 1
    frames = list(db._frame_ids.keys())
    f_len = np.array([len(db._frame_ids[f]) for f in frames], dtype=np.int64)
    f_tracks = np.array([t for f in frames for t in sorted(db._frame_ids[f])], dtype=np.int64)
    out = dict(track_ids=np.array(tids, dtype=np.int64), track_len=lens, point_kp=kp, point_frame=fr, point_loc=loc,
               frame_ids=np.array(frames, dtype=np.int64), frame_len=f_len, frame_tracks=f_tracks)
    for slot in (0, 1):
        items = sorted(db._last_insertions[slot].items())
        out[f"last{slot}_kp"] = np.array([a for a, _ in items], dtype=np.int64)
        out[f"last{slot}_track"] = np.array([b for _, b in items], dtype=np.int64)
    out["counters"] = np.array([_next_free(db), db._num_tracks, db._num_frames, db._max_length,
                                -1 if db._min_length == np.inf else db._min_length], dtype=np.int64)
    ratio = db._frame_id_to_inliers_ratio
    out["ratio_frames"] = np.array(list(ratio.keys()), dtype=np.int64)
    out["ratio"] = np.array([float(v) for v in ratio.values()], dtype=np.float64)
    nm = db._frame_id_to_num_matches
    out["num_matches_frames"] = np.array(list(nm.keys()), dtype=np.int64)
    out["num_matches"] = np.array([float(v) for v in nm.values()], dtype=np.float64)
    return out


def query_arrays(db):
    """Answers of the getters the consumers above the front-end call."""
    frames0 = sorted(db.get_frameIds().keys())
    # (the reference walks frames 0 .. num_frames-1 here, models/TrackDatabase.py:243-255: only asked of
    # stores whose first frame is 0)
    y = db.calculate_connectivity_data()[1] if frames0 and frames0[0] == 0 else np.zeros(0, dtype=np.int64)
    tids = list(db.get_trackIds().keys())
    probe = tids[:: max(1, len(tids) // 40)]
    frames = sorted(db.get_frameIds().keys())
    return dict(
        q_mean_track_length=np.array(db.get_mean_track_length(), dtype=np.float64),
        q_mean_frame_links=np.array(db.get_mean_frame_links(), dtype=np.float64),
        q_connectivity=np.asarray(y, dtype=np.int64),
        q_track_lengths=np.asarray(db.get_track_length_data(), dtype=np.int64),
        q_num=np.array([db.get_num_tracks(), db.get_num_frames(), db.get_max_track(),
                        -1 if db.get_min_track() == np.inf else db.get_min_track()], dtype=np.int64),
        q_probe_tracks=np.array(probe, dtype=np.int64),
        q_probe_frames=np.array([f for t in probe for f in db.get_frame_ids_for_track(t)], dtype=np.int64),
        q_probe_locs=np.array([db.get_feature_locations(db.get_frame_ids_for_track(t)[-1], t) for t in probe],
                              dtype=np.float64).reshape(-1, 3),
        q_probe_segment=np.array([p for t in probe for p in db.get_track_locations_in_segment(
            t, frames[0], frames[len(frames) // 2])], dtype=np.float64).reshape(-1, 3),
        q_window=np.array(sorted(db.get_tracks_in_bundle_window(frames[0], frames[min(5, len(frames) - 1)])),
                          dtype=np.int64),
        q_frame_tracks_sorted=np.array([t for f in frames for t in sorted(db.get_track_ids_for_frame(f))],
                                       dtype=np.int64),
    )


def assert_same(got, want, what=""):
    for k, v in want.items():
        assert k in got, f"{what}: {k} missing"
        g = got[k]
        assert g.shape == v.shape and np.array_equal(g, v), f"{what}: {k} differs"


def arrays_from_rows(first_frame, rows, loc, n_inl):
    """The flat-array form of the track store (what the device's track assignment emits), derived
    on the host by walking the rows the way models/TrackDatabase.py:34-76,171-184 does.  Valid for
    tables where no two rows of a pair share prev_left.  Returns a dict of numpy arrays."""
    rows, loc = np.asarray(rows), np.asarray(loc, dtype=np.float32)
    T = int(np.sum(n_inl))
    row_track = np.zeros(T, np.int32)
    row_depth = np.zeros(T, np.int32)
    last = {}
    n_tracks, g = 0, 0
    for p, m in enumerate(np.asarray(n_inl).tolist()):
        cur = {}
        for _ in range(m):
            a, c = int(rows[g, 0]), int(rows[g, 2])
            if a in last:
                row_track[g], row_depth[g] = last[a][0], last[a][1] + 1
            else:
                row_track[g], row_depth[g] = n_tracks, 0
                n_tracks += 1
            cur[c] = (int(row_track[g]), int(row_depth[g]))
            g += 1
        last = cur
    length = np.zeros(n_tracks, np.int64)
    np.maximum.at(length, row_track, row_depth + 2)
    track_start = np.concatenate([[0], np.cumsum(length)]).astype(np.int32)
    N = int(track_start[-1])
    pt_kp, pt_frame, pt_loc = np.zeros(N, np.int32), np.zeros(N, np.int32), np.zeros((N, 3), np.float32)
    pair_of = np.repeat(np.arange(len(n_inl)), np.asarray(n_inl))
    s = track_start[row_track] + 1 + row_depth
    pt_kp[s], pt_frame[s], pt_loc[s] = rows[:T, 2], first_frame + pair_of + 1, loc[:T, 3:]
    new = row_depth == 0
    s0 = track_start[row_track[new]]
    pt_kp[s0], pt_frame[s0], pt_loc[s0] = rows[:T, 0][new], first_frame + pair_of[new], loc[:T, :3][new]
    return dict(row_track=row_track, row_depth=row_depth, track_start=track_start, pt_kp=pt_kp, pt_frame=pt_frame,
                pt_loc=pt_loc, n_tracks=n_tracks, n_points=N, longest=int(length.max()) if n_tracks else 0)


def chain_tables(n_pairs, n_kp, seed, empty_every=7, keep=0.8, full=False):
    """Random inlier tables with real track structure and unique prev_left per pair (what the
    front-end emits): continuations, repeated cur_left, pairs without rows."""
    rng = np.random.default_rng(seed)
    rows, n_inl = [], []
    prev_cur = np.zeros(0, np.int64)
    for p in range(n_pairs):
        m = 0 if (empty_every and p % empty_every == empty_every - 2) else (n_kp // 2 if full else int(rng.integers(1, n_kp // 2)))
        prev_left = rng.permutation(n_kp)[:m]                           # unique
        cont = np.unique(prev_cur)
        take = cont[rng.random(cont.shape[0]) < keep][:m]
        others = np.setdiff1d(prev_left, take, assume_unique=False)
        prev_left = np.concatenate([take, others])[:m]
        rng.shuffle(prev_left)
        cur_left = rng.permutation(n_kp)[:m] if full else rng.integers(0, n_kp, size=m)
        if m > 6:
            cur_left[5] = cur_left[2]                                   # two rows end on one keypoint: the later one is continued
        r = np.stack([prev_left, rng.integers(0, n_kp, m), cur_left, rng.integers(0, n_kp, m)], axis=1)
        rows.append(r.astype(np.int32).reshape(-1, 4))
        n_inl.append(m)
        prev_cur = cur_left
    rows = np.concatenate(rows)
    loc = rng.uniform(0, 1226, (rows.shape[0], 6)).astype(np.float32)
    return rows, loc, np.asarray(n_inl, np.int64)
