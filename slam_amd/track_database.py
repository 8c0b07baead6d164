"""Track store fed by the front-end: the consumer side of the hot path.

Same observable behaviour and pickle schema as the reference's track database
(models/TrackDatabase.py:11-304), which stays host-side Python: tracks are lists of
(left keypoint index, (xl, xr, yl), frameId) points, track ids are handed out in
arrival order, and a track is continued when the earlier-frame left keypoint of a new
inlier row is the one the track ended on in the previous pair (a two-slot table keyed
by frame parity, models/TrackDatabase.py:171-184).

Besides the reference's per-row entry point (`update_tracks_from_frame`, which asks a
Matcher for pixel locations) there is `update_tracks_from_tables`, which consumes the
inlier tables the GPU pipeline emits (rows [M,4] int32 + locations [M,6] float32).
"""
import importlib.util
import os
import pickle
import random

import numpy as np


def _load_native():
    """The C helper built next to the GPU library (slam_amd/lib/_trackdb_native*.so); None when
    it has not been built: the Python loop below does the same work, only slower."""
    libdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
    try:
        names = [n for n in os.listdir(libdir) if n.startswith("_trackdb_native") and n.endswith(".so")]
    except OSError:
        return None
    for n in names:
        try:
            spec = importlib.util.spec_from_file_location("_trackdb_native", os.path.join(libdir, n))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod
        except ImportError:
            continue
    return None


_native = _load_native()

SUCCESS, FAILURE = 0, 1
LEFT, RIGHT = 0, 1


class TrackDatabase:
    def __init__(self, path_to_pkl_file=None):
        self._tracks = {}
        self._frame_ids = {}
        self._next_free_trackId = 0
        self._last_insertions = {1: {}, 0: {}}
        self._num_tracks = 0
        self._num_frames = 0
        self._mean_track_length = 0
        self._max_length = 0
        self._min_length = np.inf
        self._mean_frame_links = 0
        self._frame_id_to_inliers_ratio = {}
        self._frame_id_to_num_matches = {}
        self.camera_positions = None
        self.ex_matrices = None
        self.matcher_cache = dict()
        if path_to_pkl_file:
            self.deserialize(path_to_pkl_file)

    # -- building ----------------------------------------------------------
    def prepare_to_next_pair(self, frameId):
        self._frame_ids.setdefault(frameId, set())
        self._frame_ids[frameId + 1] = set()
        self._last_insertions[(frameId + 1) % 2] = {}

    def get_kp_trackId(self, kp, frameId):
        return self._last_insertions[frameId % 2].get(kp)

    def generate_new_track_id(self):
        tid = self._next_free_trackId
        self._next_free_trackId += 1
        return tid

    def add_to_track(self, trackId, frameId, feature_location_prev, feature_location_cur, kp_prev, kp_cur):
        track = self._tracks.get(trackId)
        if track is None:
            track = self._tracks[trackId] = [(kp_prev, feature_location_prev, frameId)]
            self._frame_ids[frameId].add(trackId)
            self._num_tracks += 1
        track.append((kp_cur, feature_location_cur, frameId + 1))
        self._frame_ids[frameId + 1].add(trackId)
        self._last_insertions[(frameId + 1) % 2][kp_cur] = trackId
        self._num_frames = len(self._frame_ids)
        self._max_length = max(self._max_length, len(track))
        self._min_length = min(self._min_length, len(track))

    def update_tracks_from_frame(self, matcher, frameId, matches_kp_indices_and_3d_points):
        """Rows are (prev_left, prev_right, cur_left, cur_right, X); pixel locations come
        from matcher.get_feature_location_frame (models/TrackDatabase.py:34-53)."""
        arrays = getattr(matcher, "feature_location_arrays", None)
        if arrays is not None and len(matches_kp_indices_and_3d_points) > 0:
            # the same rows through the table form: pixel locations gathered as arrays
            rows = np.array([m[:4] for m in matches_kp_indices_and_3d_points], dtype=np.int64).reshape(-1, 4)
            (pl, pr), (cl, cr) = arrays(frameId), arrays(frameId + 1)
            loc = np.stack([pl[rows[:, 0], 0], pr[rows[:, 1], 0], pl[rows[:, 0], 1],
                            cl[rows[:, 2], 0], cr[rows[:, 3], 0], cl[rows[:, 2], 1]], axis=1)
            self.update_tracks_from_pair_tables(frameId, rows, loc, [rows.shape[0]])
            return
        self.prepare_to_next_pair(frameId)
        for prev_left, prev_right, cur_left, cur_right, _ in matches_kp_indices_and_3d_points:
            xl, yl = matcher.get_feature_location_frame(frameId, kp=prev_left, loc=LEFT)
            xr, _ = matcher.get_feature_location_frame(frameId, kp=prev_right, loc=RIGHT)
            xl_n, yl_n = matcher.get_feature_location_frame(frameId + 1, kp=cur_left, loc=LEFT)
            xr_n, _ = matcher.get_feature_location_frame(frameId + 1, kp=cur_right, loc=RIGHT)
            trackId = self.get_kp_trackId(prev_left, frameId)
            if trackId is None:
                trackId = self.generate_new_track_id()
            self.add_to_track(trackId, frameId, (xl, xr, yl), (xl_n, xr_n, yl_n), prev_left, cur_left)

    def update_tracks_from_tables(self, frameId, rows, loc):
        """rows [M,4] int (prev_left, prev_right, cur_left, cur_right); loc [M,6] float32
        (xl, xr, yl of the earlier frame, then of the later one), in consensus order.
        Same state as update_tracks_from_frame row by row; the per-row bookkeeping of
        add_to_track is done on local names, the counters once per pair."""
        self.prepare_to_next_pair(frameId)
        rows = np.asarray(rows)
        if rows.shape[0] == 0:
            return
        prev_left = rows[:, 0].tolist()
        cur_left = rows[:, 2].tolist()
        loc = np.asarray(loc, dtype=np.float64).tolist()      # float32 -> Python floats, exactly
        tracks = self._tracks
        last_prev = self._last_insertions[frameId % 2]
        last_cur = self._last_insertions[(frameId + 1) % 2]
        in_prev = self._frame_ids[frameId]
        in_cur = self._frame_ids[frameId + 1]
        nxt = frameId + 1
        max_len, min_len = self._max_length, self._min_length
        for kp_prev, kp_cur, l in zip(prev_left, cur_left, loc):
            trackId = last_prev.get(kp_prev)
            if trackId is None:
                trackId = self._next_free_trackId
                self._next_free_trackId += 1
            track = tracks.get(trackId)
            if track is None:
                track = tracks[trackId] = [(kp_prev, (l[0], l[1], l[2]), frameId)]
                in_prev.add(trackId)
                self._num_tracks += 1
            track.append((kp_cur, (l[3], l[4], l[5]), nxt))
            in_cur.add(trackId)
            last_cur[kp_cur] = trackId
            n = len(track)
            if n > max_len:
                max_len = n
            if n < min_len:
                min_len = n
        self._max_length, self._min_length = max_len, min_len
        self._num_frames = len(self._frame_ids)

    def update_tracks_from_pair_tables(self, first_frame, rows, loc, n_inl):
        """The tables of consecutive frame pairs (first_frame, first_frame+1), ... in one call:
        n_inl[p] rows of `rows`/`loc` belong to pair p, in frame order.  Same state as calling
        update_tracks_from_tables pair by pair; done by the C helper when it is built."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        loc = np.ascontiguousarray(loc, dtype=np.float32)
        n_inl = np.ascontiguousarray(n_inl, dtype=np.int64)
        if _native is None:
            off = 0
            for p, m in enumerate(n_inl.tolist()):
                self.update_tracks_from_tables(first_frame + p, rows[off:off + m], loc[off:off + m])
                off += m
            return
        min_len = -1 if self._min_length == np.inf else int(self._min_length)
        nf, nt, mx, mn, fr = _native.replay_pairs(self._tracks, self._frame_ids, self._last_insertions,
                                                  int(first_frame), rows, loc, n_inl, int(self._next_free_trackId),
                                                  int(self._num_tracks), int(self._max_length), min_len,
                                                  int(self._num_frames))
        self._next_free_trackId, self._num_tracks, self._max_length, self._num_frames = nf, nt, mx, fr
        self._min_length = np.inf if mn < 0 else mn

    def add_inliers_ratio(self, frame_id, ratio):
        assert frame_id in self._frame_ids
        self._frame_id_to_inliers_ratio[frame_id] = ratio

    def add_num_matches(self, frame_id, num_matches):
        assert frame_id in self._frame_ids
        self._frame_id_to_num_matches[frame_id] = num_matches

    # -- queries -----------------------------------------------------------
    def get_inliers_ratio_per_frame(self):
        return self._frame_id_to_inliers_ratio

    def get_num_matches_per_frame(self):
        return self._frame_id_to_num_matches

    def get_track_ids_for_frame(self, frameId):
        return list(self._frame_ids.get(frameId, ()))

    def get_frame_ids_for_track(self, trackId):
        return [p[2] for p in self._tracks.get(trackId, ())]

    def get_feature_locations(self, frameId, trackId):
        for p in self._tracks.get(trackId, ()):
            if p[-1] == frameId:
                return p[1]
        return None

    def get_num_tracks(self):
        return self._num_tracks

    def get_num_frames(self):
        return self._num_frames

    def get_mean_track_length(self):
        if not self._num_tracks:
            return 0
        self._mean_track_length = sum(len(t) for t in self._tracks.values()) / self._num_tracks
        return self._mean_track_length

    def get_max_track(self):
        return self._max_length

    def get_min_track(self):
        return self._min_length

    def get_mean_frame_links(self):
        if not self._num_frames:
            return 0
        self._mean_frame_links = sum(len(s) for s in self._frame_ids.values()) / self._num_frames
        return self._mean_frame_links

    def get_random_track_of_length(self, length):
        pool = [(tid, t) for tid, t in self._tracks.items() if len(t) >= length]
        return random.choice(pool) if pool else None

    def get_all_tracks_of_length(self, length):
        return [(tid, t) for tid, t in self._tracks.items() if len(t) == length]

    def calculate_connectivity_data(self):
        x = np.arange(self._num_frames)
        y = np.zeros_like(x)
        for i in range(len(y) - 1):
            y[i] = len(self._frame_ids[i] & self._frame_ids[i + 1])
        return x, y

    def get_track_length_data(self):
        return np.array([len(t) for t in self._tracks.values()])

    def set_ex_camera_positions(self, camera_positions):
        self.camera_positions = camera_positions

    def set_ex_matrices(self, ex_matrices):
        self.ex_matrices = ex_matrices

    def get_camera_position_by_frameId(self, frameId):
        return self.camera_positions[frameId]

    def get_extrinsic_matrix_by_frameId(self, frameId):
        return self.ex_matrices[frameId]

    def get_extrinsic_matrices(self):
        return self.ex_matrices

    def get_trackIds(self):
        return self._tracks

    def get_frameIds(self):
        return self._frame_ids

    def get_track_locations_in_segment(self, track_id, frame_id_start, frame_id_stop):
        return [pt for _, pt, fid in self._tracks[track_id] if frame_id_start < fid <= frame_id_stop + 1]

    def get_tracks_in_bundle_window(self, start_frame_id, end_frame_id):
        out = set()
        for fid in range(start_frame_id, end_frame_id):
            out.update(self.get_track_ids_for_frame(fid))
        return out

    def get_all_points_in_track(self, track_id):
        return [pt for _, pt, _ in self._tracks[track_id]]

    def get_track_data(self, track_id):
        return self._tracks[track_id]

    def set_matcher(self, matcher_cache):
        self.matcher_cache = matcher_cache

    def get_matcher_cache(self):
        return self.matcher_cache

    # -- persistence (key names of models/TrackDatabase.py:121-137) ---------
    def _state(self):
        return {
            "tracks": self._tracks, "frame_ids": self._frame_ids, "next_free_trackId": self._next_free_trackId,
            "last_insertions": self._last_insertions, "num_tracks": self._num_tracks, "num_frames": self._num_frames,
            "mean_track_length": self._mean_track_length, "max_length": self._max_length,
            "min_length": self._min_length, "mean_frame_links": self._mean_frame_links,
            "frame_id_to_inliers_ratio": self._frame_id_to_inliers_ratio, "ex_camera_positions": self.camera_positions,
            "ex_matrices": self.ex_matrices, "frame_id_to_num_matches": self._frame_id_to_num_matches,
        }

    def serialize(self, file_path):
        with open(file_path, "wb") as f:
            pickle.dump(self._state(), f)

    def deserialize(self, file_path):
        try:
            with open(file_path, "rb") as f:
                d = pickle.load(f)
        except FileNotFoundError:
            return FAILURE
        self._tracks, self._frame_ids = d["tracks"], d["frame_ids"]
        self._next_free_trackId, self._last_insertions = d["next_free_trackId"], d["last_insertions"]
        self._num_tracks, self._num_frames = d["num_tracks"], d["num_frames"]
        self._mean_track_length, self._max_length = d["mean_track_length"], d["max_length"]
        self._min_length, self._mean_frame_links = d["min_length"], d["mean_frame_links"]
        self._frame_id_to_inliers_ratio = d["frame_id_to_inliers_ratio"]
        self.camera_positions, self.ex_matrices = d["ex_camera_positions"], d["ex_matrices"]
        self._frame_id_to_num_matches = d.get("frame_id_to_num_matches", self._frame_id_to_num_matches)
        return SUCCESS

    def __eq__(self, other):
        if not isinstance(other, TrackDatabase):
            return NotImplemented
        a, b = self._state(), other._state()
        for k in a:
            if k in ("ex_camera_positions", "ex_matrices"):
                if (a[k] is None) != (b[k] is None) or (a[k] is not None and not np.array_equal(a[k], b[k])):
                    return False
            elif a[k] != b[k]:
                return False
        return True


# ---------------------------------------------------------------------------------------------
class _LazyMap:
    """Read-only mapping whose values are made on demand (dict-like enough for the consumers of
    get_trackIds / get_frameIds: len, iteration, indexing, membership, keys/values/items, ==)."""

    def __init__(self, keys, make):
        self._keys, self._make = keys, make

    def __len__(self):
        return len(self._keys)

    def __iter__(self):
        return iter(self._keys)

    def __contains__(self, k):
        return k in self._keys

    def __getitem__(self, k):
        if k not in self._keys:
            raise KeyError(k)
        return self._make(k)

    def get(self, k, default=None):
        return self._make(k) if k in self._keys else default

    def keys(self):
        return self._keys

    def values(self):
        return (self._make(k) for k in self._keys)

    def items(self):
        return ((k, self._make(k)) for k in self._keys)

    def __eq__(self, other):
        try:
            return len(self) == len(other) and all(k in other and other[k] == self._make(k) for k in self._keys)
        except TypeError:
            return NotImplemented


def _lazy_container(name):
    slot = "_store" + name

    def get(self):
        if self._arrays is not None:
            self._materialize()
        return self.__dict__[slot]

    def put(self, value):
        self.__dict__[slot] = value

    return property(get, put)


class ArrayTrackDatabase(TrackDatabase):
    """The track store answered from flat arrays (what slam_track_assign produces on the GPU)
    instead of a dict of lists of tuples: same getters, same answers, same pickle -- but nothing is
    built per track point until somebody asks for it.

    Arrays (host numpy): n_inl [P] inlier rows per pair; row_track, row_depth [T] track id and
    position-in-track minus one of every inlier row; track_start [n_tracks + 1] and the point table
    pt_kp, pt_frame [N], pt_loc [N,3] float32 laid out track after track, points in frame order.
    Pair p is frames (first_frame + p, first_frame + p + 1).

    Reading methods work on the arrays.  The dict-of-lists containers of the reference's schema
    (models/TrackDatabase.py:17-20) are built on first use of `_tracks`, `_frame_ids` or
    `_last_insertions` -- serialize(), ==, or any method that adds rows -- and from then on the
    object behaves exactly like the plain TrackDatabase it has become."""

    _tracks = _lazy_container("_tracks")
    _frame_ids = _lazy_container("_frame_ids")
    _last_insertions = _lazy_container("_last_insertions")

    def __init__(self, first_frame, n_inl, row_track, row_depth, track_start, pt_kp, pt_frame, pt_loc, max_length=None):
        self._arrays = None
        super().__init__()
        n_inl = np.ascontiguousarray(n_inl, dtype=np.int64)
        T = int(n_inl.sum())
        n_tracks = int(np.count_nonzero(np.asarray(row_depth[:T]) == 0))
        a = dict(first=int(first_frame), P=int(n_inl.shape[0]), n_inl=n_inl,
                 off=np.concatenate([[0], np.cumsum(n_inl)]).astype(np.int64),
                 row_track=np.ascontiguousarray(row_track[:T], dtype=np.int32),
                 row_depth=np.ascontiguousarray(row_depth[:T], dtype=np.int32),
                 track_start=np.ascontiguousarray(track_start[:n_tracks + 1], dtype=np.int32))
        N = int(a["track_start"][n_tracks]) if n_tracks else 0
        a["pt_kp"] = np.ascontiguousarray(pt_kp[:N], dtype=np.int32)
        a["pt_frame"] = np.ascontiguousarray(pt_frame[:N], dtype=np.int32)
        a["pt_loc"] = np.ascontiguousarray(pt_loc[:N], dtype=np.float32).reshape(N, 3)
        a["n_tracks"], a["T"] = n_tracks, T
        self._track_cache = {}
        self._next_free_trackId = n_tracks
        self._num_tracks = n_tracks
        lengths = np.diff(a["track_start"]) if n_tracks else np.zeros(0, np.int64)
        self._max_length = int(lengths.max() if max_length is None and n_tracks else (max_length or 0))
        self._min_length = 2 if T else np.inf       # every track starts with two points (models/TrackDatabase.py:67,76)
        with_rows = np.flatnonzero(n_inl)
        # the reference refreshes the frame count whenever a row is added: the frames registered
        # up to the last pair that had rows (models/TrackDatabase.py:74)
        self._num_frames = int(with_rows[-1]) + 2 if with_rows.size else 0
        self._arrays = a

    # -- switching to the dict form ------------------------------------------------------------
    def _materialize(self):
        a, self._arrays = self._arrays, None
        n = a["n_tracks"]
        if _native is not None:
            tracks = _native.build_tracks(a["pt_kp"], a["pt_frame"], a["pt_loc"], a["track_start"], n)
        else:
            kp, fr, loc, st = a["pt_kp"].tolist(), a["pt_frame"].tolist(), a["pt_loc"].astype(np.float64).tolist(), \
                a["track_start"].tolist()
            tracks = {t: [(kp[i], tuple(loc[i]), fr[i]) for i in range(st[t], st[t + 1])] for t in range(n)}
        first, P, off = a["first"], a["P"], a["off"]
        frame_ids = {}
        for f in range(P + 1 if P else 0):
            frame_ids[first + f] = self._frame_set(a, f)
        last = {1: {}, 0: {}}
        for q in (P - 2, P - 1):
            if q < 0:
                continue
            rows = slice(int(off[q]), int(off[q + 1]))
            tid = a["row_track"][rows]
            cur_left = a["pt_kp"][a["track_start"][tid] + 1 + a["row_depth"][rows]]
            last[(first + q + 1) % 2] = dict(zip(cur_left.tolist(), tid.tolist()))     # later rows overwrite earlier
        self._tracks, self._frame_ids, self._last_insertions = tracks, frame_ids, last
        self._track_cache = {}

    @staticmethod
    def _frame_set(a, f):
        """Tracks with a point in the f-th frame of the run: the rows of the pair that ends there,
        and the tracks the pair that starts there opens."""
        off, out = a["off"], set()
        if f >= 1:
            out.update(a["row_track"][int(off[f - 1]):int(off[f])].tolist())
        if f < a["P"]:
            s = slice(int(off[f]), int(off[f + 1]))
            out.update(a["row_track"][s][a["row_depth"][s] == 0].tolist())
        return out

    def _mutating(name):                                     # noqa: N805 -- class-body helper
        base = getattr(TrackDatabase, name)

        def method(self, *args, **kw):
            if self._arrays is not None:
                self._materialize()
            return base(self, *args, **kw)
        method.__name__ = name
        method.__doc__ = base.__doc__
        return method

    for _n in ("prepare_to_next_pair", "get_kp_trackId", "generate_new_track_id", "add_to_track",
               "update_tracks_from_frame", "update_tracks_from_tables", "update_tracks_from_pair_tables", "deserialize"):
        locals()[_n] = _mutating(_n)
    del _n, _mutating

    # -- per-frame records -----------------------------------------------------------------------
    def _has_frame(self, frame_id):
        a = self._arrays
        return a["P"] > 0 and a["first"] <= frame_id <= a["first"] + a["P"]

    def add_inliers_ratio(self, frame_id, ratio):
        if self._arrays is None:
            return super().add_inliers_ratio(frame_id, ratio)
        assert self._has_frame(frame_id)
        self._frame_id_to_inliers_ratio[frame_id] = ratio

    def add_num_matches(self, frame_id, num_matches):
        if self._arrays is None:
            return super().add_num_matches(frame_id, num_matches)
        assert self._has_frame(frame_id)
        self._frame_id_to_num_matches[frame_id] = num_matches

    def set_frame_records(self, ratios, num_matches):
        """add_inliers_ratio / add_num_matches for every pair of the run at once (utils/utils.py:281-282)."""
        a = self._arrays
        frames = range(a["first"], a["first"] + a["P"])
        self._frame_id_to_inliers_ratio.update(zip(frames, np.asarray(ratios, dtype=np.float64).tolist()))
        # np.float64 values: the reference stores np.sum of find_supporters' float64 0/1 array (utils/utils.py:279,282,390)
        self._frame_id_to_num_matches.update(zip(frames, np.asarray(num_matches, dtype=np.float64)))

    # -- queries from the arrays -----------------------------------------------------------------
    def get_track_data(self, track_id):
        a = self._arrays
        if a is None:
            return super().get_track_data(track_id)
        track = self._track_cache.get(track_id)
        if track is None:
            if not 0 <= track_id < a["n_tracks"]:
                raise KeyError(track_id)
            s0, s1 = int(a["track_start"][track_id]), int(a["track_start"][track_id + 1])
            if _native is not None:
                track = _native.track_list(a["pt_kp"], a["pt_frame"], a["pt_loc"], s0, s1)
            else:
                s = slice(s0, s1)
                track = [(k, tuple(l), f) for k, l, f in zip(a["pt_kp"][s].tolist(), a["pt_loc"][s].astype(np.float64).tolist(),
                                                             a["pt_frame"][s].tolist())]
            self._track_cache[track_id] = track
        return track

    def get_track_ids_for_frame(self, frameId):
        a = self._arrays
        if a is None:
            return super().get_track_ids_for_frame(frameId)
        return list(self._frame_set(a, frameId - a["first"])) if self._has_frame(frameId) else []

    def get_frame_ids_for_track(self, trackId):
        a = self._arrays
        if a is None:
            return super().get_frame_ids_for_track(trackId)
        if not 0 <= trackId < a["n_tracks"]:
            return []
        return a["pt_frame"][int(a["track_start"][trackId]):int(a["track_start"][trackId + 1])].tolist()

    def get_feature_locations(self, frameId, trackId):
        a = self._arrays
        if a is None:
            return super().get_feature_locations(frameId, trackId)
        if not 0 <= trackId < a["n_tracks"]:
            return None
        s0, s1 = int(a["track_start"][trackId]), int(a["track_start"][trackId + 1])
        hit = np.flatnonzero(a["pt_frame"][s0:s1] == frameId)
        return tuple(a["pt_loc"][s0 + int(hit[0])].astype(np.float64).tolist()) if hit.size else None

    def get_mean_track_length(self):
        a = self._arrays
        if a is None:
            return super().get_mean_track_length()
        if not self._num_tracks:
            return 0
        self._mean_track_length = int(a["track_start"][a["n_tracks"]]) / self._num_tracks
        return self._mean_track_length

    def get_mean_frame_links(self):
        a = self._arrays
        if a is None:
            return super().get_mean_frame_links()
        if not self._num_frames:
            return 0
        # a frame holds the rows of the pair ending there plus the tracks opened by the pair starting there
        links = a["T"] + a["n_tracks"]
        self._mean_frame_links = links / self._num_frames
        return self._mean_frame_links

    def get_random_track_of_length(self, length):
        a = self._arrays
        if a is None:
            return super().get_random_track_of_length(length)
        pool = np.flatnonzero(np.diff(a["track_start"]) >= length).tolist()
        if not pool:
            return None
        tid = random.choice(pool)                  # one draw from a pool of the same size and order as the reference's
        return tid, self.get_track_data(tid)

    def get_all_tracks_of_length(self, length):
        a = self._arrays
        if a is None:
            return super().get_all_tracks_of_length(length)
        return [(t, self.get_track_data(t)) for t in np.flatnonzero(np.diff(a["track_start"]) == length).tolist()]

    def calculate_connectivity_data(self):
        a = self._arrays
        if a is None:
            return super().calculate_connectivity_data()
        if a["first"] != 0:
            raise KeyError(0)                      # the reference walks frames 0 .. num_frames-1 (models/TrackDatabase.py:247-249)
        x = np.arange(self._num_frames)
        y = np.zeros_like(x)
        m = max(len(y) - 1, 0)
        y[:m] = a["n_inl"][:m]                    # tracks present in frames i and i+1 = rows of pair i
        return x, y

    def get_track_length_data(self):
        a = self._arrays
        if a is None:
            return super().get_track_length_data()
        return np.diff(a["track_start"]).astype(np.int64)

    def get_trackIds(self):
        a = self._arrays
        if a is None:
            return super().get_trackIds()
        return _LazyMap(range(a["n_tracks"]), self.get_track_data)

    def get_frameIds(self):
        a = self._arrays
        if a is None:
            return super().get_frameIds()
        frames = range(a["first"], a["first"] + a["P"] + 1) if a["P"] else range(0)
        return _LazyMap(frames, lambda f: self._frame_set(a, f - a["first"]))

    def get_track_locations_in_segment(self, track_id, frame_id_start, frame_id_stop):
        return [pt for _, pt, fid in self.get_track_data(track_id) if frame_id_start < fid <= frame_id_stop + 1]

    def get_all_points_in_track(self, track_id):
        return [pt for _, pt, _ in self.get_track_data(track_id)]
