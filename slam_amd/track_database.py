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
