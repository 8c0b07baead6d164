"""Matcher: the reference's feature-matching class (models/Matcher.py:13-275) on the GPU.

Same constructor, methods, cache layout and return types; the brute-force matching
itself runs in the slam_amd HIP library:

* `HipBFMatcher.knnMatch(query, train, k)` is the `matcher=` plug point the reference
  already has (models/Matcher.py:15,29,127) -- a drop-in for
  cv2.BFMatcher(cv2.NORM_HAMMING);
* the default detector is absent on purpose: detection/description stays with whatever
  the caller plugs into `algo=` (a cv2 detector, or `PrecomputedFeatures` for keypoints
  and descriptors that already exist, e.g. synthetic sequences).

Unlike the reference, the default arguments are not evaluated at import time
(models/Matcher.py:15 calls cv2.SIFT_create() / cv2.BFMatcher() while the module loads).
"""
import numpy as np
import torch

from . import ops
from .constants import (CONSECUTIVE, FILTERED_MATCHES, FRAMES, HORIZONTAL_REPRESENTATION, LEFT, MATCHES, RIGHT)
from .types import DMatch, KeyPoint, MatchLists, keypoint_object_array, keypoints_from_array, keypoints_to_array


def _default_device():
    if not torch.cuda.is_available():
        raise ops._lib.SlamError("slam_amd needs a GPU: there is no CPU matching path")
    return torch.device("cuda", torch.cuda.current_device())


class HipBFMatcher:
    """Brute-force Hamming matcher with cv2.BFMatcher's knnMatch contract.

    knnMatch returns, per query row, a list of min(k, len(train)) DMatch objects sorted
    by distance; ties are resolved towards the lower train index exactly as cv2's scan
    does (models/Matcher.py:127,140,199,219 rely on list position == queryIdx for k=1).
    """

    def __init__(self, device=None):
        self.device = device

    def _dev(self):
        if self.device is None:
            self.device = _default_device()
        return self.device

    def pack(self, desc):
        """numpy [N,D] uint8 -> (device table [N,32|64], D); float descriptors (SIFT: [N,128]
        float32, matched with the L2 norm as cv2.BFMatcher() does) -> (device table [N,D] f32, D)."""
        if np.asarray(desc).dtype.kind == "f":
            desc = np.ascontiguousarray(desc, dtype=np.float32)
            if desc.ndim != 2 or desc.shape[1] not in (64, 128):
                raise ValueError("float descriptors must be [N, 128] or [N, 64]")
            return torch.from_numpy(desc).to(self._dev()), desc.shape[1]
        desc = np.ascontiguousarray(desc, dtype=np.uint8)
        if desc.ndim != 2:
            raise ValueError("descriptors must be a 2-D uint8 array")
        if desc.shape[1] > 64:
            raise ValueError("binary descriptors longer than 64 bytes are not supported")
        dev = self._dev()
        if desc.shape[0] == 0:
            return torch.zeros((0, 32 if desc.shape[1] <= 32 else 64), dtype=torch.uint8, device=dev), desc.shape[1]
        return ops.pack_descriptors(torch.from_numpy(desc).to(dev), desc.shape[1]), desc.shape[1]

    def knn_arrays(self, query, train, k=2, mode=0, thr=0.0, yq=None, yt=None):
        """Array form: (idx [Nq,2] int32, dist [Nq,2] int32 (float32 for float descriptors),
        keep [Nq] uint8 or None) as numpy.
        query/train may be numpy [N,D] or already-packed device tables (tensor, D)."""
        q, Dq = query if isinstance(query, tuple) else self.pack(query)
        t, Dt = train if isinstance(train, tuple) else self.pack(train)
        if Dq != Dt:
            raise ValueError(f"descriptor lengths differ: {Dq} vs {Dt}")
        nq, nt = q.shape[0], t.shape[0]
        if nq == 0:
            return (np.zeros((0, 2), np.int32), np.zeros((0, 2), np.float32 if q.dtype == torch.float32 else np.int32),
                    np.zeros(0, np.uint8) if mode else None)
        dev = self._dev()
        if q.dtype != t.dtype:
            raise ValueError("query and train descriptors must have the same type")
        if nt == 0:
            t = torch.zeros((1, q.shape[1]), dtype=q.dtype, device=dev)
        seg = torch.tensor([[0, nq, 0, nt, 0]], dtype=torch.int32, device=dev)
        tyq = torch.from_numpy(np.ascontiguousarray(yq, dtype=np.float32)).to(dev) if yq is not None else None
        tyt = torch.from_numpy(np.ascontiguousarray(yt, dtype=np.float32)).to(dev) if yt is not None else None
        if tyt is not None and tyt.numel() == 0:
            tyt = torch.zeros(1, dtype=torch.float32, device=dev)
        if q.dtype == torch.float32:
            # integer-valued descriptors (cv2's SIFT output) go to the int8 matrix cores: same result
            q8, t8 = ops.l2_pack_int8(q), ops.l2_pack_int8(t)
            if q8.usable() and t8.usable():
                idx, dist, keep = ops.l2_match_int8(q8, t8, seg, nq, nt, k=k, mode=mode, thr=thr, yq=tyq, yt=tyt,
                                                    n_rows=nq)
            else:
                idx, dist, keep = ops.l2_match(q, t, seg, nq, nt, k=k, mode=mode, thr=thr, yq=tyq, yt=tyt, n_rows=nq)
        else:
            idx, dist, keep = ops.hamming_match(q, t, seg, nq, nt, Dq, k=k, mode=mode, thr=thr, yq=tyq, yt=tyt,
                                                n_rows=nq)
        return idx.cpu().numpy(), dist.cpu().numpy(), (keep.cpu().numpy() if keep is not None else None)

    def knnMatch(self, queryDescriptors, trainDescriptors, k=2):
        if k not in (1, 2):
            raise ValueError("HipBFMatcher.knnMatch supports k = 1 or 2")
        idx, dist, _ = self.knn_arrays(queryDescriptors, trainDescriptors, k=k)
        return _dmatch_lists(idx, dist, k)


class PrecomputedFeatures:
    """`algo=` stand-in for keypoints/descriptors that already exist (synthetic sequences,
    features exported by another detector).  frames[i] has kpL/kpR [N,2] float32 and
    descL/descR [N,D] uint8.  "Images" are (side, frame index) tokens."""

    def __init__(self, frames):
        self.frames = frames

    def read_images(self, idx):
        if not 0 <= idx < len(self.frames):
            raise IndexError(f"frame {idx} is not in the feature store")
        return (LEFT, idx), (RIGHT, idx)

    def detectAndCompute(self, image, mask=None):
        side, idx = image
        f = self.frames[idx]
        kp, desc = (f.kpL, f.descL) if side == LEFT else (f.kpR, f.descR)
        return keypoints_from_array(kp), desc

    def detect(self, image, mask=None):
        return self.detectAndCompute(image)[0]

    def compute(self, image, keypoints, descriptors=None):
        return keypoints, self.detectAndCompute(image)[1]


_defaults = {"algo": None, "image_reader": None}


def set_default_features(algo=None, image_reader=None):
    """What a bare `Matcher()` uses for detection and images, process-wide: callers that construct
    the matcher themselves with no arguments (models/TrajectorySolver.py:15, utils/utils.py:240)
    are pointed at a feature source this way, e.g. set_default_features(PrecomputedFeatures(frames)).
    None restores the built-in default (cv2's SIFT and the reference's image files when cv2 is
    installed)."""
    _defaults["algo"], _defaults["image_reader"] = algo, image_reader


def _default_detector():
    if _defaults["algo"] is not None:
        return _defaults["algo"]
    try:
        import cv2
    except ImportError:
        return None                 # detect()/compute() then raise with the message of _need_algo
    return cv2.SIFT_create()


def _default_image_reader():
    """utils.utils.read_images (utils/utils.py:16-35): KITTI grey images by index through cv2."""
    if _defaults["image_reader"] is not None:
        return _defaults["image_reader"]
    if _defaults["algo"] is not None and hasattr(_defaults["algo"], "read_images"):
        return _defaults["algo"].read_images
    try:
        import cv2
    except ImportError:
        return _no_images
    import os
    from .constants import DATA_PATH, SEP

    def read(idx):
        name = '{:06d}.png'.format(idx)
        img1 = cv2.imread(DATA_PATH + f'image_0{SEP}' + name, 0)
        img2 = cv2.imread(DATA_PATH + f'image_1{SEP}' + name, 0)
        if img1 is None or img2 is None:
            raise FileNotFoundError(f"frame {idx} is not under {os.path.abspath(DATA_PATH)}")
        return img1, img2
    return read


def _no_images(idx):
    raise FileNotFoundError(
        "no image source: pass image_reader= (a function idx -> (left, right) images) or an algo= with "
        "read_images(), e.g. slam_amd.matcher.PrecomputedFeatures")


class Matcher:
    """A class for matching features between two images using a specified algorithm and matcher."""

    def __init__(self, algo=None, matcher=None, file_index=0, threshold=.6, display=HORIZONTAL_REPRESENTATION,
                 image_reader=None):
        self.cache = {}
        self.display = display
        self._file_index = file_index
        if algo is None:
            algo = _default_detector()      # cv2.SIFT_create() as models/Matcher.py:15, when cv2 is installed
        self.algo = algo
        self.matcher = matcher if matcher is not None else HipBFMatcher()
        self.threshold = threshold
        if image_reader is None:
            image_reader = getattr(algo, "read_images", None) or _default_image_reader()
        self._image_reader = image_reader
        self._img1, self._img2 = None, None
        self._filtered_matches = None
        self._img1_kp, self._img1_dsc = None, None
        self._img2_kp, self._img2_dsc = None, None
        self._matches = None
        self._packed = {}    # frame -> {LEFT: (device table, D), RIGHT: ...}, matcher-side descriptor cache
        self._stereo_keep = {}  # frame -> row-filter mask of the last k=1 stereo matching (thr it was run with)
        self._synced_frame = None
        try:
            self.read_images(idx=file_index)
        except FileNotFoundError:
            # a dataset that is not on disk must not stop construction when features are fed in later
            self.cache[file_index] = {FRAMES: (None, None), LEFT: None, RIGHT: None}

    # -- images / cache -------------------------------------------------------
    def read_images(self, idx):
        self._file_index = idx
        self._img1, self._img2 = self._image_reader(idx)
        if idx not in self.cache:
            self.cache[idx] = {FRAMES: (self._img1, self._img2), LEFT: None, RIGHT: None}
        else:
            self.cache[idx][FRAMES] = (self._img1, self._img2)

    def get_matches(self, idx=None):
        if idx is None:
            return self._matches
        return self.cache[idx][MATCHES]

    def get_filtered_matches(self, idx=None):
        if idx is None:
            return self._filtered_matches
        return self.cache[idx][FILTERED_MATCHES]

    def set_matches(self, matches, idx=None):
        self.cache[idx][MATCHES] = matches

    def get_kp(self, idx=None):
        if idx is None:
            return self._img1_kp, self._img2_kp
        return self.cache[idx][LEFT][0], self.cache[idx][RIGHT][0]

    def get_feature_location_frame(self, frameId, kp, loc):
        kpi = self.get_kp(frameId)[loc]
        x, y = kpi[kp].pt
        return x, y

    def feature_location_arrays(self, frameId):
        """([N,2] float32 left pixels, [M,2] float32 right pixels) of a cached frame: what
        get_feature_location_frame reads one keypoint at a time."""
        kpl, kpr = self.get_kp(frameId)
        return keypoints_to_array(kpl), keypoints_to_array(kpr)

    def get_dsc(self):
        return self._img1_dsc, self._img2_dsc

    def get_images(self, file_index=0):
        if file_index in self.cache:
            return self.cache[file_index][FRAMES]
        self.read_images(file_index)
        return self._img1, self._img2

    def get_matcher_cache(self):
        self._convert_cache_for_serialization()
        return self.cache

    def set_cache(self, matcher_cache):
        """Adopt a cache dict (models/Matcher.py:226-227).  Device-side copies of the previous
        cache's descriptors are dropped, and entries in the serialised form of
        get_matcher_cache (keypoints / matches as plain dicts, models/Matcher.py:262-274) are
        turned back into objects with .pt / .queryIdx so that matching and
        get_feature_location_frame work on the adopted cache."""
        self.cache = matcher_cache
        self._packed = {}
        self._stereo_keep = {}
        self._synced_frame = None
        for entry in self.cache.values():
            if not isinstance(entry, dict):
                continue
            for side in (LEFT, RIGHT):
                if entry.get(side) is not None:
                    kps, dsc = entry[side]
                    if len(kps) and isinstance(kps[0], dict):
                        entry[side] = (_as_object_array([_keypoint_from_dict(k) for k in kps]), dsc)
            for key in (MATCHES, FILTERED_MATCHES):
                ms = entry.get(key)
                if ms is not None and len(ms) and isinstance(ms[0], dict):
                    entry[key] = [[DMatch(m['queryIdx'], m['trainIdx'], m['distance'])] for m in ms]

    # -- detection (delegated to algo) ------------------------------------------
    def _need_algo(self):
        if self.algo is None:
            raise RuntimeError("Matcher has no detector: pass algo= (a cv2 detector or PrecomputedFeatures)")

    def detect(self, debug=False):
        self._need_algo()
        self._img1_kp = self.algo.detect(self._img1, None)
        self._img2_kp = self.algo.detect(self._img2, None)
        self._synced_frame = None
        if debug:
            print(f"number of kp in im1: {len(self._img1_kp)}, and in im2: {len(self._img2_kp)}")

    def compute(self, debug=False):
        self._need_algo()
        self._img1_dsc = self.algo.compute(self._img1, self._img1_kp, None)[1]
        self._img2_dsc = self.algo.compute(self._img2, self._img2_kp, None)[1]
        self._synced_frame = None
        if debug:
            print(f"first descriptor in im1: \n{self._img1_dsc[0]}")
            print(f"first descriptor in im2: \n{self._img2_dsc[0]}")

    def detect_and_compute(self, file_index=0):
        self._need_algo()
        if self._file_index != file_index or self._img1 is None:
            self.read_images(file_index)
        self._img1_kp, self._img1_dsc = self.algo.detectAndCompute(self._img1, None)
        self._img2_kp, self._img2_dsc = self.algo.detectAndCompute(self._img2, None)
        self.set_features(self._file_index, self._img1_kp, self._img1_dsc, self._img2_kp, self._img2_dsc)

    def set_features(self, file_index, kp_left, dsc_left, kp_right, dsc_right):
        """Install keypoints/descriptors of a frame (what detect_and_compute stores,
        models/Matcher.py:116-119)."""
        self._file_index = file_index
        if file_index not in self.cache:
            self.cache[file_index] = {FRAMES: (None, None), LEFT: None, RIGHT: None}
        self._img1_kp, self._img1_dsc, self._img2_kp, self._img2_dsc = kp_left, dsc_left, kp_right, dsc_right
        self.cache[file_index][LEFT] = (_as_object_array(kp_left), np.copy(dsc_left))
        self.cache[file_index][RIGHT] = (_as_object_array(kp_right), np.copy(dsc_right))
        self._packed.pop(file_index, None)
        self._stereo_keep.pop(file_index, None)
        self._synced_frame = file_index

    # -- matching ---------------------------------------------------------------
    def _table(self, frame, side):
        """Packed device descriptor table of a cached frame (uploaded once)."""
        entry = self._packed.setdefault(frame, {})
        if side not in entry:
            dsc = self.cache[frame][side][1]
            pack = getattr(self.matcher, "pack", None)
            entry[side] = pack(dsc) if pack is not None else None
        return entry[side]

    def _knn(self, q_dsc, t_dsc, k, q_table=None, t_table=None, **kw):
        """(idx, dist, keep) arrays; uses the HIP matcher's array path when the plugged
        matcher has one, otherwise falls back to the object's knnMatch output."""
        arrays = getattr(self.matcher, "knn_arrays", None)
        if arrays is not None:
            return arrays(q_table if q_table is not None else q_dsc, t_table if t_table is not None else t_dsc, k=k,
                          **kw)
        res = self.matcher.knnMatch(q_dsc, t_dsc, k=k)
        idx = np.full((len(res), 2), -1, np.int32)
        dist = np.full((len(res), 2), np.inf, np.float64)
        for i, row in enumerate(res):
            for c, m in enumerate(row[:2]):
                idx[i, c], dist[i, c] = m.trainIdx, m.distance
        found = idx >= 0
        # the norm follows the descriptor type, as it does for cv2's matchers on this path: binary
        # descriptors (uint8) are compared with a Hamming norm, whose distances are integers; anything
        # else is a float norm (cv2.BFMatcher() on SIFT), held as float32 like cv2 holds it
        binary = np.asarray(q_dsc).dtype == np.uint8 and np.asarray(t_dsc).dtype == np.uint8
        d_found = dist[found]
        if binary and np.isfinite(d_found).all() and (d_found == np.floor(d_found)).all() and (d_found < 2 ** 31).all():
            out = np.full(dist.shape, np.iinfo(np.int32).max, np.int32)
            out[found] = d_found.astype(np.int32)
        else:
            out = np.full(dist.shape, np.finfo(np.float32).max, np.float32)
            out[found] = d_found.astype(np.float32)
        return idx, out, None

    def find_matching_features(self, with_significance_test=False, debug=False, row_thresh=None):
        """Match the current frame's left descriptors against its right descriptors
        (models/Matcher.py:121-145).  row_thresh (extension): also evaluate the rectified
        row filter on the GPU for the k=1 case and remember it for find_stereo_matches."""
        fi = self._file_index
        cached = self._synced_frame == fi   # descriptors in hand are the cached (already uploaded) ones
        qt = self._table(fi, LEFT) if cached else None
        tt = self._table(fi, RIGHT) if cached else None
        if with_significance_test:
            idx, dist, _ = self._knn(self._img1_dsc, self._img2_dsc, 2, qt, tt)
            self._matches = _dmatch_lists(idx, dist, 2)
            self.apply_threshold(debug)
        else:
            kw = {}
            if row_thresh is not None and hasattr(self.matcher, "knn_arrays"):
                kw = dict(mode=2, thr=float(row_thresh), yq=keypoints_to_array(self._img1_kp)[:, 1],
                          yt=keypoints_to_array(self._img2_kp)[:, 1])
            idx, dist, keep = self._knn(self._img1_dsc, self._img2_dsc, 1, qt, tt, **kw)
            self._matches = _dmatch_lists(idx, dist, 1)
            self._filtered_matches = self._matches
            if keep is not None:
                self._stereo_keep[fi] = (float(row_thresh), idx[:, 0].copy(), keep.astype(bool))
        self.cache[fi][MATCHES] = self._matches
        self.cache[fi][FILTERED_MATCHES] = self._filtered_matches

    def apply_threshold(self, debug=False):
        """Ratio test with self.threshold on self._matches (models/Matcher.py:148-160)."""
        filtered = Matcher.apply_thresholds(self._matches, self.threshold)
        if debug:
            print(f"When applying significance test with threshold {self.threshold}, out of "
                  f"{len(self._matches)} matches {len(filtered)} matches remained")
        self._filtered_matches = filtered
        self.cache[self._file_index][FILTERED_MATCHES] = filtered

    def filter_matches(self, indices, file_index):
        """Keep only the matches at the given list positions (models/Matcher.py:162-172)."""
        if isinstance(self._matches, MatchLists):
            self._matches = self._matches.subset(indices)
        else:
            self._matches = [self._matches[ind] for ind in indices]
        self.cache[file_index][MATCHES] = self._matches

    @staticmethod
    def apply_thresholds(matches, threshold):
        """keep [m] iff m.distance < threshold * n.distance (models/Matcher.py:178-184).
        Raises if a query has fewer than two neighbours, as unpacking `for m, n in` does.
        Integer (Hamming) distances are tested on the GPU (slam_ratio_filter: the same fp64
        compare); float distances (an L2 matcher's), or match lists handled on a machine
        without a GPU, are compared here exactly as the reference's loop does."""
        n = len(matches)
        if n == 0:
            return []
        if (isinstance(matches, MatchLists) and matches.k == 2 and matches.distance.dtype.kind == "i"
                and torch.cuda.is_available()):
            if (matches.train[:, 1] < 0).any():
                raise ValueError("not enough values to unpack (expected 2, got 1)")      # the reference's `for m, n in`
            dev = _default_device()
            keep = ops.ratio_filter(torch.from_numpy(np.ascontiguousarray(matches.train, dtype=np.int32)).to(dev),
                                    torch.from_numpy(np.ascontiguousarray(matches.distance, dtype=np.int32)).to(dev),
                                    threshold).cpu().numpy()
            return matches.subset(np.flatnonzero(keep)).first_neighbours()
        idx = np.empty((n, 2), np.int32)
        dist = np.empty((n, 2), np.int32)
        exact = True
        for i, pair in enumerate(matches):
            m, nn = pair  # ValueError on a short list, like the reference
            d0, d1 = m.distance, nn.distance
            # (a non-finite distance from a plugged-in float matcher must reach the plain compare below, not int())
            if exact and 0 <= d0 < 2 ** 31 and 0 <= d1 < 2 ** 31 and d0 == int(d0) and d1 == int(d1):
                idx[i] = (m.trainIdx, nn.trainIdx)
                dist[i] = (int(d0), int(d1))
            else:
                exact = False
        if not exact or not torch.cuda.is_available():
            return [[m] for m, nn in matches if m.distance < threshold * nn.distance]
        dev = _default_device()
        keep = ops.ratio_filter(torch.from_numpy(idx).to(dev), torch.from_numpy(dist).to(dev), threshold).cpu().numpy()
        return [[matches[i][0]] for i in np.flatnonzero(keep)]

    def _match_left_to_left(self, ref, other, thresh):
        ref_dsc = self.cache[ref][LEFT][1]
        oth_dsc = self.cache[other][LEFT][1]
        if hasattr(self.matcher, "knn_arrays"):
            idx, dist, keep = self._knn(ref_dsc, oth_dsc, 2, self._table(ref, LEFT), self._table(other, LEFT), mode=1,
                                        thr=float(thresh))
            if len(idx) and (idx[:, 1] < 0).any():
                raise ValueError("not enough values to unpack (expected 2, got 1)")  # the reference's `for m, n in`
            rows = np.flatnonzero(keep)
            return MatchLists(rows, idx[rows], dist[rows], 1)
        matches = self.matcher.knnMatch(ref_dsc, oth_dsc, k=2)
        return Matcher.apply_thresholds(matches, thresh)

    def match_between_consecutive_frames(self, prev_frame_index, cur_frame_index, thresh=0.6, debug=False):
        """2-NN + ratio test between the left images of two cached frames; the filtered
        list is also cached under CONSECUTIVE (models/Matcher.py:187-204)."""
        filtered = self._match_left_to_left(prev_frame_index, cur_frame_index, thresh)
        self.cache[cur_frame_index][CONSECUTIVE] = filtered
        return filtered

    def match_between_any_frames(self, reference_frame, other_frame, thresh=0.6, debug=False):
        """As above for two arbitrary cached frames, nothing cached (models/Matcher.py:206-224)."""
        return self._match_left_to_left(reference_frame, other_frame, thresh)

    # -- serialisation helpers (models/Matcher.py:229-274) --------------------------
    def _convert_cache_for_serialization(self):
        for frame_id, entry in self.cache.items():
            for side in (LEFT, RIGHT):
                if entry.get(side) is not None:
                    kps, dsc = entry[side]
                    entry[side] = [k if isinstance(k, dict) else self.keypoint_to_dict(k) for k in kps], dsc
            if entry.get(MATCHES) is not None:
                entry[MATCHES] = [m if isinstance(m, dict) else self.dmatch_to_dict(m) for m in entry[MATCHES]]
            if entry.get(FILTERED_MATCHES) is not None:
                entry[FILTERED_MATCHES] = [m if isinstance(m, dict) else self.dmatch_to_dict(m)
                                           for m in entry[FILTERED_MATCHES]]
            entry[CONSECUTIVE] = None
        self._stereo_keep = {}
        self._synced_frame = None

    def keypoint_to_dict(self, keypoint):
        return {'pt': keypoint.pt, 'size': keypoint.size, 'angle': keypoint.angle, 'response': keypoint.response,
                'octave': keypoint.octave, 'class_id': keypoint.class_id}

    def dmatch_to_dict(self, dmatch):
        return {'queryIdx': dmatch[0].queryIdx, 'trainIdx': dmatch[0].trainIdx, 'distance': dmatch[0].distance}


def _keypoint_from_dict(d):
    x, y = d['pt']
    return KeyPoint(x, y, d.get('size', 1.0), d.get('angle', -1.0), d.get('response', 0.0), d.get('octave', 0),
                    d.get('class_id', -1))


def _as_object_array(kps):
    """np.copy(list of keypoints) as the reference stores it: a 1-D object array."""
    return keypoint_object_array(kps)


def _dmatch_lists(idx, dist, k):
    """[n,2] index / distance arrays -> per query a list of up to k DMatch (missing neighbours, index -1,
    left out): a MatchLists, which makes the lists when they are asked for."""
    return MatchLists(np.arange(idx.shape[0]), idx, dist, k)
