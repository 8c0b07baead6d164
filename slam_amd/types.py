"""Light stand-ins for the two cv2 value types that cross the front-end boundary.

Callers above the front-end only touch these attributes: DMatch.queryIdx / trainIdx /
imgIdx / distance (utils/utils.py:65-66,373-375; models/Matcher.py:155,182,273-274)
and KeyPoint.pt plus the serialised fields (models/Matcher.py:262-270;
models/TrackDatabase.py:39-43 through Matcher.get_feature_location_frame).
"""
import operator
from collections.abc import Sequence

import numpy as np


class DMatch:
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx=-1, trainIdx=-1, distance=float("inf"), imgIdx=0):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.imgIdx = int(imgIdx)
        self.distance = float(distance)   # cv2 stores float32; small integers are exact

    def __repr__(self):
        return f"DMatch(q={self.queryIdx}, t={self.trainIdx}, d={self.distance})"

    def __eq__(self, other):
        return (isinstance(other, DMatch) and self.queryIdx == other.queryIdx and self.trainIdx == other.trainIdx
                and self.distance == other.distance)


class MatchLists(Sequence):
    """What knnMatch returns -- per query row a list of up to k DMatch, nearest first -- held as
    the arrays the GPU produced; a row's list (and its DMatch objects) is made the first time it is
    asked for and kept, so `m[i] is m[i]`.  Behaves as the list of lists the reference's callers
    index, iterate, slice, len() and random.sample (models/Matcher.py:127-145,154,169-170,181;
    utils/utils.py:65-66,372-375; utils/plotters.py:634-645); code that has a use for the arrays
    reads .query / .train / .distance instead of walking the objects.

    query [n] row -> queryIdx; train, distance [n,2] (train < 0: no such neighbour); k = 1 or 2."""
    __slots__ = ("query", "train", "distance", "k", "_rows")

    def __init__(self, query, train, distance, k):
        self.query = np.asarray(query)
        self.train = np.asarray(train).reshape(-1, 2)
        self.distance = np.asarray(distance).reshape(-1, 2)
        self.k = int(k)
        self._rows = None

    def __len__(self):
        return self.query.shape[0]

    def _make_all(self):
        """Every row at once: the arrays go through tolist() first (Python scalars are far cheaper to
        handle one by one than numpy's)."""
        rows = self._rows
        if rows is None:
            rows = self._rows = [None] * len(self)
        ql, tl, dl = self.query.tolist(), self.train.tolist(), self.distance.tolist()
        if self.k == 1:
            for i, row in enumerate(rows):
                if row is None:
                    rows[i] = [DMatch(ql[i], tl[i][0], dl[i][0])] if tl[i][0] >= 0 else []
        else:
            for i, row in enumerate(rows):
                if row is None:
                    rows[i] = [DMatch(ql[i], tl[i][c], dl[i][c]) for c in range(self.k) if tl[i][c] >= 0]
        return rows

    def __iter__(self):
        return iter(self._make_all())

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self._make_all()[i]
        i = operator.index(i)
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("list index out of range")
        rows = self._rows
        if rows is None:
            rows = self._rows = [None] * n
        row = rows[i]
        if row is None:
            t, d, q = self.train[i], self.distance[i], int(self.query[i])
            row = rows[i] = [DMatch(q, int(t[c]), d[c].item()) for c in range(self.k) if t[c] >= 0]
        return row

    def subset(self, positions):
        """The rows at `positions`, in that order, as another MatchLists (rows already made are shared)."""
        pos = np.asarray(positions, dtype=np.int64).reshape(-1)
        out = MatchLists(self.query[pos], self.train[pos], self.distance[pos], self.k)
        if self._rows is not None:
            rows = self._rows
            out._rows = [rows[p] for p in pos.tolist()]
        return out

    def first_neighbours(self):
        """MatchLists of the same rows holding only each row's nearest neighbour: the `[m]` lists
        the ratio test keeps (models/Matcher.py:156,183), sharing the DMatch objects already made."""
        out = MatchLists(self.query, self.train, self.distance, 1)
        if self._rows is not None:
            out._rows = [None if r is None else r[:1] for r in self._rows]
        return out

    def __eq__(self, other):
        if isinstance(other, (list, tuple, MatchLists)):
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    __hash__ = None

    def __reduce__(self):
        return (list, (list(self),))          # pickles as the plain list of lists it stands for

    def __repr__(self):
        return f"MatchLists({len(self)} rows, k={self.k})"


class KeyPoint:
    """cv2.KeyPoint's fields.  Only .pt is stored per object; size / angle / response / octave /
    class_id (read by nothing on this path but the cache serialisation, models/Matcher.py:262-270)
    live in one optional tuple and otherwise answer with cv2's defaults -- a frame has thousands of
    keypoints and creating them is what a feature source pays per frame."""
    __slots__ = ("pt", "_meta")
    _DEFAULTS = (1.0, -1.0, 0.0, 0, -1)

    def __init__(self, x, y, size=1.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
        # cv2.KeyPoint.pt is a tuple of two Python floats holding float32 values
        self.pt = (float(np.float32(x)), float(np.float32(y)))
        meta = (size, angle, response, octave, class_id)
        if meta != KeyPoint._DEFAULTS:
            self._meta = meta

    def _field(self, i):
        try:
            return self._meta[i]
        except AttributeError:
            return KeyPoint._DEFAULTS[i]

    size = property(lambda self: self._field(0))
    angle = property(lambda self: self._field(1))
    response = property(lambda self: self._field(2))
    octave = property(lambda self: self._field(3))
    class_id = property(lambda self: self._field(4))

    def __getstate__(self):
        return (self.pt, getattr(self, "_meta", None))

    def __setstate__(self, state):
        self.pt = state[0]
        if state[1] is not None:
            self._meta = state[1]

    def __repr__(self):
        return f"KeyPoint{self.pt}"


class KeyPointList(list):
    """A list of KeyPoint that remembers the [N,2] float32 array it was made from, so that code
    which needs the coordinates as an array does not walk the objects again.  The array is
    trusted only while the list still has its length (callers above the front-end never edit
    keypoints in place)."""
    __slots__ = ("array",)


class KeyPointArray(np.ndarray):
    """The 1-D object array of KeyPoint the matcher cache holds (models/Matcher.py:118-119), with
    the same remembered coordinate array; views and slices of it do not inherit the coordinates."""

    def __array_finalize__(self, obj):
        self.array = None


def keypoint_object_array(kps):
    """list of KeyPoint -> 1-D object array of the same objects (np.copy of a keypoint list, as the
    reference stores it), carrying the coordinate array when the list has one."""
    out = np.empty(len(kps), dtype=object).view(KeyPointArray)
    for i, kp in enumerate(kps):
        out[i] = kp
    coords = getattr(kps, "array", None)
    out.array = coords if coords is not None and coords.shape[0] == len(kps) else None
    return out


def keypoints_from_array(xy):
    """[N,2] array -> list of KeyPoint (pt = the float32 values as Python floats, as cv2 holds them)."""
    arr = np.ascontiguousarray(np.asarray(xy, dtype=np.float32).reshape(-1, 2))
    out = KeyPointList()
    out.array = arr
    new = KeyPoint.__new__
    out.extend([new(KeyPoint) for _ in range(arr.shape[0])])
    for kp, x, y in zip(out, arr[:, 0].tolist(), arr[:, 1].tolist()):
        kp.pt = (x, y)
    return out


def keypoints_to_array(kps):
    """Sequence of objects with .pt -> [N,2] float32."""
    if isinstance(kps, np.ndarray) and kps.dtype != object:
        return np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 2)
    arr = getattr(kps, "array", None)
    if arr is not None and arr.shape[0] == len(kps):
        return arr
    return np.array([kp.pt for kp in kps], dtype=np.float32).reshape(-1, 2)
