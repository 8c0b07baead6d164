"""Light stand-ins for the two cv2 value types that cross the front-end boundary.

Callers above the front-end only touch these attributes: DMatch.queryIdx / trainIdx /
imgIdx / distance (utils/utils.py:65-66,373-375; models/Matcher.py:155,182,273-274)
and KeyPoint.pt plus the serialised fields (models/Matcher.py:262-270;
models/TrackDatabase.py:39-43 through Matcher.get_feature_location_frame).
"""
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


class KeyPoint:
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, x, y, size=1.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
        # cv2.KeyPoint.pt is a tuple of two Python floats holding float32 values
        self.pt = (float(np.float32(x)), float(np.float32(y)))
        self.size, self.angle, self.response, self.octave, self.class_id = size, angle, response, octave, class_id

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
    for x, y in arr.tolist():
        kp = new(KeyPoint)
        kp.pt = (x, y)
        kp.size, kp.angle, kp.response, kp.octave, kp.class_id = 1.0, -1.0, 0.0, 0, -1
        out.append(kp)
    return out


def keypoints_to_array(kps):
    """Sequence of objects with .pt -> [N,2] float32."""
    if isinstance(kps, np.ndarray) and kps.dtype != object:
        return np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 2)
    arr = getattr(kps, "array", None)
    if arr is not None and arr.shape[0] == len(kps):
        return arr
    return np.array([kp.pt for kp in kps], dtype=np.float32).reshape(-1, 2)
