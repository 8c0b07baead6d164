"""`utils.utils` for unchanged callers (models/TrajectorySolver.py:6, models/LoopClosure.py:6,12,
114-132, models/BundleAdjustment.py:5,9, main.py:6): the front-end names are re-exports of
slam_amd.frontend; nothing is implemented here.

The rest of the reference's utils/utils.py -- image and ground-truth readers, pose algebra for the
GTSAM back-end, angle metrics (utils/utils.py:24-40,119-153,180-233,467-472,500-524,563-605) -- is
outside the front-end this package replaces.  Those names are looked up in the reference's own
utils/utils.py when another `utils` directory on sys.path holds it and its imports (cv2, gtsam) are
installed; otherwise asking for one raises an ImportError that says so."""
import importlib.util
import os
import sys

from slam_amd.constants import SEED, VERTICAL_REPRESENTATION                               # noqa: F401
from slam_amd.frontend import (_rectified_stereo_classifier, array_to_dict, coords_from_kps,  # noqa: F401
                               find_stereo_matches, find_supporters, get_3d_points_cloud,
                               get_matches_that_appear_in_all_4_images, least_squares, match_next_pair,
                               ransac_for_pnp, ransac_num_of_iterations, read_cameras,
                               rectificatied_stereo_pattern, rodriguez_to_mat, set_cameras, set_sequence_length,
                               solvePnP, track_camera_for_many_images)

_OUT_OF_SCOPE = ("read_images", "read_gt", "invert_Rt_transformation", "get_gtsam_calib_mat", "select_good_matches",
                 "find_all_minima", "apply_Rt_transformation", "project_point_on_image", "visualize_track",
                 "get_gt_trajectory", "isRotationMatrix", "rotationMatrixToEulerAngles",
                 "get_rotation_matrices_distances", "unit_vector", "angle_between", "MAC_OS_PATH", "WINDOWS_OS_PATH",
                 "SEP", "DATA_PATH", "WINDOWS_GT_PATH")
_original = None
_original_error = None


def _load_original():
    """The reference's own utils/utils.py from the next `utils` directory on sys.path."""
    global _original, _original_error
    if _original is not None or _original_error is not None:
        return _original
    here = os.path.abspath(__file__)
    for d in sys.path:
        cand = os.path.join(d or ".", "utils", "utils.py")
        if os.path.isfile(cand) and os.path.abspath(cand) != here:
            try:
                spec = importlib.util.spec_from_file_location("_reference_utils_utils", cand)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _original = mod
                return mod
            except ImportError as exc:             # cv2 / gtsam missing
                _original_error = f"{cand} could not be imported ({exc})"
                return None
    _original_error = "no other utils/utils.py is on sys.path"
    return None


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        mod = _load_original()
        if mod is not None and hasattr(mod, name):
            return getattr(mod, name)
        raise ImportError(f"utils.utils.{name} is not part of the front-end that slam_amd replaces (matching, "
                          f"triangulation, PnP-RANSAC); it stays the reference's own code, and {_original_error}")
    raise AttributeError(f"module 'utils.utils' has no attribute '{name}'")
