"""Unchanged callers: with compat/ ahead of everything else on sys.path, the import lines of the
reference's callers (models/TrajectorySolver.py:3,6; models/LoopClosure.py:6,12; utils/utils.py:240)
resolve to the GPU front-end, and the calls they make with no arguments work."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "compat")
REFERENCE = "/root/reference"           # present in the build container only


def _run(code, extra_path=()):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([COMPAT, ROOT, *extra_path]), PYTHONDONTWRITEBYTECODE="1")
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True,
                          cwd="/tmp", timeout=300)


def test_caller_import_lines_resolve_to_the_front_end():
    r = _run("""
        from models.Matcher import Matcher
        from utils.utils import track_camera_for_many_images, read_cameras, get_3d_points_cloud, array_to_dict
        import utils.utils
        import slam_amd.frontend, slam_amd.matcher
        assert Matcher is slam_amd.matcher.Matcher
        assert track_camera_for_many_images is slam_amd.frontend.track_camera_for_many_images
        for name in ("match_next_pair", "ransac_for_pnp", "find_supporters", "least_squares", "coords_from_kps",
                     "rectificatied_stereo_pattern", "get_matches_that_appear_in_all_4_images", "solvePnP",
                     "rodriguez_to_mat", "ransac_num_of_iterations", "find_stereo_matches"):
            assert getattr(utils.utils, name) is getattr(slam_amd.frontend, name), name
        # what is not the front-end stays the reference's own and says so
        try:
            from utils.utils import get_gt_trajectory
        except ImportError as e:
            assert "not part of the front-end" in str(e), e
        else:
            raise SystemExit("expected an ImportError")
        try:
            utils.utils.no_such_name
        except AttributeError:
            pass
        # the reference builds its matcher with no arguments (models/TrajectorySolver.py:15)
        from slam_amd.matcher import PrecomputedFeatures, set_default_features
        from slam_amd.synthetic import SyntheticSequence
        seq = SyntheticSequence(50, 2)
        set_default_features(PrecomputedFeatures(seq.frames))
        m = Matcher()
        assert m.algo is not None and m.cache[0][4] == ((0, 0), (1, 0))          # FRAMES of frame 0 were "read"
        m.detect_and_compute(1)
        assert len(m.get_kp(1)[0]) == 50
        print("ok")
    """)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "models")), reason="the reference checkout is not here")
def test_reference_modules_next_to_the_shims():
    """compat/ first, the reference checkout behind it: models.Matcher and utils.utils are the
    shims, models.Constants and models.TrackDatabase are the reference's own files."""
    r = _run("""
        import models.Matcher, models.Constants, models.TrackDatabase, utils.utils
        assert models.Matcher.__file__.startswith(%r)
        assert models.Constants.__file__.startswith(%r) and models.TrackDatabase.__file__.startswith(%r)
        from models.Constants import PATH_TO_SAVE_TRACKER_FILE, SEED
        assert SEED == 6
        try:
            utils.utils.get_gt_trajectory            # the reference's own utils/utils.py needs cv2 / gtsam
        except ImportError as e:
            assert "could not be imported" in str(e) or "not part of" in str(e)
        print("ok")
    """ % (COMPAT, REFERENCE, REFERENCE), extra_path=(REFERENCE,))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


@pytest.mark.gpu
def test_unchanged_calls_run_the_vo_loop_and_loop_closure(device):
    """The calls exactly as the reference makes them -- Matcher() and track_camera_for_many_images()
    with no arguments (models/TrajectorySolver.py:15,140) -- on precomputed features."""
    r = _run("""
        import random
        import numpy as np
        from models.Matcher import Matcher
        from utils.utils import track_camera_for_many_images
        import utils.utils
        from slam_amd import vo
        from slam_amd.matcher import PrecomputedFeatures, set_default_features
        from slam_amd.synthetic import SyntheticSequence, read_cameras
        seq = SyntheticSequence(300, 6)
        set_default_features(PrecomputedFeatures(seq.frames))
        utils.utils.set_sequence_length(len(seq.frames))
        utils.utils.set_cameras(*read_cameras())
        random.seed(6)
        cam, db = track_camera_for_many_images()
        state = random.getstate()
        random.seed(6)
        cam_b, db_b = vo.track_camera_batched(seq.frames, read_cameras(), thresh=1)
        assert random.getstate() == state and np.array_equal(cam, cam_b) and db == db_b
        assert db.get_num_frames() == 6 and db.get_num_tracks() > 50
        # loop closure's consensus matching, written as models/LoopClosure.py:112-139 writes it
        K, M1, M2 = utils.utils.read_cameras()
        matcher = Matcher()
        matcher.read_images(0)
        prev_map = utils.utils.match_next_pair(0, matcher)
        _, prev_3d = utils.utils.get_3d_points_cloud(prev_map, K, M1, M2, matcher, file_index=0, debug=False)
        cur_map = utils.utils.match_next_pair(3, matcher)
        consecutive = matcher.match_between_any_frames(0, 3, thresh=0.6)
        cons, _ = utils.utils.get_matches_that_appear_in_all_4_images(consecutive, utils.utils.array_to_dict(prev_map),
                                                                     utils.utils.array_to_dict(cur_map), prev_3d)
        kp1, kp2 = matcher.get_kp(idx=3)
        Rt, ratio, sup = utils.utils.ransac_for_pnp(cons, K, kp1, kp2, M2, thresh=2, debug=False, max_iterations=500,
                                                    return_supporters=True)
        assert len(cons) >= 10 and ratio > 0.8 and np.abs(Rt - seq.relative_pose(0, 3)).max() < 0.05
        print("ok")
    """)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]
