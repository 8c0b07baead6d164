"""`models.Matcher` for unchanged callers (models/TrajectorySolver.py:3,15; utils/utils.py:240;
models/LoopClosure.py): re-exports only.  A bare `Matcher()` uses cv2's SIFT and the reference's
image files when cv2 is installed, or whatever slam_amd.matcher.set_default_features installed."""
from slam_amd.constants import *                                                      # noqa: F401,F403
from slam_amd.matcher import (FILTERED_MATCHES, HipBFMatcher, Matcher, PrecomputedFeatures,  # noqa: F401
                              set_default_features)
