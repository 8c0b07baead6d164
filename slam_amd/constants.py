"""Constants shared with callers of the front-end.

The matcher-cache keys and status codes are the integers the reference's callers use
(models/Constants.py:4-59): LEFT/RIGHT index the per-frame (keypoints, descriptors)
entries and select an image in get_feature_location_frame, MATCHES / CONSECUTIVE /
FRAMES and the string "FILTERED_MATCHES" (models/Matcher.py:10) key the cache dict.
"""
import os

KPS = 0
DSC = 2
HORIZONTAL_REPRESENTATION = 0
VERTICAL_REPRESENTATION = 1
MATCHES = 2
LEFT = 0
RIGHT = 1
CONSECUTIVE = 3
FRAMES = 4
FILTERED_MATCHES = "FILTERED_MATCHES"
SUCCESS = 0
FAILURE = 1
SEED = 6

# thresholds the visual-odometry loop and loop closure pass (utils/utils.py:236,267,274-275,313;
# models/LoopClosure.py:112,129,132-133)
STEREO_ROW_THRESH = 2
VO_RATIO_THRESH = 1
LC_RATIO_THRESH = 0.6
PIXEL_THRESH = 2
VO_MAX_ITERATIONS = 1000
LC_MAX_ITERATIONS = 500
LC_MIN_CONSENSUS = 10

# dataset location the reference hard-codes (utils/utils.py:16-21)
SEP = "\\" if os.name == 'nt' else "/"
DATA_PATH = "VAN_ex/dataset/sequences/05\\" if os.name == 'nt' else "VAN_ex/dataset/sequences/05/"
GT_PATH = "VAN_ex/dataset/poses/05.txt"
