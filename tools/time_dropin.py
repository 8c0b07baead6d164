#!/usr/bin/env python3
"""Times the reference-signature path frame by frame (Matcher + track_camera_for_many_images on the
host mirror, exact `random` stream) and prints where the time goes."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import frontend                                           # noqa: E402
from slam_amd.matcher import Matcher, PrecomputedFeatures                # noqa: E402
from slam_amd.synthetic import SyntheticSequence, read_cameras           # noqa: E402

F = int(os.environ.get("DROPIN_FRAMES", "21"))
N = int(os.environ.get("DROPIN_KP", "2000"))
seq = SyntheticSequence(N, F)
cams = read_cameras()
import random
random.seed(6)
matcher = Matcher(algo=PrecomputedFeatures(seq.frames))
frontend.track_camera_for_many_images(thresh=1, matcher=Matcher(algo=PrecomputedFeatures(seq.frames)), num_of_frames=3,
                                      cameras=cams)          # warm-up
random.seed(6)
t0 = time.perf_counter()
frontend.track_camera_for_many_images(thresh=1, matcher=Matcher(algo=PrecomputedFeatures(seq.frames)), num_of_frames=F,
                                      cameras=cams)
print(f"{F - 1} frame pairs, {N} keypoints: {1e3 * (time.perf_counter() - t0) / (F - 1):.2f} ms per pair (no profiler)")
random.seed(6)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
cam, db = frontend.track_camera_for_many_images(thresh=1, matcher=matcher, num_of_frames=F, cameras=cams)
pr.disable()
dt = time.perf_counter() - t0
print(f"under cProfile: {dt:.3f} s = {1e3 * dt / (F - 1):.1f} ms per pair, {db.get_num_tracks()} tracks")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
