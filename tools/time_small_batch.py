#!/usr/bin/env python3
"""Latency of the batched front-end with the device RANSAC on few frame pairs at a time (what a
caller that follows a live camera sees): plain launches against one replay of a recorded HIP graph.
    python tools/time_small_batch.py [keypoints]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import pipeline                                     # noqa: E402
from slam_amd.synthetic import SyntheticSequence, read_cameras    # noqa: E402


def wall(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        torch.cuda.synchronize()          # a caller waits for every result before it hands over the next frame
    return (time.perf_counter() - t0) / reps


def main():
    n_kp = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    dev = torch.device("cuda:0")
    k, m1, m2 = read_cameras()
    fe = pipeline.FrontEnd(k, m1, m2)
    for pairs in (1, 2, 8, 32, 128):
        seq = SyntheticSequence(n_kp, pairs + 1)
        tb = pipeline.FrameTables.from_frames(seq.frames, dev)
        ws = fe.prepare(tb, n_hyp=1000)
        fe.prepare_ransac(ws)
        plain = wall(lambda: fe.run_ransac(ws, 6), 50)
        graph = pipeline.capture_ransac(fe, ws, 6)
        replay = wall(graph.replay, 50)
        print(f"pairs={pairs:4d} keypoints={n_kp}: plain launches {1e3 * plain:.3f} ms, graph replay {1e3 * replay:.3f} ms "
              f"({1e6 * replay / pairs:.0f} us per pair)", flush=True)


if __name__ == "__main__":
    main()
