#!/usr/bin/env python3
"""One-off measurement of the full-sequence configurations (BASELINE.json configs[2] and,
scaled to one GPU, configs[4]): the whole front-end over a KITTI-00-sized synthetic
sequence resident in HBM, one launch per stage.  Prints one JSON line.

    python tools/full_sequence.py --frames 4541 --kp 2000 --hyp 1000
    python tools/full_sequence.py --frames 4541 --kp 2000 --hyp 1000 --ransac   # hypotheses made on the device
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import pipeline, sharding                                  # noqa: E402
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4541)
    ap.add_argument("--kp", type=int, default=2000)
    ap.add_argument("--hyp", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--ransac", action="store_true", help="device-side RANSAC instead of a given hypothesis set")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    t0 = time.perf_counter()
    seq = SyntheticSequence(a.kp, a.frames)
    t_gen = time.perf_counter() - t0
    P = a.frames - 1
    k, m1, m2 = read_cameras()
    t0 = time.perf_counter()
    Rt_host = None
    if not a.ransac:
        Rt_host = np.stack([seeded_hypotheses(seq.relative_pose(p, p + 1), a.hyp, seed=6 + p) for p in range(P)])
    t_hyp = time.perf_counter() - t0
    fe = pipeline.FrontEnd(k, m1, m2)
    t0 = time.perf_counter()
    tb = pipeline.FrameTables.from_frames(seq.frames, dev)
    ws = fe.prepare(tb, n_hyp=a.hyp)
    Rt = torch.from_numpy(Rt_host).to(dev) if Rt_host is not None else None
    if a.ransac:
        fe.prepare_ransac(ws)
    torch.cuda.synchronize()
    t_up = time.perf_counter() - t0
    go = (lambda: fe.run_ransac(ws, seed=6)) if a.ransac else (lambda: fe.run(ws, Rt))
    go()
    torch.cuda.synchronize()
    times = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        go()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    shards = sharding.TableGather(ws, 1, 0, pose="pose" if a.ransac else "best_Rt").run()
    db = sharding.replay_tracks(shards)
    t_replay = time.perf_counter() - t0
    n_inl = ws.n_inl.cpu().numpy()[:P]
    print(json.dumps({
        "frames": a.frames, "keypoints": a.kp, "hypotheses_per_pair": a.hyp,
        "hypotheses": "device RANSAC (sample, P3P, stopping rule, refinement)" if a.ransac else "given seeded set",
        "mean_ransac_iterations": float(ws.iters.cpu().numpy()[:P].mean()) if a.ransac else None,
        "pose_error_vs_truth_max": (float(np.abs(ws.pose.cpu().numpy()[:P].reshape(P, 3, 4) - np.stack(
            [seq.relative_pose(p, p + 1) for p in range(P)])).max()) if a.ransac else None),
        "host_generate_s": t_gen, "host_hypotheses_s": t_hyp, "upload_and_alloc_s": t_up,
        "front_end_s": min(times), "frame_pairs_per_s": P / min(times),
        "hbm_allocated_GB": torch.cuda.max_memory_allocated() / 1e9,
        "track_db_replay_host_s": t_replay, "tracks": db.get_num_tracks(), "mean_track_length": db.get_mean_track_length(),
        "max_track": db.get_max_track(), "mean_inliers_per_pair": float(n_inl.mean()),
        "mean_consensus_per_pair": float(ws.cons.count.cpu().numpy()[:P].mean())}))


if __name__ == "__main__":
    main()
