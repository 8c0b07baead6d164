#!/usr/bin/env python3
"""Loop-closure stress of BASELINE.json configs[4]: every keyframe against all earlier keyframes
at least `--min-gap` keyframes back (the reference: keyframes every 20 frames, candidates >= 30
frames apart), consensus matching at ratio 0.6 with a 500-iteration RANSAC, all pairs in one
launch per stage.  Prints one JSON line.

    python tools/lc_stress.py --keyframes 228 --kp 8000 --hyp 500
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import pipeline                                            # noqa: E402
from slam_amd.synthetic import SyntheticSequence, read_cameras             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=228)
    ap.add_argument("--kp", type=int, default=8000)
    ap.add_argument("--hyp", type=int, default=500)
    ap.add_argument("--min-gap", type=int, default=2, help="in keyframes (30 frames at one keyframe per 20)")
    ap.add_argument("--step", type=float, default=2.0, help="metres between consecutive keyframes")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    t0 = time.perf_counter()
    seq = SyntheticSequence(a.kp, a.keyframes, step=a.step)
    t_gen = time.perf_counter() - t0
    kf = np.arange(a.keyframes)
    pairs = np.asarray([(i, j) for i in kf for j in kf if i - j >= a.min_gap], dtype=np.int64)
    k, m1, m2 = read_cameras()
    fe = pipeline.FrontEnd(k, m1, m2, ratio_thresh=0.6)
    tb = pipeline.FrameTables.from_frames(seq.frames, dev)
    ws = fe.prepare(tb, n_hyp=a.hyp, pairs=pairs)
    fe.prepare_ransac(ws)
    torch.cuda.synchronize()
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        fe.run_ransac(ws, seed=6)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    P = ws.P
    n_cons = ws.cons.count.cpu().numpy()[:P]
    found = ws.found.cpu().numpy()[:P].astype(bool)
    nL = np.diff(tb.offL)
    desc_pairs = float((nL[pairs[:, 0]] * nL[pairs[:, 1]]).sum() + (nL * np.diff(tb.offR)).sum())
    print(json.dumps({
        "keyframes": a.keyframes, "keypoints": a.kp, "keyframe_pairs": int(P), "hypotheses_per_pair": a.hyp,
        "host_generate_s": t_gen, "consensus_matching_s": min(times), "keyframe_pairs_per_s": P / min(times),
        "descriptor_pairs": desc_pairs, "descriptor_pairs_per_s": desc_pairs / min(times),
        "hbm_allocated_GB": torch.cuda.max_memory_allocated() / 1e9,
        "pairs_with_10_consensus_and_a_pose": int((found & (n_cons >= 10)).sum()),
        "mean_consensus": float(n_cons.mean()), "mean_inlier_ratio_of_valid": float(
            np.nanmean(np.where(found & (n_cons >= 10), ws.ratio.cpu().numpy()[:P], np.nan)))}))


if __name__ == "__main__":
    main()
