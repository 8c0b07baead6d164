#!/usr/bin/env python3
"""Where does the dead-reckoning drift of the chained visual-odometry poses come from?

Runs the same synthetic sequence through the device RANSAC (ransac.hip: per-pair generators, P3P,
LM refinement on the left image) and through the host solver (pnp.py with the exact `random` stream),
and compares every relative pose with the ground truth: mean (= bias) and spread of the six pose
components per frame pair, and the end-point drift of the chained trajectory, for

  noise 0.3 px (the benchmark's data)      -- what the benchmark sequence shows
  noise 0                                   -- a solver bias would stay, a data bias goes
  refinement on left + right residuals      -- is the left-image-only cost (utils/utils.py:452) the source?
  ground-truth 3-D points, noisy pixels     -- is it the triangulated depth (utils/utils.py:527-560)?

    python tools/drift_study.py --frames 600 --kp 2000 > gpurun_out/r3/drift.json
"""
import argparse
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rotvec(R):
    c = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    th = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2.0
    return w if th < 1e-9 else w * th / np.sin(th)


def pose_errors(rel, gt):
    """Per pair: rotation error as a rotation vector of R_est R_gt^T, translation error t_est - t_gt."""
    dr = np.array([rotvec(a[:, :3] @ b[:, :3].T) for a, b in zip(rel, gt)])
    dt = np.array([a[:, 3] - b[:, 3] for a, b in zip(rel, gt)])
    return dr, dt


def summarize(name, rel, gt, m1, step):
    from slam_amd.sharding import chain_poses
    dr, dt = pose_errors(rel, gt)
    n = len(rel)
    cam = chain_poses(np.asarray(rel), first=m1)[1]
    cam_gt = chain_poses(np.asarray(gt), first=m1)[1]
    err = np.linalg.norm(cam - cam_gt, axis=1)
    sem = lambda a: a.std(axis=0, ddof=1) / np.sqrt(n)
    # what the per-pair statistics alone predict for the end point: a yaw bias b turns the heading by b*n and the
    # position by step*b*n^2/2; a yaw spread s walks the heading randomly: step*s*n^1.5/sqrt(3); a translation bias adds up
    yaw_b, yaw_s = dr[:, 1].mean(), dr[:, 1].std(ddof=1)
    pred_bias = step * abs(yaw_b) * n * n / 2.0
    pred_walk = step * yaw_s * n ** 1.5 / np.sqrt(3.0)
    pred_trans = np.linalg.norm(dt.mean(axis=0)) * n
    return {
        "name": name, "pairs": n, "path_m": step * n,
        "rot_bias_rad": dr.mean(axis=0).tolist(), "rot_sem_rad": sem(dr).tolist(), "rot_std_rad": dr.std(axis=0, ddof=1).tolist(),
        "t_bias_m": dt.mean(axis=0).tolist(), "t_sem_m": sem(dt).tolist(), "t_std_m": dt.std(axis=0, ddof=1).tolist(),
        "pose_err_max": float(max(np.abs(np.asarray(rel) - np.asarray(gt)).max(), 0.0)),
        "end_point_drift_m": float(err[-1]), "max_drift_m": float(err.max()), "drift_pct_of_path": float(100 * err.max() / (step * n)),
        "predicted_from_yaw_bias_m": float(pred_bias), "predicted_from_yaw_walk_m": float(pred_walk),
        "predicted_from_translation_bias_m": float(pred_trans),
    }


def host_runs(seq, k, m1, m2, max_iterations=1000):
    """The host path (exact `random` stream, pnp.py) and two variations of its refinement input, pair by pair."""
    from slam_amd import frontend, pnp, vo
    ws, pairs = vo.front_end_tables(seq.frames, (k, m1, m2), thresh=1)
    random.seed(6)
    rel, rel_lr, rel_gtX = [], [], []
    tr = m2[:, 3]
    for p, tab in enumerate(pairs):
        Rt, ratio, mask = frontend.ransac_on_arrays(tab["X"], tab["uvL"], tab["uvR"], k, m2, thresh=2, max_iterations=max_iterations)
        rel.append(Rt)
        sup = np.flatnonzero(mask)
        X, uvL, uvR = tab["X"][sup], tab["uvL"][sup].astype(np.float64), tab["uvR"][sup].astype(np.float64)
        rel_lr.append(pnp.refine_pnp(X, uvL, k, Rt, points_2d_right=uvR, t_right=tr))
        # ground-truth 3-D points of the same supporters (in the earlier camera's frame), noisy pixels as they are
        f0 = seq.frames[p]
        ids = f0.idL[tab["rows"][sup, 0]]
        ok = ids >= 0
        Xw = seq.world_points(ids[ok])
        Xc = Xw @ f0.Rt[:, :3].T + f0.Rt[:, 3]
        rel_gtX.append(pnp.refine_pnp(Xc, uvL[ok], k, Rt))
    return rel, rel_lr, rel_gtX


def device_run(seq, k, m1, m2, n_hyp=1000, stereo_refine=False):
    import torch
    from slam_amd import pipeline
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seq.frames, torch.device("cuda", 0))
    ws = fe.prepare(tb, n_hyp=n_hyp)
    fe.run_ransac(ws, 6, stereo_refine=stereo_refine)
    torch.cuda.synchronize()
    P = ws.P
    return list(ws.pose.cpu().numpy()[:P].reshape(P, 3, 4)), float(ws.ratio[:P].mean().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--kp", type=int, default=2000)
    ap.add_argument("--long-frames", type=int, default=0, help="also the device RANSAC alone on a sequence this long")
    a = ap.parse_args()
    from slam_amd.synthetic import SyntheticSequence, read_cameras
    k, m1, m2 = read_cameras()
    out = []
    for noise in (0.3, 0.0):
        seq = SyntheticSequence(a.kp, a.frames, noise=noise, keep_world=True)
        gt = [seq.relative_pose(p, p + 1) for p in range(a.frames - 1)]
        dev, ratio = device_run(seq, k, m1, m2)
        out.append(dict(summarize(f"device RANSAC, noise {noise}", dev, gt, m1, seq.step), mean_inlier_ratio=ratio))
        dev, ratio = device_run(seq, k, m1, m2, stereo_refine=True)
        out.append(dict(summarize(f"device RANSAC, refinement on left+right residuals, noise {noise}", dev, gt, m1,
                                  seq.step), mean_inlier_ratio=ratio))
        rel, rel_lr, rel_gtX = host_runs(seq, k, m1, m2)
        out.append(summarize(f"host solver (left-image refinement), noise {noise}", rel, gt, m1, seq.step))
        out.append(summarize(f"host solver + refinement on left+right residuals, noise {noise}", rel_lr, gt, m1, seq.step))
        out.append(summarize(f"host solver, refinement on ground-truth 3-D points, noise {noise}", rel_gtX, gt, m1, seq.step))
        print(f"[drift] noise {noise} done", file=sys.stderr, flush=True)
    if a.long_frames:
        seq = SyntheticSequence(a.kp, a.long_frames, noise=0.3)
        gt = [seq.relative_pose(p, p + 1) for p in range(a.long_frames - 1)]
        dev, ratio = device_run(seq, k, m1, m2)
        out.append(dict(summarize("device RANSAC, noise 0.3, long", dev, gt, m1, seq.step), mean_inlier_ratio=ratio))
        dev, ratio = device_run(seq, k, m1, m2, stereo_refine=True)
        out.append(dict(summarize("device RANSAC, refinement on left+right residuals, noise 0.3, long", dev, gt, m1,
                                  seq.step), mean_inlier_ratio=ratio))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
