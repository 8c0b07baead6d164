#!/usr/bin/env python3
"""Fuzzes the filtered scoring kernels against the plain fp64 kernel on the GPU: poses in a tight
cloud around the truth, observations displaced to within +-1e-2 relative of the threshold.
Any disagreement in a supporter count is printed and fails the run."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, ops                                                   # noqa: E402
from slam_amd.synthetic import read_cameras, seeded_hypotheses             # noqa: E402

dev = torch.device("cuda:0")
k, m1, m2 = read_cameras()
tr = m2[:, 3]
eye = np.hstack([np.eye(3), np.zeros((3, 1))])
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = 0
total = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    rng = np.random.default_rng(1000 + seed)
    n, H = 1501 + seed, 4096
    thresh = float(rng.choice([0.3, 1.0, 2.0, 2.0, 3.7]))
    scale = float(rng.choice([1.0, 1.0, 30.0, 0.05]))
    X = rng.uniform(-20, 20, n) * scale
    Y = rng.uniform(-3, 2, n) * scale
    Z = np.exp(rng.uniform(np.log(4), np.log(120), n)) * scale
    pose = seeded_hypotheses(eye, 2, seed=seed)[1]
    pose = eye + (pose - eye) * 0.05
    pose[:, 3] *= scale
    L = X[:, None] * pose[:, 0] + Y[:, None] * pose[:, 1] + Z[:, None] * pose[:, 2] + pose[:, 3]
    uL = k[0, 0] * L[:, 0] / L[:, 2] + k[0, 2]
    vL = k[1, 1] * L[:, 1] / L[:, 2] + k[1, 2]
    uR = k[0, 0] * (L[:, 0] + tr[0] * scale) / L[:, 2] + k[0, 2]
    delta = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-7, -2, n)
    ang = rng.uniform(0, 2 * np.pi, n)
    r = thresh * (1.0 + delta)
    right = rng.random(n) < 0.5
    uL32 = (uL + np.where(right, 0.1, r) * np.cos(ang)).astype(np.float32)
    vL32 = (vL + np.where(right, 0.1, r) * np.sin(ang)).astype(np.float32)
    uR32 = (uR + np.where(right, r, 0.1) * np.cos(ang)).astype(np.float32)
    vR32 = (vL + np.where(right, r, 0.1) * np.sin(ang)).astype(np.float32)
    Rt = seeded_hypotheses(pose, H, seed=seed + 1)
    Rt = pose + (Rt - pose) * 10.0 ** rng.uniform(-6, -1, (H, 1, 1))
    args = (t(Rt), k, tr * scale, t(X), t(Y), t(Z), t(uL32), t(vL32), t(uR32), t(vR32))
    plain = ops.score_hypotheses(*args, thresh=thresh, cap=n, n_host=n, want_masks=False, fast=False)
    for impl in ("wide", "split", "f64", "mfma"):
        _lib.select_kernels(score=impl)
        fast = ops.score_hypotheses(*args, thresh=thresh, cap=n, n_host=n, want_masks=False, fast=True)
        torch.cuda.synchronize()
        diff = int((fast != plain).sum())
        total += H * n
        if diff:
            bad += diff
            print(f"seed {seed} impl {impl}: {diff} hypotheses disagree (thresh {thresh}, scale {scale})")
    p = plain.cpu().numpy()[0]
    print(f"seed {seed}: thresh {thresh} scale {scale} counts min/median/max {p.min()}/{int(np.median(p))}/{p.max()} of {n}", flush=True)
print(f"{total:.3g} evaluations checked, {bad} disagreements")
sys.exit(1 if bad else 0)
