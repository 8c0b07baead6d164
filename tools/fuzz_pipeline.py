#!/usr/bin/env python3
"""End-to-end fuzz of the batched front-end against the CPU oracle: random sequences (keypoint count,
frame count, descriptor length, bit-flip rate, pixel noise, ratio threshold, hypothesis count), every
stage compared exactly -- stereo map, triangulation validity, consensus rows, supporter counts of
every hypothesis, winner, inlier count, inlier rows and feature locations."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu as oracle                                                     # noqa: E402
from slam_amd import pipeline                                                        # noqa: E402
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses    # noqa: E402

dev = torch.device("cuda:0")
USE_L2 = os.environ.get("FUZZ_L2", "0") == "1"     # SIFT-like integer-valued float descriptors, L2 norm


def knn(q, t, kk):
    return oracle.l2_knn(q, t, kk, threads=8) if USE_L2 else oracle.hamming_knn(q, t, kk)


def ratio_keep(tidx, tdist, ratio):
    if not USE_L2:
        return oracle.ratio_filter(tidx, tdist, ratio)
    return np.array([i2 >= 0 and float(a) < ratio * float(b) for (a, b), (_, i2) in zip(tdist, tidx)], dtype=np.uint8)

k, m1, m2 = read_cameras()
P, Q = k @ m1, k @ m2
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    rng = np.random.default_rng(9000 + seed)
    n_kp = int(rng.choice([40, 100, 257, 400, 700, 1100]))
    n_fr = int(rng.integers(2, 6))
    D = int(rng.choice([32, 61, 61, 64]))
    ratio = float(rng.choice([0.6, 0.8, 1.0]))
    H = int(rng.choice([1, 63, 64, 200, 513]))
    seq = SyntheticSequence(n_kp, n_fr, seed=seed, desc_bytes=D, flip=float(rng.choice([0.02, 0.08, 0.15])),
                            noise=float(rng.choice([0.1, 0.3, 1.0])))
    if USE_L2:
        dim = int(rng.choice([64, 128]))
        proj = rng.integers(0, 2, (D * 8, dim)).astype(np.float32)

        def to_float(desc):        # a fixed random embedding of the bits into integer-valued float dimensions
            return np.minimum(np.unpackbits(desc, axis=1).astype(np.float32) @ proj, 255.0).astype(np.float32)

        seq.frames = [type(f)(f.kpL, f.kpR, to_float(f.descL), to_float(f.descR), f.idL, f.idR, f.Rt)
                      for f in seq.frames]
    npair = n_fr - 1
    Rt_host = np.stack([seeded_hypotheses(seq.relative_pose(p, p + 1), H, seed=seed + p) for p in range(npair)])
    fe = pipeline.FrontEnd(k, m1, m2, ratio_thresh=ratio)
    tb = pipeline.FrameTables.from_frames(seq.frames, dev)
    ws = fe.prepare(tb, n_hyp=H)
    fe.run(ws, torch.from_numpy(Rt_host).to(dev))
    torch.cuda.synchronize()
    errs = []
    per_frame = []
    for f in seq.frames:
        idx, _ = knn(f.descL, f.descR, 1)
        keep = oracle.row_filter(idx, f.kpL[:, 1], f.kpR[:, 1], 2.0)
        j = np.where(keep > 0, idx[:, 0], 0)
        X, Y, Z, valid = oracle.triangulate_dlt(P, Q, f.kpL[:, 0], f.kpL[:, 1], f.kpR[j, 0], f.kpR[j, 1])
        per_frame.append((np.where(keep > 0, idx[:, 0], -1), X, Y, Z, (valid * (keep > 0)).astype(np.uint8)))
    if not np.array_equal(ws.smap.cpu().numpy(), np.concatenate([p[0] for p in per_frame])):
        errs.append("stereo map")
    if not np.array_equal(ws.valid.cpu().numpy(), np.concatenate([p[4] for p in per_frame])):
        errs.append("triangulation validity")
    counts = ws.counts.cpu().numpy()
    n_inl = ws.n_inl.cpu().numpy()
    inl_rows, inl_loc = ws.inl_rows.cpu().numpy(), ws.inl_loc.cpu().numpy()
    cap = ws.cap
    for p in range(npair):
        f1 = seq.frames[p + 1]
        tidx, tdist = knn(seq.frames[p].descL, f1.descL, 2)
        tkeep = ratio_keep(tidx, tdist, ratio)
        rows = oracle.join(tidx, tkeep, per_frame[p][0], per_frame[p][4], per_frame[p + 1][0])
        got = ws.cons.rows.cpu().numpy()[p * cap:p * cap + rows.shape[0]]
        if int(ws.cons.count[p]) != rows.shape[0] or not np.array_equal(got, rows):
            errs.append(f"consensus rows of pair {p}")
            continue
        if rows.shape[0] == 0:
            continue
        pts = (per_frame[p][1][rows[:, 0]], per_frame[p][2][rows[:, 0]], per_frame[p][3][rows[:, 0]],
               f1.kpL[rows[:, 2], 0], f1.kpL[rows[:, 2], 1], f1.kpR[rows[:, 3], 0], f1.kpR[rows[:, 3], 1])
        rc = oracle.score_hypotheses(Rt_host[p], k, m2[:, 3], *pts, 2.0)
        if not np.array_equal(counts[p], rc):
            errs.append(f"supporter counts of pair {p}")
        if int(ws.best[p]) != int(np.argmax(rc)) or n_inl[p] != rc.max():
            errs.append(f"winner of pair {p}")
            continue
        # the winner's inlier rows and feature locations, packed back to back in pair order
        _, wmask = oracle.score_hypotheses(Rt_host[p][int(np.argmax(rc))][None], k, m2[:, 3], *pts, 2.0, want_masks=True)
        bits = np.unpackbits(wmask[0].view(np.uint8), bitorder="little")[:rows.shape[0]].astype(bool)
        off = int(n_inl[:p].sum())
        want_rows = rows[bits]
        f0 = seq.frames[p]
        want_loc = np.stack([f0.kpL[want_rows[:, 0], 0], f0.kpR[want_rows[:, 1], 0], f0.kpL[want_rows[:, 0], 1],
                             f1.kpL[want_rows[:, 2], 0], f1.kpR[want_rows[:, 3], 0], f1.kpL[want_rows[:, 2], 1]], axis=1)
        if not np.array_equal(inl_rows[off:off + len(want_rows)], want_rows):
            errs.append(f"inlier rows of pair {p}")
        if not np.array_equal(inl_loc[off:off + len(want_rows)], want_loc.astype(np.float32)):
            errs.append(f"inlier locations of pair {p}")
    tag = "ok " if not errs else "BAD"
    print(f"seed {seed}: {tag} kp {n_kp} frames {n_fr} D {D} ratio {ratio} H {H} consensus "
          f"{ws.cons.count.cpu().numpy()[:npair].tolist()} {'; '.join(errs)}", flush=True)
    bad += len(errs)
print(f"{bad} mismatches")
sys.exit(1 if bad else 0)
