#!/usr/bin/env python3
"""Interleaved A/B timing, in one process, of the kernel variants the library selects through
SLAM_KNN_IMPL (matrix-core / VALU popcount) and SLAM_SCORE_IMPL (fp32 first level / fp64 only);
both are read on every call."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, pipeline                                                     # noqa: E402
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses  # noqa: E402

P, N, H = int(os.environ.get("AB_PAIRS", "128")), 2000, int(os.environ.get("AB_HYP", "10000"))
dev = torch.device("cuda:0")
seq = SyntheticSequence(N, P + 1)
Rt = torch.from_numpy(np.stack([seeded_hypotheses(seq.relative_pose(p, p + 1), H, seed=6 + p) for p in range(P)])).to(dev)
k, m1, m2 = read_cameras()
fe = pipeline.FrontEnd(k, m1, m2)
tb = pipeline.FrameTables.from_frames(seq.frames, dev)
ws = fe.prepare(tb, n_hyp=H)
fe.run(ws, Rt)
torch.cuda.synchronize()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {"score": {"mixed": [], "f64": []}, "knn": {"valu": [], "fp4": []}}
for rnd in range(8):
    for impl, simpl in (("fp4", "mixed"), ("valu", "f64")):
        _lib.select_kernels(score=simpl)
        res["score"][simpl].append(timed(lambda: fe.score(ws, Rt, stream)))
        _lib.select_kernels(knn=impl)
        res["knn"][impl].append(timed(lambda: (fe.stereo_match(ws, stream), fe.temporal_match(ws, stream))))
for kname, d in res.items():
    for impl, v in d.items():
        print(f"{kname:6s} {impl:5s} median {np.median(v):.4f} ms  min {np.min(v):.4f} ms")
