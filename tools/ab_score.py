#!/usr/bin/env python3
"""A/B of the scoring kernel variants on the benchmark's batch, in one process, interleaved:
counts must be identical, times are HIP-event medians per launch.
    AB_PAIRS=511 AB_HYP=10000 python tools/ab_score.py wide mfma"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, pipeline                                                     # noqa: E402
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses  # noqa: E402

impls = sys.argv[1:] or ["wide", "mfma"]
P, N, H = int(os.environ.get("AB_PAIRS", "511")), int(os.environ.get("AB_KP", "2000")), int(os.environ.get("AB_HYP", "10000"))
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
evals = float((ws.cons.count[:P].double() * H).sum().item())
counts, times = {}, {i: [] for i in impls}
for rnd in range(6):
    for impl in impls:
        _lib.select_kernels(score=impl)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ws.counts.zero_()
        e0.record()
        fe.score(ws, Rt, stream)
        e1.record()
        torch.cuda.synchronize()
        times[impl].append(e0.elapsed_time(e1))
        c = ws.counts.cpu().numpy().copy()
        if impl in counts:
            assert np.array_equal(c, counts[impl]), f"{impl}: counts differ between runs"
        counts[impl] = c
ref = counts[impls[0]]
for impl in impls:
    diff = int((counts[impl] != ref).sum())
    print(f"{impl:6s} median {np.median(times[impl]):.4f} ms  min {np.min(times[impl]):.4f} ms  "
          f"{evals / np.median(times[impl]) / 1e9:.3f} T evaluations/s  counts differing from {impls[0]}: {diff}", flush=True)
sys.exit(1 if any((counts[i] != ref).any() for i in impls) else 0)
