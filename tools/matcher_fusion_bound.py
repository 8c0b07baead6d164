#!/usr/bin/env python3
"""Upper bound on what fusing the step's two matchings into one launch could save: the same queries
(512 frames x 2000 left descriptors) against 2000-row train sets in two launches, and against one
4000-row train set (R_i and L_{i+1} back to back) in one launch -- the fused kernel would do exactly
that scan with one prologue per wavefront and one grid tail.  Times only; the 4000-row result is
not what either matching asks for."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import ops                                                               # noqa: E402

dev = torch.device("cuda:0")
F, N, D = 512, 2000, 61
rng = np.random.default_rng(0)
q = ops.pack_descriptors(torch.from_numpy(rng.integers(0, 256, (F * N, D), dtype=np.uint8)).to(dev), D)
t = ops.pack_descriptors(torch.from_numpy(rng.integers(0, 256, (2 * F * N, D), dtype=np.uint8)).to(dev), D)
off = np.arange(F, dtype=np.int64) * N
seg_a = ops.make_segments(off, np.full(F, N), 2 * off, np.full(F, N), off, dev)              # first half of each item's rows
seg_b = ops.make_segments(off, np.full(F, N), 2 * off + N, np.full(F, N), off, dev)          # second half
seg_ab = ops.make_segments(off, np.full(F, N), 2 * off, np.full(F, 2 * N), off, dev)         # both, one scan


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rnd in range(3):
    two = timed(lambda: (ops.hamming_match(q, t, seg_a, N, N, D, k=1, n_rows=F * N),
                         ops.hamming_match(q, t, seg_b, N, N, D, k=2, n_rows=F * N)))
    one = timed(lambda: ops.hamming_match(q, t, seg_ab, N, 2 * N, D, k=2, n_rows=F * N))
    print(f"two launches (k=1 + k=2, 2000 rows each) {two:.4f} ms | one launch over 4000 rows (k=2) {one:.4f} ms | "
          f"saving at most {100 * (1 - one / two):.1f} %", flush=True)
