#!/usr/bin/env python3
"""Times the two L2 matchers (fp32 VALU, int8 matrix cores) on a batch of SIFT-like frame matches."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import ops                                                   # noqa: E402

dev = torch.device("cuda:0")
B, N, D = int(os.environ.get("L2_BATCH", "128")), 2000, 128
rng = np.random.default_rng(0)
q = torch.from_numpy(rng.integers(0, 256, (B * N, D)).astype(np.float32)).to(dev)
t = torch.from_numpy(rng.integers(0, 256, (B * N, D)).astype(np.float32)).to(dev)
off = np.arange(B) * N
seg = torch.from_numpy(np.stack([off, np.full(B, N), off, np.full(B, N), off], 1).astype(np.int32)).to(dev)
q8, t8 = ops.l2_pack_int8(q), ops.l2_pack_int8(t)
assert q8.usable() and t8.usable()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


ms_f, a = timed(lambda: ops.l2_match(q, t, seg, N, N, k=2, mode=1, thr=0.8))
ms_i, b = timed(lambda: ops.l2_match_int8(q8, t8, seg, N, N, k=2, mode=1, thr=0.8))
ms_p, _ = timed(lambda: ops.l2_pack_int8(q))
same = all(torch.equal(x, y) for x, y in zip(a, b))
pairs = B * N * N
print(f"batch {B} x {N}x{N} dim {D}: fp32 VALU {ms_f:.3f} ms, int8 MFMA {ms_i:.3f} ms ({ms_f / ms_i:.1f}x), "
      f"pack {ms_p:.3f} ms per table, identical {same}")
print(f"int8: {pairs / (ms_i * 1e-3):.3e} descriptor pairs/s, {2 * D * pairs / (ms_i * 1e-3) / 1e12:.0f} TOP/s int8")
