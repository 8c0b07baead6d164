#!/usr/bin/env python3
"""Fuzzes the matrix-core matcher against the VALU matcher on the GPU (both are checked against the
CPU oracle by the tests at small sizes): random batches of ragged items, descriptor lengths on
both strides, sizes around the tile / super-tile / key-chunk boundaries, heavy ties (few distinct
descriptors), all three filter modes.  Any difference in index, distance or keep flag fails."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, ops                                                   # noqa: E402

dev = torch.device("cuda:0")
EDGE = [0, 1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 2047, 2048, 2049, 3000]
bad = 0
pairs = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    rng = np.random.default_rng(5000 + seed)
    D = int(rng.choice([8, 16, 29, 30, 31, 32, 33, 48, 61, 62, 63, 64]))
    stride = 32 if D <= 32 else 64
    batch = int(rng.integers(1, 9))
    nq = [int(rng.choice(EDGE)) if rng.random() < 0.6 else int(rng.integers(0, 2500)) for _ in range(batch)]
    nt = [int(rng.choice(EDGE)) if rng.random() < 0.6 else int(rng.integers(0, 2500)) for _ in range(batch)]
    distinct = int(rng.choice([2, 5, 50, 100000]))              # few distinct rows = many exact ties
    pool = rng.integers(0, 256, size=(min(distinct, 4096), D), dtype=np.uint8)
    def table(n):
        rows = pool[rng.integers(0, pool.shape[0], n)].copy()
        flip = rng.random(n) < 0.5                                 # near-duplicates: one bit flipped
        rows[flip, rng.integers(0, D)] ^= np.uint8(1 << int(rng.integers(0, 8)))
        return rows
    q = np.concatenate([table(n) for n in nq] + [np.zeros((0, D), np.uint8)])
    t = np.concatenate([table(n) for n in nt] + [np.zeros((0, D), np.uint8)])
    qoff = np.concatenate([[0], np.cumsum(nq)])
    toff = np.concatenate([[0], np.cumsum(nt)])
    seg = np.stack([qoff[:-1], nq, toff[:-1], nt, qoff[:-1]], axis=1).astype(np.int32)
    mode = int(rng.integers(0, 3))
    thr = float(rng.choice([0.5, 0.6, 0.8, 1.0])) if mode == 1 else float(rng.choice([0.5, 2.0, 10.0]))
    yq = torch.from_numpy(rng.uniform(0, 370, max(q.shape[0], 1)).astype(np.float32)).to(dev)
    yt = torch.from_numpy(rng.uniform(0, 370, max(t.shape[0], 1)).astype(np.float32)).to(dev)
    qd = ops.pack_descriptors(torch.from_numpy(q).to(dev), D, stride) if q.shape[0] else torch.zeros((0, stride), dtype=torch.uint8, device=dev)
    td = ops.pack_descriptors(torch.from_numpy(t).to(dev), D, stride) if t.shape[0] else torch.zeros((0, stride), dtype=torch.uint8, device=dev)
    out = {}
    for impl in ("valu", "fp4"):
        _lib.select_kernels(knn=impl)
        idx, dist, keep = ops.hamming_match(qd, td, torch.from_numpy(seg).to(dev), max(nq), max(nt), D, k=2, mode=mode,
                                            thr=thr, yq=yq if mode == 2 else None, yt=yt if mode == 2 else None,
                                            n_rows=int(q.shape[0]))
        torch.cuda.synchronize()
        out[impl] = (idx.cpu().numpy(), dist.cpu().numpy(), None if keep is None else keep.cpu().numpy())
    b = out["valu"]
    d = 0
    for a in (out["fp4"],):
        d += int((a[0] != b[0]).sum() + (a[1] != b[1]).sum() + (0 if a[2] is None else (a[2] != b[2]).sum()))
    pairs += int(sum(x * y for x, y in zip(nq, nt)))
    if d:
        bad += d
        print(f"seed {seed}: {d} differences (D {D}, mode {mode}, nq {nq}, nt {nt}, distinct {distinct})")
    else:
        print(f"seed {seed}: ok  D {D} mode {mode} batch {batch} distinct {distinct} rows {q.shape[0]}x{t.shape[0]}", flush=True)
print(f"{pairs:.3g} descriptor pairs checked, {bad} differences")
sys.exit(1 if bad else 0)
