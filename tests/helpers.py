"""Shared helpers for the parity tests (oracle side + tensor plumbing)."""
import numpy as np
import torch

from oracle import cpu as oracle
from slam_amd import ops


def pad_desc(d, stride):
    """[n, D] uint8 -> [n, stride] with 0xAB garbage in the pad (the kernels must ignore it)."""
    n, D = d.shape
    out = np.full((n, stride), 0xAB, dtype=np.uint8)
    out[:, :D] = d
    return out


def ragged_tables(items, stride):
    """items: list of (q [nq,D], t [nt,D]) -> concatenated padded tables + segment columns."""
    qs, ts, seg = [], [], []
    qo = to = 0
    for q, t in items:
        qs.append(pad_desc(q, stride))
        ts.append(pad_desc(t, stride))
        seg.append((qo, q.shape[0], to, t.shape[0], qo))
        qo += q.shape[0]
        to += t.shape[0]
    Q = np.concatenate(qs) if qs else np.zeros((0, stride), np.uint8)
    T = np.concatenate(ts) if ts else np.zeros((0, stride), np.uint8)
    return Q, T, np.array(seg, dtype=np.int32).reshape(-1, 5)


def gpu_match(items, D, stride, k, device, mode=0, thr=0.0, yq=None, yt=None):
    Q, T, seg = ragged_tables(items, stride)
    if T.shape[0] == 0:
        T = np.zeros((1, stride), np.uint8)
    if Q.shape[0] == 0:
        Q = np.zeros((1, stride), np.uint8)
    n_rows = int(sum(q.shape[0] for q, _ in items))
    max_nq = max([q.shape[0] for q, _ in items] + [0])
    max_nt = max([t.shape[0] for _, t in items] + [0])
    tq = torch.from_numpy(Q).to(device)
    tt = torch.from_numpy(T).to(device)
    tseg = torch.from_numpy(seg).to(device)
    tyq = torch.from_numpy(np.ascontiguousarray(yq, dtype=np.float32)).to(device) if yq is not None else None
    tyt = torch.from_numpy(np.ascontiguousarray(yt, dtype=np.float32)).to(device) if yt is not None else None
    idx, dist, keep = ops.hamming_match(tq, tt, tseg, max_nq, max_nt, D, k=k, mode=mode, thr=thr, yq=tyq, yt=tyt,
                                        n_rows=max(n_rows, 1))
    torch.cuda.synchronize()
    idx, dist = idx.cpu().numpy()[:n_rows], dist.cpu().numpy()[:n_rows]
    keep = keep.cpu().numpy()[:n_rows] if keep is not None else None
    return idx, dist, keep


def oracle_match(items, D, k):
    idxs, dists = [], []
    for q, t in items:
        i, d = oracle.hamming_knn(q[:, :D], t[:, :D] if t.shape[0] else np.zeros((0, D), np.uint8), 2, desc_bytes=D)
        if k == 1:
            i[:, 1] = -1
            d[:, 1] = np.iinfo(np.int32).max
        idxs.append(i)
        dists.append(d)
    if not idxs:
        return np.zeros((0, 2), np.int32), np.zeros((0, 2), np.int32)
    return np.concatenate(idxs), np.concatenate(dists)
