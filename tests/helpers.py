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


def scipy_pnp_minimum(X, uv, k, start, uv_right=None, t_right=None):
    """The minimiser of the reprojection error found by a solver that shares no code with this
    repository: scipy.optimize.least_squares (MINPACK's Levenberg-Marquardt) over a rotation vector
    and a translation, started from `start` [3,4].  The cost is the one cv2.solvePnP(
    SOLVEPNP_ITERATIVE) minimises for the reference (left-image pixels, utils/utils.py:452); with
    uv_right / t_right the right image's residuals are added.  Returns (pose [3,4], cost)."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation

    X, uv = np.asarray(X, np.float64), np.asarray(uv, np.float64)
    fx, fy, cx, cy = k[0, 0], k[1, 1], k[0, 2], k[1, 2]

    def residuals(p):
        Y = X @ Rotation.from_rotvec(p[:3]).as_matrix().T + p[3:]
        r = [fx * Y[:, 0] / Y[:, 2] + cx - uv[:, 0], fy * Y[:, 1] / Y[:, 2] + cy - uv[:, 1]]
        if uv_right is not None:
            Yr = Y + np.asarray(t_right, np.float64)
            r += [fx * Yr[:, 0] / Yr[:, 2] + cx - uv_right[:, 0], fy * Yr[:, 1] / Yr[:, 2] + cy - uv_right[:, 1]]
        return np.concatenate(r)

    p0 = np.concatenate([Rotation.from_matrix(start[:, :3]).as_rotvec(), start[:, 3]])
    sol = least_squares(residuals, p0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=2000)
    pose = np.hstack([Rotation.from_rotvec(sol.x[:3]).as_matrix(), sol.x[3:, None]])
    return pose, float((sol.fun ** 2).sum())
