"""ctypes front for the C oracle (oracle/slam_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py -- always as the checker, never as the thing
being measured as "the product" and never from the slam_amd package.

Parity status: ratio test, row filter, triangulation, join and supporter counts are pinned against
outputs of the reference's own function bodies (tests/golden/reference_golden.npz,
tests/test_reference_golden.py); the k-NN scan restates cv2's knnMatch and the minimal-sample
generator has no counterpart in the reference: "parity unpinned" for those (see slam_oracle.c).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libslam_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc if the shared object is missing or stale."""
    src = os.path.join(_HERE, "slam_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_max_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def max_threads():
    return int(lib().oracle_max_threads())


def hamming_knn(q, t, k, desc_bytes=None, threads=1):
    """q [Nq, stride] u8, t [Nt, stride] u8 -> (idx [Nq,k] i32, dist [Nq,k] i32)."""
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    nq, qs = q.shape if q.ndim == 2 else (0, 0)
    nt, ts = t.shape if t.ndim == 2 else (0, 0)
    if desc_bytes is None:
        desc_bytes = qs if nq else ts
    idx = np.full((nq, k), -1, dtype=np.int32)
    dist = np.full((nq, k), np.iinfo(np.int32).max, dtype=np.int32)
    if nq:
        lib().oracle_hamming_knn(_p(q), nq, qs, _p(t), nt, ts if nt else qs,
                                 int(desc_bytes), int(k), _p(idx), _p(dist), int(threads))
    return idx, dist


def ratio_filter(idx, dist, thr):
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    dist = np.ascontiguousarray(dist, dtype=np.int32)
    n = idx.shape[0]
    keep = np.zeros(n, dtype=np.uint8)
    if n:
        assert idx.shape[1] == 2
        lib().oracle_ratio_filter(_p(idx), _p(dist), n, ctypes.c_double(thr), _p(keep))
    return keep


def row_filter(idx, yq, yt, thr):
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    yq = np.ascontiguousarray(yq, dtype=np.float32)
    yt = np.ascontiguousarray(yt, dtype=np.float32)
    n = idx.shape[0]
    keep = np.zeros(n, dtype=np.uint8)
    if n:
        lib().oracle_row_filter(_p(idx), idx.shape[1], n, _p(yq), _p(yt),
                                ctypes.c_double(thr), _p(keep))
    return keep


def triangulate_dlt(P, Q, x1, y1, x2, y2, threads=1):
    P = np.ascontiguousarray(P, dtype=np.float64).reshape(12)
    Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(12)
    x1, y1, x2, y2 = (np.ascontiguousarray(a, dtype=np.float32) for a in (x1, y1, x2, y2))
    n = x1.shape[0]
    X = np.zeros(n); Y = np.zeros(n); Z = np.zeros(n)
    valid = np.zeros(n, dtype=np.uint8)
    if n:
        lib().oracle_triangulate_dlt(_p(P), _p(Q), _p(x1), _p(y1), _p(x2), _p(y2), n,
                                     _p(X), _p(Y), _p(Z), _p(valid), int(threads))
    return X, Y, Z, valid


def join(t_idx, t_keep, prev_map, prev_valid, cur_map):
    t_idx = np.ascontiguousarray(t_idx, dtype=np.int32)
    t_keep = np.ascontiguousarray(t_keep, dtype=np.uint8)
    prev_map = np.ascontiguousarray(prev_map, dtype=np.int32)
    prev_valid = np.ascontiguousarray(prev_valid, dtype=np.uint8)
    cur_map = np.ascontiguousarray(cur_map, dtype=np.int32)
    nq = t_idx.shape[0]
    rows = np.zeros((max(nq, 1), 4), dtype=np.int32)
    stride = t_idx.shape[1] if t_idx.ndim == 2 else 1
    m = lib().oracle_join(_p(t_idx), stride, _p(t_keep), nq, _p(prev_map), _p(prev_valid),
                          _p(cur_map), cur_map.shape[0], _p(rows)) if nq else 0
    return rows[:m].copy()


def score_hypotheses(Rt, K, t_right, X, Y, Z, uL, vL, uR, vR, thresh, want_masks=False, threads=1):
    Rt = np.ascontiguousarray(Rt, dtype=np.float64).reshape(-1, 12)
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    t_right = np.ascontiguousarray(t_right, dtype=np.float64).reshape(3)
    X, Y, Z = (np.ascontiguousarray(a, dtype=np.float64) for a in (X, Y, Z))
    uL, vL, uR, vR = (np.ascontiguousarray(a, dtype=np.float32) for a in (uL, vL, uR, vR))
    H, N = Rt.shape[0], X.shape[0]
    counts = np.zeros(H, dtype=np.int32)
    words = (N + 63) // 64
    masks = np.zeros((H, words), dtype=np.uint64) if want_masks else None
    if H:
        lib().oracle_score_hypotheses(_p(Rt), H, _p(K), _p(t_right), _p(X), _p(Y), _p(Z),
                                      _p(uL), _p(vL), _p(uR), _p(vR), N, ctypes.c_double(thresh),
                                      _p(counts), _p(masks) if want_masks else None, int(threads))
    return (counts, masks) if want_masks else counts


def ransac_sample(seed, n_sets, H, n):
    """[n_sets, H, 4] int32 minimal samples of the batched RANSAC's per-hypothesis generator."""
    n = np.ascontiguousarray(n, dtype=np.int32)
    out = np.zeros((n_sets, H, 4), dtype=np.int32)
    if n_sets and H:
        lib().oracle_ransac_sample(ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), int(n_sets), int(H), _p(n), _p(out))
    return out


def l2_knn(q, t, k, threads=1):
    """q [Nq, D] f32, t [Nt, D] f32 -> (idx [Nq,k] i32, dist [Nq,k] f32), cv2 NORM_L2 semantics."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    t = np.ascontiguousarray(t, dtype=np.float32)
    nq = q.shape[0] if q.ndim == 2 else 0
    nt = t.shape[0] if t.ndim == 2 else 0
    dim = q.shape[1] if nq else (t.shape[1] if nt else 0)
    idx = np.full((nq, k), -1, dtype=np.int32)
    dist = np.full((nq, k), np.finfo(np.float32).max, dtype=np.float32)
    if nq:
        lib().oracle_l2_knn(_p(q), nq, _p(t), nt, int(dim), int(k), _p(idx), _p(dist), int(threads))
    return idx, dist
