"""Tensor-level calls into the slam_amd C ABI.

Arguments are PyTorch-ROCm tensors living on one GPU; PyTorch is used for device
memory and streams only.  Every function checks device, dtype and contiguity,
passes raw device pointers to the HIP library, and returns tensors on the same
device.  Nothing here computes on the CPU.
"""
import ctypes

import numpy as np
import torch

from . import _lib

INT32_MAX = 2147483647


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise _lib.SlamError(f"{name}: tensor must live on the GPU (got {t.device}); slam_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def _host_f64(a, n, name):
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64)).reshape(-1)
    if arr.size != n:
        raise ValueError(f"{name}: expected {n} values, got {arr.size}")
    return arr, arr.ctypes.data_as(ctypes.c_void_p)


def make_segments(q_start, q_count, t_start, t_count, out_start, device):
    """[batch,5] int32 segment table (q_start, q_count, t_start, t_count, out_start)."""
    seg = np.stack([np.asarray(a, dtype=np.int64) for a in (q_start, q_count, t_start, t_count, out_start)], axis=1)
    if seg.size and (seg.min() < 0 or seg.max() > INT32_MAX):
        raise ValueError("segment table entries must fit in non-negative int32")
    return torch.from_numpy(seg.astype(np.int32)).to(device)


def pack_descriptors(desc, desc_bytes=None, dst_stride=None, out=None):
    """[n, src_stride] uint8 -> [n, 32|64] uint8, zero padded (slam_pack_descriptors).
    out: an existing [n, dst_stride] uint8 table to fill instead of a new one."""
    lib = _lib.load()
    if desc.dim() != 2:
        raise ValueError("pack_descriptors: expected [n, bytes]")
    n, src_stride = desc.shape
    desc_bytes = src_stride if desc_bytes is None else int(desc_bytes)
    if dst_stride is None:
        dst_stride = 32 if desc_bytes <= 32 else 64
    if out is None:
        out = torch.empty((n, dst_stride), dtype=torch.uint8, device=desc.device)
    elif out.shape != (n, dst_stride) or out.dtype != torch.uint8 or not out.is_contiguous() or out.device != desc.device:
        raise ValueError("pack_descriptors: out must be a contiguous [n, dst_stride] uint8 table on the input's device")
    st = lib.slam_pack_descriptors(_dev(desc, torch.uint8, "desc"), n, desc_bytes, src_stride,
                                   ctypes.c_void_p(out.data_ptr()), dst_stride, _stream())
    _lib.check(st, "slam_pack_descriptors")
    return out


def hamming_match(q, t, seg, max_nq, max_nt, desc_bytes, k=2, mode=0, thr=0.0, yq=None, yt=None, n_rows=None):
    """Batched Hamming k-NN with the fused filter (slam_hamming_match).

    q, t: [rows, 32|64] uint8 packed descriptor tables.  seg: [batch,5] int32.
    Returns (idx [n_rows,2] i32, dist [n_rows,2] i32, keep [n_rows] u8 or None).
    """
    lib = _lib.load()
    if q.dim() != 2 or t.dim() != 2 or q.shape[1] != t.shape[1]:
        raise ValueError("hamming_match: q and t must be [rows, stride] with equal stride")
    stride = q.shape[1]
    batch = seg.shape[0]
    if n_rows is None:
        n_rows = q.shape[0]
    dev = q.device
    idx = torch.full((n_rows, 2), -1, dtype=torch.int32, device=dev)
    dist = torch.full((n_rows, 2), INT32_MAX, dtype=torch.int32, device=dev)
    keep = torch.zeros((n_rows,), dtype=torch.uint8, device=dev) if mode else None
    null = ctypes.c_void_p(0)
    st = lib.slam_hamming_match(
        _dev(q, torch.uint8, "q"), _dev(t, torch.uint8, "t"), _dev(seg, torch.int32, "seg"), batch, int(max_nq),
        int(max_nt), int(desc_bytes), stride, int(k), int(mode), float(thr),
        _dev(yq, torch.float32, "yq") if yq is not None else null,
        _dev(yt, torch.float32, "yt") if yt is not None else null,
        ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(dist.data_ptr()),
        ctypes.c_void_p(keep.data_ptr()) if keep is not None else null, _stream())
    _lib.check(st, "slam_hamming_match")
    return idx, dist, keep


FLT_MAX = float(np.finfo(np.float32).max)


def l2_match(q, t, seg, max_nq, max_nt, k=2, mode=0, thr=0.0, yq=None, yt=None, n_rows=None):
    """Batched float L2 k-NN with the fused filter (slam_l2_match).

    q, t: [rows, 128|64] float32 descriptor tables.  seg: [batch,5] int32.
    Returns (idx [n_rows,2] i32, dist [n_rows,2] f32, keep [n_rows] u8 or None).
    """
    lib = _lib.load()
    if q.dim() != 2 or t.dim() != 2 or q.shape[1] != t.shape[1]:
        raise ValueError("l2_match: q and t must be [rows, dim] with equal dim")
    batch = seg.shape[0]
    if n_rows is None:
        n_rows = q.shape[0]
    dev = q.device
    idx = torch.full((n_rows, 2), -1, dtype=torch.int32, device=dev)
    dist = torch.full((n_rows, 2), FLT_MAX, dtype=torch.float32, device=dev)
    keep = torch.zeros((n_rows,), dtype=torch.uint8, device=dev) if mode else None
    null = ctypes.c_void_p(0)
    st = lib.slam_l2_match(
        _dev(q, torch.float32, "q"), _dev(t, torch.float32, "t"), _dev(seg, torch.int32, "seg"), batch, int(max_nq),
        int(max_nt), q.shape[1], int(k), int(mode), float(thr),
        _dev(yq, torch.float32, "yq") if yq is not None else null,
        _dev(yt, torch.float32, "yt") if yt is not None else null,
        ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(dist.data_ptr()),
        ctypes.c_void_p(keep.data_ptr()) if keep is not None else null, _stream())
    _lib.check(st, "slam_l2_match")
    return idx, dist, keep


class Int8Table:
    """An integer-valued float descriptor table packed for the matrix-core L2 matcher:
    desc [n, dim] int8 (value - 128), norm2 [n] int32, bad [1] int32 (entries that were not
    integers in 0..255: the table is usable only when it is 0)."""

    def __init__(self, desc, norm2, bad):
        self.desc, self.norm2, self.bad = desc, norm2, bad

    def usable(self):
        """Synchronises: reads the counter back."""
        return int(self.bad.item()) == 0


def l2_pack_int8(desc):
    """[n, 128|64] float32 -> Int8Table (slam_l2_pack_int8)."""
    lib = _lib.load()
    if desc.dim() != 2:
        raise ValueError("l2_pack_int8: expected [n, dim]")
    n, dim = desc.shape
    dev = desc.device
    out = torch.empty((n, dim), dtype=torch.int8, device=dev)
    norm2 = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    bad = torch.zeros((1,), dtype=torch.int32, device=dev)
    st = lib.slam_l2_pack_int8(_dev(desc, torch.float32, "desc"), n, dim, ctypes.c_void_p(out.data_ptr()),
                               ctypes.c_void_p(norm2.data_ptr()), ctypes.c_void_p(bad.data_ptr()), _stream())
    _lib.check(st, "slam_l2_pack_int8")
    return Int8Table(out, norm2, bad)


def l2_match_int8(q, t, seg, max_nq, max_nt, k=2, mode=0, thr=0.0, yq=None, yt=None, n_rows=None):
    """l2_match on packed tables (q, t: Int8Table; slam_l2_match_int8): same results as l2_match on
    the float tables they were packed from, on the int8 matrix cores."""
    lib = _lib.load()
    if q.desc.shape[1] != t.desc.shape[1]:
        raise ValueError("l2_match_int8: q and t must have equal dim")
    batch = seg.shape[0]
    if n_rows is None:
        n_rows = q.desc.shape[0]
    dev = q.desc.device
    idx = torch.full((n_rows, 2), -1, dtype=torch.int32, device=dev)
    dist = torch.full((n_rows, 2), FLT_MAX, dtype=torch.float32, device=dev)
    keep = torch.zeros((n_rows,), dtype=torch.uint8, device=dev) if mode else None
    null = ctypes.c_void_p(0)
    st = lib.slam_l2_match_int8(
        _dev(q.desc, torch.int8, "q"), _dev(q.norm2, torch.int32, "q.norm2"), _dev(t.desc, torch.int8, "t"),
        _dev(t.norm2, torch.int32, "t.norm2"), _dev(seg, torch.int32, "seg"), batch, int(max_nq), int(max_nt),
        q.desc.shape[1], int(k), int(mode), float(thr),
        _dev(yq, torch.float32, "yq") if yq is not None else null,
        _dev(yt, torch.float32, "yt") if yt is not None else null,
        ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(dist.data_ptr()),
        ctypes.c_void_p(keep.data_ptr()) if keep is not None else null, _stream())
    _lib.check(st, "slam_l2_match_int8")
    return idx, dist, keep


def hamming_knn2(q, t, seg, max_nq, max_nt, desc_bytes, k=2, n_rows=None):
    """Unfiltered batched Hamming k-NN (slam_hamming_knn2)."""
    lib = _lib.load()
    stride = q.shape[1]
    if n_rows is None:
        n_rows = q.shape[0]
    idx = torch.full((n_rows, 2), -1, dtype=torch.int32, device=q.device)
    dist = torch.full((n_rows, 2), INT32_MAX, dtype=torch.int32, device=q.device)
    st = lib.slam_hamming_knn2(_dev(q, torch.uint8, "q"), _dev(t, torch.uint8, "t"), _dev(seg, torch.int32, "seg"),
                               seg.shape[0], int(max_nq), int(max_nt), int(desc_bytes), stride, int(k),
                               ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(dist.data_ptr()), _stream())
    _lib.check(st, "slam_hamming_knn2")
    return idx, dist


def ratio_filter(idx, dist, thr):
    """keep [n] u8 for match tables idx/dist [n,2] i32 (slam_ratio_filter)."""
    lib = _lib.load()
    n = idx.shape[0]
    keep = torch.zeros(n, dtype=torch.uint8, device=idx.device)
    st = lib.slam_ratio_filter(_dev(idx, torch.int32, "idx"), _dev(dist, torch.int32, "dist"), n, float(thr),
                               ctypes.c_void_p(keep.data_ptr()), _stream())
    _lib.check(st, "slam_ratio_filter")
    return keep


def row_filter(y1, y2, thr):
    """keep [n] u8 = |y1 - y2| <= thr in fp64 for paired float32 rows (slam_row_filter)."""
    lib = _lib.load()
    n = y1.shape[0]
    keep = torch.zeros(n, dtype=torch.uint8, device=y1.device)
    st = lib.slam_row_filter(_dev(y1, torch.float32, "y1"), _dev(y2, torch.float32, "y2"), n, float(thr),
                             ctypes.c_void_p(keep.data_ptr()), _stream())
    _lib.check(st, "slam_row_filter")
    return keep


def triangulate_dlt(P, Q, x1, y1, x2, y2):
    """DLT triangulation of n correspondences -> (X, Y, Z f64 [n], valid u8 [n])."""
    lib = _lib.load()
    _, pP = _host_f64(P, 12, "P")
    _, pQ = _host_f64(Q, 12, "Q")
    n = x1.shape[0]
    dev = x1.device
    X = torch.empty(n, dtype=torch.float64, device=dev)
    Y = torch.empty_like(X)
    Z = torch.empty_like(X)
    valid = torch.empty(n, dtype=torch.uint8, device=dev)
    st = lib.slam_triangulate_dlt(pP, pQ, _dev(x1, torch.float32, "x1"), _dev(y1, torch.float32, "y1"),
                                  _dev(x2, torch.float32, "x2"), _dev(y2, torch.float32, "y2"), n,
                                  ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(Y.data_ptr()),
                                  ctypes.c_void_p(Z.data_ptr()), ctypes.c_void_p(valid.data_ptr()), _stream())
    _lib.check(st, "slam_triangulate_dlt")
    return X, Y, Z, valid


def triangulate_matched(P, Q, xq, yq, xt, yt, seg, max_nq, idx, keep, n_rows=None):
    """Triangulate every kept stereo match of a batch of frames.

    Returns (X, Y, Z f64 [n_rows], valid u8 [n_rows], stereo_map i32 [n_rows]).
    """
    lib = _lib.load()
    _, pP = _host_f64(P, 12, "P")
    _, pQ = _host_f64(Q, 12, "Q")
    if n_rows is None:
        n_rows = idx.shape[0]
    dev = idx.device
    X = torch.full((n_rows,), float("nan"), dtype=torch.float64, device=dev)
    Y = torch.full_like(X, float("nan"))
    Z = torch.full_like(X, float("nan"))
    valid = torch.zeros(n_rows, dtype=torch.uint8, device=dev)
    smap = torch.full((n_rows,), -1, dtype=torch.int32, device=dev)
    st = lib.slam_triangulate_matched(
        pP, pQ, _dev(xq, torch.float32, "xq"), _dev(yq, torch.float32, "yq"), _dev(xt, torch.float32, "xt"),
        _dev(yt, torch.float32, "yt"), _dev(seg, torch.int32, "seg"), seg.shape[0], int(max_nq),
        _dev(idx, torch.int32, "idx"), _dev(keep, torch.uint8, "keep"), ctypes.c_void_p(X.data_ptr()),
        ctypes.c_void_p(Y.data_ptr()), ctypes.c_void_p(Z.data_ptr()), ctypes.c_void_p(valid.data_ptr()),
        ctypes.c_void_p(smap.data_ptr()), _stream())
    _lib.check(st, "slam_triangulate_matched")
    return X, Y, Z, valid, smap


class Consensus:
    """SoA consensus tables of a batch of frame pairs (slot = pair * cap + m)."""

    __slots__ = ("rows", "X", "Y", "Z", "uL", "vL", "uR", "vR", "count", "cap", "n_pairs")

    def __init__(self, n_pairs, cap, device):
        n = max(n_pairs * cap, 1)
        self.n_pairs, self.cap = n_pairs, cap
        self.rows = torch.full((n, 4), -1, dtype=torch.int32, device=device)
        self.X = torch.zeros(n, dtype=torch.float64, device=device)
        self.Y = torch.zeros_like(self.X)
        self.Z = torch.zeros_like(self.X)
        self.uL = torch.zeros(n, dtype=torch.float32, device=device)
        self.vL = torch.zeros_like(self.uL)
        self.uR = torch.zeros_like(self.uL)
        self.vR = torch.zeros_like(self.uL)
        self.count = torch.zeros(max(n_pairs, 1), dtype=torch.int32, device=device)


def join_consensus(pair, max_nq, t_idx, t_keep, stereo_map, valid3d, X, Y, Z, xL, yL, xR, yR, cap):
    """Four-image consensus join for a batch of frame pairs (slam_join_consensus).

    pair: [n_pairs,8] int32 (prevL_row0, prevL_count, curL_row0, curL_count, t_row0, curR_row0,
    prevR_row0, 0).
    """
    lib = _lib.load()
    n_pairs = pair.shape[0]
    out = Consensus(n_pairs, int(cap), pair.device)
    p = ctypes.c_void_p
    st = lib.slam_join_consensus(
        _dev(pair, torch.int32, "pair"), n_pairs, int(max_nq), _dev(t_idx, torch.int32, "t_idx"),
        _dev(t_keep, torch.uint8, "t_keep"), _dev(stereo_map, torch.int32, "stereo_map"),
        _dev(valid3d, torch.uint8, "valid3d"), _dev(X, torch.float64, "X"), _dev(Y, torch.float64, "Y"),
        _dev(Z, torch.float64, "Z"), _dev(xL, torch.float32, "xL"), _dev(yL, torch.float32, "yL"),
        _dev(xR, torch.float32, "xR"), _dev(yR, torch.float32, "yR"), int(cap), p(out.rows.data_ptr()),
        p(out.X.data_ptr()), p(out.Y.data_ptr()), p(out.Z.data_ptr()), p(out.uL.data_ptr()), p(out.vL.data_ptr()),
        p(out.uR.data_ptr()), p(out.vR.data_ptr()), p(out.count.data_ptr()), _stream())
    _lib.check(st, "slam_join_consensus")
    return out


def score_hypotheses(Rt, K, t_right, X, Y, Z, uL, vL, uR, vR, thresh=2.0, cap=None, n_dev=None, n_host=None,
                     want_masks=False, work=None, fast=True):
    """Supporter counts of H pose hypotheses per point set (slam_score_hypotheses).

    Rt: [n_sets, H, 3, 4] (or [H,3,4] for one set) f64.  Points: flat SoA with set s
    at [s*cap, s*cap+n_s).  Returns counts [n_sets,H] i32 (and masks [n_sets,H,words] as
    int64 bit patterns when want_masks).  `fast` lets the library use its filtered path
    (needs a scratch tensor `work` of 16*n_sets*cap doubles, allocated here if absent);
    counts are identical with fast=False.
    """
    lib = _lib.load()
    if Rt.dim() == 3:
        Rt = Rt.unsqueeze(0)
    n_sets, H = Rt.shape[0], Rt.shape[1]
    if cap is None:
        cap = X.shape[0] // max(n_sets, 1)
    if n_host is None:
        n_host = cap
    _, pK = _host_f64(K, 9, "K")
    _, pt = _host_f64(t_right, 3, "t_right")
    dev = Rt.device
    counts = torch.zeros((n_sets, H), dtype=torch.int32, device=dev)
    words = (cap + 63) // 64
    masks = torch.zeros((n_sets, H, max(words, 1)), dtype=torch.int64, device=dev) if want_masks else None
    null = ctypes.c_void_p(0)
    if fast and not want_masks and work is None:
        work = torch.empty(16 * max(n_sets, 1) * max(cap, 1), dtype=torch.float64, device=dev)
    st = lib.slam_score_hypotheses(
        _dev(Rt, torch.float64, "Rt"), n_sets, H, pK, pt, _dev(X, torch.float64, "X"), _dev(Y, torch.float64, "Y"),
        _dev(Z, torch.float64, "Z"), _dev(uL, torch.float32, "uL"), _dev(vL, torch.float32, "vL"),
        _dev(uR, torch.float32, "uR"), _dev(vR, torch.float32, "vR"), int(cap),
        _dev(n_dev, torch.int32, "n_dev") if n_dev is not None else null, int(n_host), float(thresh),
        ctypes.c_void_p(counts.data_ptr()), ctypes.c_void_p(masks.data_ptr()) if masks is not None else null,
        max(words, 1) if masks is not None else 0,
        _dev(work, torch.float64, "work") if (fast and work is not None) else null, _stream())
    _lib.check(st, "slam_score_hypotheses")
    return (counts, masks) if want_masks else counts


# ---- batched PnP-RANSAC around score_hypotheses ------------------------------------
def ransac_sample(seed, n_sets, H, n_dev, device=None):
    """[n_sets, H, 4] int32 minimal samples (slam_ransac_sample); n_dev [n_sets] int32 on the GPU."""
    lib = _lib.load()
    out = torch.empty((max(n_sets, 1), max(H, 1), 4), dtype=torch.int32, device=n_dev.device if device is None else device)
    st = lib.slam_ransac_sample(ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), n_sets, H,
                                _dev(n_dev, torch.int32, "n_dev"), 0, ctypes.c_void_p(out.data_ptr()), _stream())
    _lib.check(st, "slam_ransac_sample")
    return out[:n_sets, :H]


def p3p_hypotheses(K, X, Y, Z, uL, vL, cap, n_dev, samples):
    """[n_sets, H, 12] float64 poses from minimal samples (slam_p3p_hypotheses); NaN = no solution."""
    lib = _lib.load()
    n_sets, H = samples.shape[0], samples.shape[1]
    Kh, Kp = _host_f64(K, 9, "K")
    out = torch.empty((max(n_sets, 1), max(H, 1), 12), dtype=torch.float64, device=samples.device)
    st = lib.slam_p3p_hypotheses(Kp, _dev(X, torch.float64, "X"), _dev(Y, torch.float64, "Y"),
                                 _dev(Z, torch.float64, "Z"), _dev(uL, torch.float32, "uL"),
                                 _dev(vL, torch.float32, "vL"), int(cap), _dev(n_dev, torch.int32, "n_dev"), 0,
                                 _dev(samples, torch.int32, "samples"), n_sets, H, 0, H, ctypes.c_void_p(0),
                                 ctypes.c_void_p(out.data_ptr()), _stream())
    _lib.check(st, "slam_p3p_hypotheses")
    return out[:n_sets, :H]


def ransac_select(counts, Rt, n_dev, cap, max_iterations, n_avail=None, want_stopped=False):
    """(best, best_count, best_Rt, iters[, stopped]) of the stopping rule replayed over the first
    n_avail (default all) columns of counts [n_sets, H]."""
    lib = _lib.load()
    n_sets, H = counts.shape
    dev = counts.device
    best = torch.empty(max(n_sets, 1), dtype=torch.int32, device=dev)
    best_count = torch.empty_like(best)
    iters = torch.empty_like(best)
    best_Rt = torch.empty((max(n_sets, 1), 12), dtype=torch.float64, device=dev)
    stopped = torch.empty_like(best)
    st = lib.slam_ransac_select(_dev(counts, torch.int32, "counts"), _dev(Rt, torch.float64, "Rt"), n_sets, H,
                                H if n_avail is None else int(n_avail), _dev(n_dev, torch.int32, "n_dev"), 0, int(cap),
                                int(max_iterations), ctypes.c_void_p(best.data_ptr()),
                                ctypes.c_void_p(best_count.data_ptr()), ctypes.c_void_p(best_Rt.data_ptr()),
                                ctypes.c_void_p(iters.data_ptr()), ctypes.c_void_p(stopped.data_ptr()), _stream())
    _lib.check(st, "slam_ransac_select")
    if want_stopped:
        return best[:n_sets], best_count[:n_sets], best_Rt[:n_sets], iters[:n_sets], stopped[:n_sets]
    return best[:n_sets], best_count[:n_sets], best_Rt[:n_sets], iters[:n_sets]


def refine_poses(K, X, Y, Z, uL, vL, cap, n_dev, masks, Rt0, iterations=30):
    """(Rt [n_sets,12], ok [n_sets]) Levenberg-Marquardt refinement on the masked points (slam_refine_poses)."""
    lib = _lib.load()
    n_sets = Rt0.shape[0]
    Kh, Kp = _host_f64(K, 9, "K")
    out = torch.empty((max(n_sets, 1), 12), dtype=torch.float64, device=Rt0.device)
    ok = torch.zeros(max(n_sets, 1), dtype=torch.int32, device=Rt0.device)
    st = lib.slam_refine_poses(Kp, _dev(X, torch.float64, "X"), _dev(Y, torch.float64, "Y"),
                               _dev(Z, torch.float64, "Z"), _dev(uL, torch.float32, "uL"),
                               _dev(vL, torch.float32, "vL"), int(cap), _dev(n_dev, torch.int32, "n_dev"), 0,
                               _dev(masks, torch.int64, "masks"), masks.shape[-1], _dev(Rt0, torch.float64, "Rt0"),
                               n_sets, int(iterations), ctypes.c_void_p(out.data_ptr()),
                               ctypes.c_void_p(ok.data_ptr()), _stream())
    _lib.check(st, "slam_refine_poses")
    return out[:n_sets], ok[:n_sets]


def refine_winners(K, t_right, X, Y, Z, uL, vL, uR, vR, cap, n_dev, Rt0, thresh=2.0, iterations=30, stereo=False):
    """(Rt [n_sets,12], ok [n_sets], ref_count [n_sets]): the supporters of Rt0 found on the
    device, the refinement on them, and the supporter count of the result (slam_refine_winners).
    stereo=True adds the right image's residuals to the refinement cost
    (slam_refine_winners_stereo; the reference minimises the left image's alone)."""
    lib = _lib.load()
    entry = lib.slam_refine_winners_stereo if stereo else lib.slam_refine_winners
    n_sets = Rt0.shape[0]
    Kh, Kp = _host_f64(K, 9, "K")
    th, tp = _host_f64(t_right, 3, "t_right")
    dev = Rt0.device
    out = torch.empty((max(n_sets, 1), 12), dtype=torch.float64, device=dev)
    ok = torch.zeros(max(n_sets, 1), dtype=torch.int32, device=dev)
    ref_count = torch.zeros(max(n_sets, 1), dtype=torch.int32, device=dev)
    st = entry(Kp, tp, float(thresh), _dev(X, torch.float64, "X"), _dev(Y, torch.float64, "Y"),
               _dev(Z, torch.float64, "Z"), _dev(uL, torch.float32, "uL"), _dev(vL, torch.float32, "vL"),
               _dev(uR, torch.float32, "uR"), _dev(vR, torch.float32, "vR"), int(cap), _dev(n_dev, torch.int32, "n_dev"), 0,
               _dev(Rt0, torch.float64, "Rt0"), n_sets, int(iterations), ctypes.c_void_p(out.data_ptr()),
               ctypes.c_void_p(ok.data_ptr()), ctypes.c_void_p(ref_count.data_ptr()), _stream())
    _lib.check(st, "slam_refine_winners_stereo" if stereo else "slam_refine_winners")
    return out[:n_sets], ok[:n_sets], ref_count[:n_sets]


def track_assign(inl_rows, inl_loc, n_inl, kp_cap, first_frame=0):
    """Track ids and the track-point table of consecutive frame pairs (slam_track_assign).

    inl_rows [cap,4] int32, inl_loc [cap,6] float32: dense inlier tables; n_inl [P] int32; every
    keypoint index < kp_cap.  Returns a dict of device tensors: row_track, row_depth [cap],
    track_start [cap+1], pt_kp, pt_frame [2 cap], pt_loc [2 cap,3], summary [8] (rows, tracks,
    points, longest track, flags).  Nothing is synchronised."""
    lib = _lib.load()
    cap, P = inl_rows.shape[0], n_inl.shape[0]
    dev = inl_rows.device
    i32 = lambda n: torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    out = dict(row_track=i32(cap), row_depth=i32(cap), track_start=i32(cap + 1), pt_kp=i32(2 * cap), pt_frame=i32(2 * cap),
               pt_loc=torch.empty((max(2 * cap, 1), 3), dtype=torch.float32, device=dev), summary=i32(8))
    nbytes = int(lib.slam_track_assign_work_bytes(P, cap, int(kp_cap)))
    work = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = lib.slam_track_assign(_dev(inl_rows, torch.int32, "inl_rows"), _dev(inl_loc, torch.float32, "inl_loc"),
                               _dev(n_inl, torch.int32, "n_inl"), P, cap, int(kp_cap), int(first_frame),
                               p(out["row_track"]), p(out["row_depth"]), p(out["track_start"]), p(out["pt_kp"]),
                               p(out["pt_frame"]), p(out["pt_loc"]), p(out["summary"]), p(work), nbytes, _stream())
    _lib.check(st, "slam_track_assign")
    out["work"] = work                 # alive until the stream has run the kernels
    return out
