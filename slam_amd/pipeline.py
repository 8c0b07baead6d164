"""Device-resident, batched front-end over a contiguous range of stereo frames.

This is the fast path: where the reference walks frame by frame through Python
(track_camera_for_many_images, utils/utils.py:236-294), a frame range is laid out
once as structure-of-arrays tables in HBM and each stage is ONE kernel launch over
the whole range:

  1. stereo matching L_f -> R_f (k = 1) + row filter      utils/utils.py:302-315
  2. DLT triangulation of every kept stereo match          utils/utils.py:318-335
  3. temporal matching L_f -> L_{f+1} (k = 2) + ratio test  models/Matcher.py:187-204
  4. four-image consensus join per pair (f, f+1)           utils/utils.py:338-386
  5. supporter counting of H pose hypotheses per pair      utils/utils.py:389-410
  6. first-best winner, its supporter mask, inlier tables    utils/utils.py:438-442,455,277-278

With `FrontEnd.run_ransac` the hypotheses themselves are made on the device too (minimal
samples, P3P, the reference's stopping rule, refinement of the winner: utils/utils.py:413-464).

Nothing is copied back to the host between stages.  Buffers are allocated once per
(frame range, H) by `FrontEnd.prepare` and reused by every `run`.
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops, sharding


class FrameTables:
    """SoA keypoint / descriptor tables of F stereo frames in HBM.

    descL/descR: [rows, 32|64] uint8, zero padded (or [rows, 128|64] float32); xL,yL,xR,yR: [rows] float32;
    offL/offR: host int64 [F+1] row offsets of each frame's left/right keypoints.
    """

    def __init__(self, descL, descR, xL, yL, xR, yR, offL, offR, desc_bytes):
        # a table without a single row still gets an address the library can be handed (one unused row)
        ne = lambda t: t if t.shape[0] else torch.zeros((1,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self.descL, self.descR = ne(descL), ne(descR)
        self.xL, self.yL, self.xR, self.yR = ne(xL), ne(yL), ne(xR), ne(yR)
        self.offL = np.asarray(offL, dtype=np.int64)
        self.offR = np.asarray(offR, dtype=np.int64)
        self.desc_bytes = int(desc_bytes)
        self.n_frames = len(self.offL) - 1
        self.device = descL.device
        # Float descriptors that hold integers 0..255 (cv2's SIFT output) are also kept packed for
        # the int8 matrix-core L2 matcher (ops.Int8Table); None: use the fp32 matcher.
        self.i8L = self.i8R = None
        if descL.dtype == torch.float32:
            i8L, i8R = ops.l2_pack_int8(descL), ops.l2_pack_int8(descR)
            if i8L.usable() and i8R.usable():
                self.i8L, self.i8R = i8L, i8R

    @classmethod
    def from_frames(cls, frames, device):
        """frames: objects with kpL/kpR [N,2] float32 and descL/descR [N,D] uint8 (binary
        descriptors, Hamming norm) or [N,128|64] float32 (SIFT-like, L2 norm)."""
        D = frames[0].descL.shape[1]
        is_float = frames[0].descL.dtype.kind == "f"
        stride = 32 if D <= 32 else 64
        if is_float and D not in (64, 128):
            raise ValueError("float descriptors must have 128 or 64 dimensions")
        if not is_float and D > 64:
            raise ValueError("binary descriptors longer than 64 bytes are not supported")

        def table(arrs):
            if is_float:
                raw = np.ascontiguousarray(np.concatenate(arrs, axis=0), dtype=np.float32)
                return torch.from_numpy(raw).to(device)
            raw = np.ascontiguousarray(np.concatenate(arrs, axis=0))
            return ops.pack_descriptors(torch.from_numpy(raw).to(device), D, stride)

        def col(side, c):
            a = np.concatenate([getattr(f, side)[:, c] for f in frames]).astype(np.float32)
            return torch.from_numpy(np.ascontiguousarray(a)).to(device)

        offL = np.concatenate([[0], np.cumsum([f.descL.shape[0] for f in frames])])
        offR = np.concatenate([[0], np.cumsum([f.descR.shape[0] for f in frames])])
        return cls(table([f.descL for f in frames]), table([f.descR for f in frames]), col("kpL", 0), col("kpL", 1),
                   col("kpR", 0), col("kpR", 1), offL, offR, D)


    @classmethod
    def from_detector_output(cls, per_frame, device):
        """Ingestion of what a detector hands over, frame by frame: per_frame[i] =
        ((keypoints_left, descriptors_left), (keypoints_right, descriptors_right)), exactly the two
        pairs `algo.detectAndCompute` returns for a stereo frame and the matcher cache keeps under
        LEFT / RIGHT (models/Matcher.py:116-119).  keypoints: a sequence of objects with `.pt`
        (cv2.KeyPoint) or an [N,2] array; descriptors: [N,D] uint8 -- ORB 32 bytes, AKAZE 61 -- or
        [N,128|64] float32, or None for an image without keypoints (cv2 returns None there)."""
        from .types import keypoints_to_array

        class _Frame:
            __slots__ = ("kpL", "kpR", "descL", "descR")

        width = dtype = None
        for sides in per_frame:
            for _, d in sides:
                if d is not None and len(d):
                    d = np.asarray(d)
                    if d.ndim != 2:
                        raise ValueError("descriptors must be [N, D]")
                    if width is None:
                        width, dtype = d.shape[1], (np.float32 if d.dtype.kind == "f" else np.uint8)
                    elif d.shape[1] != width:
                        raise ValueError(f"descriptor widths differ: {width} vs {d.shape[1]}")
        if width is None:
            raise ValueError("no descriptors in any frame")
        frames = []
        for (kl, dl), (kr, dr) in per_frame:
            f = _Frame()
            for side, k, d in (("L", kl, dl), ("R", kr, dr)):
                xy = keypoints_to_array(k) if k is not None and len(k) else np.zeros((0, 2), np.float32)
                desc = np.ascontiguousarray(d, dtype=dtype) if d is not None and len(d) else np.zeros((0, width), dtype)
                if xy.shape[0] != desc.shape[0]:
                    raise ValueError(f"{xy.shape[0]} keypoints but {desc.shape[0]} descriptors")
                setattr(f, "kp" + side, xy)
                setattr(f, "desc" + side, desc)
            frames.append(f)
        return cls.from_frames(frames, device)

    @classmethod
    def from_matcher_cache(cls, cache, frame_ids, device):
        """The same from a Matcher's cache dict (models/Matcher.py:51,118-119; also in the serialised
        form of get_matcher_cache, keypoints as dicts): frames in the order of frame_ids."""
        from .constants import LEFT, RIGHT

        def side(entry):
            kps, dsc = entry
            if len(kps) and isinstance(kps[0], dict):
                kps = np.array([k["pt"] for k in kps], dtype=np.float32).reshape(-1, 2)
            return kps, dsc
        return cls.from_detector_output([(side(cache[f][LEFT]), side(cache[f][RIGHT])) for f in frame_ids], device)


class Workspace:
    """Every buffer one FrontEnd.run touches, allocated once."""


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


class FrontEnd:
    def __init__(self, k, m1, m2, row_thresh=2.0, ratio_thresh=1.0, pix_thresh=2.0):
        """k, m1, m2 as returned by read_cameras (utils/utils.py:75-86)."""
        self.k = np.ascontiguousarray(k, dtype=np.float64)
        self.P = np.ascontiguousarray(self.k @ np.asarray(m1, dtype=np.float64))   # utils/utils.py:325
        self.Q = np.ascontiguousarray(self.k @ np.asarray(m2, dtype=np.float64))
        self.t_right = np.ascontiguousarray(np.asarray(m2, dtype=np.float64)[:, 3])  # utils/utils.py:396
        self.row_thresh, self.ratio_thresh, self.pix_thresh = float(row_thresh), float(ratio_thresh), float(pix_thresh)
        self.lib = _lib.load()

    def prepare(self, tb, n_hyp=0, want_masks=False, table_pairs=None, table_rows=None, pairs=None):
        """Allocate outputs and index tables for the frame range held by `tb`.

        pairs: [P,2] frame indices (reference, candidate) to run stages 3-6 on; default: the
        consecutive pairs (f, f+1) of the visual-odometry loop.  Arbitrary pairs are what
        loop closure asks for (models/LoopClosure.py:112-139: any two keyframes).
        table_pairs / table_rows: capacity of the table arena that travels in the multi-GPU
        gather, when the ranks agreed on a common shape (sharding.agree_table_shape);
        default: this range's own."""
        dev = tb.device
        F = tb.n_frames
        nL = np.diff(tb.offL)
        nR = np.diff(tb.offR)
        ws = Workspace()
        if pairs is None:
            fa = np.arange(max(F - 1, 0), dtype=np.int64)
            fb = fa + 1
            t_off = tb.offL[:-2] if F >= 2 else np.zeros(0, dtype=np.int64)   # one temporal row per left keypoint
            t_rows = int(tb.offL[-1])
        else:
            pr = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
            if pr.size and (pr.min() < 0 or pr.max() >= F):
                raise ValueError("pairs: frame index out of range")
            fa, fb = pr[:, 0], pr[:, 1]
            t_off = np.concatenate([[0], np.cumsum(nL[fa])[:-1]]).astype(np.int64) if len(fa) else np.zeros(0, np.int64)
            t_rows = int(nL[fa].sum())
        ws.pairs_ab = np.stack([fa, fb], axis=1)
        ws.tb, ws.F, ws.P, ws.H = tb, F, len(fa), int(n_hyp)
        ws.max_nL, ws.max_nR = int(nL.max()) if F else 0, int(nR.max()) if F else 0
        rows = int(tb.offL[-1])
        ws.rows = rows
        ws.seg_s = ops.make_segments(tb.offL[:-1], nL, tb.offR[:-1], nR, tb.offL[:-1], dev)
        ws.seg_t = ops.make_segments(tb.offL[fa], nL[fa], tb.offL[fb], nL[fb], t_off, dev)
        ws.t_off = np.asarray(t_off, dtype=np.int64)
        pair = np.stack([tb.offL[fa], nL[fa], tb.offL[fb], nL[fb], t_off, tb.offR[fb], tb.offR[fa],
                         np.zeros(len(fa), dtype=np.int64)], axis=1)
        if pair.size and pair.max() > 2147483647:
            raise ValueError("row offsets exceed int32")
        ws.pair = torch.from_numpy(pair.astype(np.int32).reshape(-1, 8)).to(dev)
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        u8 = lambda *s: torch.empty(s, dtype=torch.uint8, device=dev)
        f64 = lambda *s: torch.empty(s, dtype=torch.float64, device=dev)
        r = max(rows, 1)
        dist_t = torch.float32 if tb.descL.dtype == torch.float32 else torch.int32
        ws.s_idx, ws.s_dist, ws.s_keep = i32(r, 2), torch.empty((r, 2), dtype=dist_t, device=dev), u8(r)
        ws.X, ws.Y, ws.Z, ws.valid, ws.smap = f64(r), f64(r), f64(r), u8(r), i32(r)
        rt = max(t_rows, 1)
        ws.t_idx, ws.t_dist, ws.t_keep = i32(rt, 2), torch.empty((rt, 2), dtype=dist_t, device=dev), u8(rt)
        ws.cap = int(nL[fa].max()) if len(fa) else 0
        ws.cons = ops.Consensus(ws.P, ws.cap, dev)
        ws.counts = torch.zeros((max(ws.P, 1), max(ws.H, 1)), dtype=torch.int32, device=dev)
        ws.mask_words = (ws.cap + 63) // 64
        ws.masks = (torch.zeros((max(ws.P, 1), max(ws.H, 1), max(ws.mask_words, 1)), dtype=torch.int64, device=dev)
                    if want_masks else None)
        ws.score_work = f64(16 * max(ws.P, 1) * max(ws.cap, 1))    # the size include/slam_amd.h asks for
        ws.work_ready = 0          # 1 once the join has laid out score_work for the current consensus sets
        # winner selection + inlier tables
        np_ = max(ws.P, 1)
        ws.best, ws.best_count = i32(np_), i32(np_)
        ws.final_count = torch.zeros((np_, 1), dtype=torch.int32, device=dev)
        ws.final_mask = torch.zeros((np_, 1, max(ws.mask_words, 1)), dtype=torch.int64, device=dev)
        ws.inl_off = i32(np_)
        # everything rank 0 needs from this range lives in one arena (sharding.TableLayout)
        ws.layout = sharding.TableLayout(max(np_, table_pairs or 0), max(ws.P * ws.cap, 1, table_rows or 0))
        ws.arena = torch.zeros(ws.layout.nbytes, dtype=torch.uint8, device=dev)
        ws.table_views = ws.layout.views(ws.arena)
        ws.best_Rt = ws.table_views["pose_a"][:np_]
        ws.pose = ws.table_views["pose_b"][:np_]
        ws.inl_rows = ws.table_views["rows"]
        ws.inl_loc = ws.table_views["loc"]
        ws.n_inl = ws.table_views["n_inl"][:np_]
        # host constants must outlive the calls
        ws.hP = self.P.ctypes.data_as(ctypes.c_void_p)
        ws.hQ = self.Q.ctypes.data_as(ctypes.c_void_p)
        ws.hK = self.k.ctypes.data_as(ctypes.c_void_p)
        ws.hT = self.t_right.ctypes.data_as(ctypes.c_void_p)
        return ws

    # -- the five launches ---------------------------------------------------
    def _match(self, q, t, seg, batch, max_nq, max_nt, desc_bytes, k, mode, thr, yq, yt, idx, dist, keep, stream, what,
               q8=None, t8=None):
        if q8 is not None and t8 is not None:   # integer-valued float descriptors: L2 on the int8 matrix cores
            st = self.lib.slam_l2_match_int8(_p(q8.desc), _p(q8.norm2), _p(t8.desc), _p(t8.norm2), _p(seg), batch,
                                             max_nq, max_nt, q8.desc.shape[1], k, mode, thr, _p(yq), _p(yt), _p(idx),
                                             _p(dist), _p(keep), stream)
        elif q.dtype == torch.float32:    # float descriptors: L2 norm, as cv2.BFMatcher() on SIFT
            st = self.lib.slam_l2_match(_p(q), _p(t), _p(seg), batch, max_nq, max_nt, q.shape[1], k, mode, thr,
                                        _p(yq), _p(yt), _p(idx), _p(dist), _p(keep), stream)
        else:
            st = self.lib.slam_hamming_match(_p(q), _p(t), _p(seg), batch, max_nq, max_nt, desc_bytes, q.shape[1], k,
                                             mode, thr, _p(yq), _p(yt), _p(idx), _p(dist), _p(keep), stream)
        _lib.check(st, what)

    def stereo_match(self, ws, stream):
        tb = ws.tb
        return self._match(tb.descL, tb.descR, ws.seg_s, ws.F, ws.max_nL, ws.max_nR, tb.desc_bytes, 1, 2,
                           self.row_thresh, tb.yL, tb.yR, ws.s_idx, ws.s_dist, ws.s_keep, stream, "match(stereo)",
                           q8=tb.i8L, t8=tb.i8R)

    def triangulate(self, ws, stream):
        tb = ws.tb
        st = self.lib.slam_triangulate_matched(ws.hP, ws.hQ, _p(tb.xL), _p(tb.yL), _p(tb.xR), _p(tb.yR), _p(ws.seg_s),
                                               ws.F, ws.max_nL, _p(ws.s_idx), _p(ws.s_keep), _p(ws.X), _p(ws.Y),
                                               _p(ws.Z), _p(ws.valid), _p(ws.smap), stream)
        _lib.check(st, "slam_triangulate_matched")

    def temporal_match(self, ws, stream):
        tb = ws.tb
        return self._match(tb.descL, tb.descL, ws.seg_t, ws.P, ws.max_nL, ws.max_nL, tb.desc_bytes, 2, 1,
                           self.ratio_thresh, None, None, ws.t_idx, ws.t_dist, ws.t_keep, stream, "match(temporal)",
                           q8=tb.i8L, t8=tb.i8L)

    def join(self, ws, stream):
        tb, c = ws.tb, ws.cons
        # the join also lays out the scoring workspace of its consensus sets (unless masks are
        # wanted: that scoring path reads the plain tables)
        work = ws.score_work if ws.masks is None else None
        st = self.lib.slam_join_consensus_prepared(_p(ws.pair), ws.P, ws.max_nL, _p(ws.t_idx), _p(ws.t_keep),
                                                   _p(ws.smap), _p(ws.valid), _p(ws.X), _p(ws.Y), _p(ws.Z), _p(tb.xL),
                                                   _p(tb.yL), _p(tb.xR), _p(tb.yR), ws.cap, _p(c.rows), _p(c.X), _p(c.Y),
                                                   _p(c.Z), _p(c.uL), _p(c.vL), _p(c.uR), _p(c.vR), _p(c.count), ws.hK,
                                                   self.pix_thresh, _p(work), stream)
        _lib.check(st, "slam_join_consensus_prepared")
        ws.work_ready = 1 if work is not None else 0

    def score(self, ws, Rt, stream):
        """Rt: [P, H, 3, 4] (or [P,H,12]) float64 on the device."""
        c = ws.cons
        st = self.lib.slam_score_hypotheses_range(_p(Rt), ws.P, ws.H, 0, ws.H, None, ws.hK, ws.hT, _p(c.X), _p(c.Y),
                                                  _p(c.Z), _p(c.uL), _p(c.vL), _p(c.uR), _p(c.vR), ws.cap, _p(c.count),
                                                  0, self.pix_thresh, _p(ws.counts), _p(ws.masks),
                                                  ws.mask_words if ws.masks is not None else 0, _p(ws.score_work),
                                                  ws.work_ready, stream)
        _lib.check(st, "slam_score_hypotheses")

    def select(self, ws, Rt, stream, dense=True):
        """Winner per pair (first best), its supporters (the reference's final find_supporters,
        utils/utils.py:455) and the order-preserving inlier tables.  dense=True packs the pairs
        back to back and tests the winner's supporters inside the table kernel; dense=False
        scores the winner once more with a mask and leaves each pair in its own slot."""
        st = self.lib.slam_select_best(_p(ws.counts), _p(Rt), ws.P, ws.H, _p(ws.best), _p(ws.best_count),
                                       _p(ws.best_Rt), stream)
        _lib.check(st, "slam_select_best")
        if dense:           # supporter test of the winner, compaction and packing in one launch
            c, tb = ws.cons, ws.tb
            st = self.lib.slam_winner_tables(_p(ws.pair), ws.P, _p(c.rows), _p(c.count), ws.cap, ws.hK, ws.hT,
                                             self.pix_thresh, _p(ws.best_Rt), _p(ws.best_count), _p(c.X), _p(c.Y),
                                             _p(c.Z), _p(c.uL), _p(c.vL), _p(c.uR), _p(c.vR), _p(tb.xL), _p(tb.yL),
                                             _p(tb.xR), _p(ws.inl_rows), _p(ws.inl_loc), _p(ws.n_inl), stream)
            _lib.check(st, "slam_winner_tables")
            return
        self._score_one(ws, ws.best_Rt, stream)
        self._tables(ws, stream, dense=False)

    # -- device-side RANSAC ----------------------------------------------------
    def prepare_ransac(self, ws):
        """Buffers for run_ransac (hypotheses are generated on the device)."""
        if ws.H <= 0:
            raise ValueError("prepare(..., n_hyp=H) with H > 0 is needed for RANSAC")
        dev = ws.tb.device
        np_ = max(ws.P, 1)
        ws.samples = torch.empty((np_, ws.H, 4), dtype=torch.int32, device=dev)
        ws.Rt_gen = torch.empty((np_, ws.H, 12), dtype=torch.float64, device=dev)
        ws.iters = torch.zeros(np_, dtype=torch.int32, device=dev)
        ws.stopped = torch.zeros(np_, dtype=torch.int32, device=dev)
        ws.ref_Rt = torch.empty((np_, 12), dtype=torch.float64, device=dev)
        ws.ref_ok = torch.zeros(np_, dtype=torch.int32, device=dev)
        ws.ratio = torch.empty(np_, dtype=torch.float64, device=dev)
        ws.found = torch.zeros(np_, dtype=torch.int32, device=dev)
        return ws

    def _score_one(self, ws, pose, stream):
        """Supporter count and mask of one pose per pair (find_supporters of a single Rt)."""
        c = ws.cons
        st = self.lib.slam_score_hypotheses(_p(pose), ws.P, 1, ws.hK, ws.hT, _p(c.X), _p(c.Y), _p(c.Z), _p(c.uL),
                                            _p(c.vL), _p(c.uR), _p(c.vR), ws.cap, _p(c.count), 0, self.pix_thresh,
                                            _p(ws.final_count), _p(ws.final_mask), max(ws.mask_words, 1), None,
                                            stream)
        _lib.check(st, "slam_score_hypotheses(single)")

    def _tables(self, ws, stream, dense=True):
        c, tb = ws.cons, ws.tb
        if dense:
            # exclusive prefix sum of the supporter counts: device-side plumbing
            torch.cumsum(ws.final_count.view(-1), 0, dtype=torch.int32, out=ws.inl_off)
            ws.inl_off.sub_(ws.final_count.view(-1))
        st = self.lib.slam_build_inlier_tables(_p(ws.pair), ws.P, _p(c.rows), _p(c.count), ws.cap, _p(ws.final_mask),
                                               max(ws.mask_words, 1), _p(tb.xL), _p(tb.yL), _p(tb.xR),
                                               _p(ws.inl_off) if dense else None, _p(ws.inl_rows), _p(ws.inl_loc),
                                               _p(ws.n_inl), stream)
        _lib.check(st, "slam_build_inlier_tables")

    def ransac(self, ws, seed, stream, max_iterations=None, refine_iterations=30, first_slice=128, stereo_refine=False):
        """PnP-RANSAC of every pair on the device (utils/utils.py:413-464): H minimal samples
        per pair from a generator keyed by (seed, pair, hypothesis), P3P, supporter counts, the
        stopping rule replayed over the counts, refinement of the winner on its supporters,
        the refined pose's supporters as the inlier tables.  first_slice: where the stopping rule is
        consulted -- after that many hypotheses of every pair, then (a tuple) after each further
        bound for the pairs still running, then after the rest (0: score all at once).  stereo_refine:
        the refinement also minimises the right image's residuals (not the reference's cost, which is
        the left image's alone at utils/utils.py:452; it removes that cost's heading bias).  Results: ws.pose [P,12],
        ws.ratio [P], ws.found [P], ws.iters [P], ws.best_count [P], the inlier tables."""
        c = ws.cons
        max_it = ws.H if max_iterations is None else int(max_iterations)
        lib = self.lib
        _lib.check(lib.slam_ransac_sample(ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), ws.P, ws.H, _p(c.count), 0,
                                          _p(ws.samples), stream), "slam_ransac_sample")

        def solve_range(h0, hn, skip):
            st = lib.slam_p3p_hypotheses(ws.hK, _p(c.X), _p(c.Y), _p(c.Z), _p(c.uL), _p(c.vL), ws.cap, _p(c.count), 0,
                                         _p(ws.samples), ws.P, ws.H, h0, hn, _p(skip), _p(ws.Rt_gen), stream)
            _lib.check(st, "slam_p3p_hypotheses")

        def score_range(h0, hn, skip):
            # (the second slice reuses the point records the first one prepared)
            st = lib.slam_score_hypotheses_range(_p(ws.Rt_gen), ws.P, ws.H, h0, hn, _p(skip), ws.hK, ws.hT, _p(c.X),
                                                 _p(c.Y), _p(c.Z), _p(c.uL), _p(c.vL), _p(c.uR), _p(c.vR), ws.cap,
                                                 _p(c.count), 0, self.pix_thresh, _p(ws.counts), None, 0,
                                                 _p(ws.score_work), 1 if h0 > 0 else ws.work_ready, stream)
            _lib.check(st, "slam_score_hypotheses_range")

        def select(n_avail):
            st = lib.slam_ransac_select(_p(ws.counts), _p(ws.Rt_gen), ws.P, ws.H, n_avail, _p(c.count), 0, ws.cap,
                                        max_it, _p(ws.best), _p(ws.best_count), _p(ws.best_Rt), _p(ws.iters),
                                        _p(ws.stopped), stream)
            _lib.check(st, "slam_ransac_select")

        # The stopping rule usually ends the loop within the first few dozen hypotheses (about
        # 6 at 98 % inliers, about 100 at 50 %): score a first slice for every pair, apply the
        # rule, and go on to the next, larger slice only for the pairs it has not stopped yet.  The
        # outcome is the one of scoring everything: the rule never looks past the iteration it stops at.
        n_used = min(ws.H, max_it)
        bounds = first_slice if isinstance(first_slice, (tuple, list)) else ((first_slice,) if first_slice else ())
        bounds = sorted({min(int(b), n_used) for b in bounds if int(b) > 0} | {n_used})
        done = 0
        for b in bounds:
            skip = ws.stopped if done > 0 else None
            solve_range(done, b - done, skip)
            score_range(done, b - done, skip)
            select(b)
            done = b
        # supporters of the winner, refinement on them and the refined pose's supporter count in
        # one launch; the tables are the refined pose's (the reference reports its mask and ratio
        # whether or not the refined pose is adopted)
        refine = lib.slam_refine_winners_stereo if stereo_refine else lib.slam_refine_winners
        _lib.check(refine(ws.hK, ws.hT, self.pix_thresh, _p(c.X), _p(c.Y), _p(c.Z), _p(c.uL), _p(c.vL), _p(c.uR), _p(c.vR),
                          ws.cap, _p(c.count), 0, _p(ws.best_Rt), ws.P, int(refine_iterations), _p(ws.ref_Rt),
                          _p(ws.ref_ok), _p(ws.final_count), stream), "slam_refine_winners")
        _lib.check(lib.slam_ransac_adopt(ws.P, _p(c.count), 0, ws.cap, _p(ws.best_count), _p(ws.final_count),
                                         _p(ws.best_Rt), _p(ws.ref_Rt), _p(ws.pose), _p(ws.ratio), _p(ws.found),
                                         stream), "slam_ransac_adopt")
        tb = ws.tb
        _lib.check(lib.slam_winner_tables(_p(ws.pair), ws.P, _p(c.rows), _p(c.count), ws.cap, ws.hK, ws.hT,
                                          self.pix_thresh, _p(ws.ref_Rt), _p(ws.final_count), _p(c.X), _p(c.Y), _p(c.Z),
                                          _p(c.uL), _p(c.vL), _p(c.uR), _p(c.vR), _p(tb.xL), _p(tb.yL), _p(tb.xR),
                                          _p(ws.inl_rows), _p(ws.inl_loc), _p(ws.n_inl), stream), "slam_winner_tables")

    def run_ransac(self, ws, seed, max_iterations=None, first_slice=128, stereo_refine=False):
        """Stages 1-4 followed by the device-side RANSAC, all on the current stream."""
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if ws.F == 0:
            return ws
        if not hasattr(ws, "samples"):
            self.prepare_ransac(ws)
        self.stereo_match(ws, stream)
        self.triangulate(ws, stream)
        if ws.P > 0:
            self.temporal_match(ws, stream)
            self.join(ws, stream)
            self.ransac(ws, seed, stream, max_iterations=max_iterations, first_slice=first_slice,
                        stereo_refine=stereo_refine)
        return ws

    def run(self, ws, Rt=None):
        """Enqueue the whole front-end for the frame range on the current stream."""
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if ws.F == 0:
            return ws
        self.stereo_match(ws, stream)
        self.triangulate(ws, stream)
        if ws.P > 0:
            self.temporal_match(ws, stream)
            self.join(ws, stream)
            if Rt is not None and ws.H > 0:
                if Rt.dtype != torch.float64 or not Rt.is_contiguous() or Rt.numel() != ws.P * ws.H * 12:
                    raise ValueError("Rt must be a contiguous float64 tensor of shape [P, H, 3, 4]")
                self.score(ws, Rt, stream)
                self.select(ws, Rt, stream)
        return ws


def capture(fe, ws, Rt=None):
    """Record one `fe.run(ws, Rt)` into a HIP graph (torch.cuda.CUDAGraph).  `graph.replay()`
    re-issues every launch of the front-end with one host call; the buffers of `ws`, `ws.tb`
    and `Rt` are baked in, so new frames are fed by writing into those tensors in place."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fe.run(ws, Rt)                      # warm-up outside the capture (lazy module loads)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    # thread-local capture mode: helper threads of the process (e.g. the collective library's
    # watchdog polling its events) must not invalidate the recording
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        fe.run(ws, Rt)
    return graph


def capture_ransac(fe, ws, seed, **kw):
    """Record one `fe.run_ransac(ws, seed, **kw)` -- matching, triangulation, join and the whole
    device-side RANSAC -- into a HIP graph.  None of its launches waits for the host (the slices
    after the first are skipped on the device for the pairs the stopping rule has ended), so one
    `graph.replay()` replaces about twenty launches: what a caller with one or a few frame pairs at a
    time is bound by.  Buffers and the seed are baked in; feed new frames by writing into `ws.tb`."""
    if not hasattr(ws, "samples"):
        fe.prepare_ransac(ws)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fe.run_ransac(ws, seed, **kw)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        fe.run_ransac(ws, seed, **kw)
    return graph


def consensus_to_host(ws):
    """Copy the per-pair consensus tables back: list of dicts (rows [M,4], X [M,3], ...)."""
    c = ws.cons
    count = c.count.cpu().numpy()[:ws.P]
    if (count > ws.cap).any():
        raise _lib.SlamError("consensus table overflow: a pair produced more rows than cap")
    rows = c.rows.cpu().numpy()
    X, Y, Z = c.X.cpu().numpy(), c.Y.cpu().numpy(), c.Z.cpu().numpy()
    uL, vL, uR, vR = (t.cpu().numpy() for t in (c.uL, c.vL, c.uR, c.vR))
    out = []
    for p in range(ws.P):
        s = slice(p * ws.cap, p * ws.cap + int(count[p]))
        out.append({"rows": rows[s], "X": np.stack([X[s], Y[s], Z[s]], axis=1), "uvL": np.stack([uL[s], vL[s]], axis=1),
                    "uvR": np.stack([uR[s], vR[s]], axis=1)})
    return out
