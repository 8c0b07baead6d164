"""Frame-range sharding across the GPUs of one node, and the one exchange step.

The reference is a single process (utils/utils.py:236-294).  Its per-pair work (two
matchings, triangulation, join, hypothesis scoring) is independent across frame pairs,
so a sequence is cut into contiguous ranges of pairs, one range per rank, each rank
also holding the first frame of the next range (a one-frame halo, no communication).
The only exchange is a gather of the per-pair inlier tables and poses to rank 0
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU
tests), where the track database is rebuilt by replaying the pairs in frame order --
track ids are handed out in arrival order, so that tail is sequential by construction.
"""
import gc

import numpy as np
import torch
import torch.distributed as dist

from .track_database import TrackDatabase


def shard_pairs(n_frames, world):
    """Contiguous ranges of frame pairs: rank r owns pairs [s_r, e_r) and needs frames
    s_r .. e_r inclusive.  Earlier ranks take the remainder."""
    n_pairs = max(n_frames - 1, 0)
    base, rem = divmod(n_pairs, world)
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < rem else 0)
        out.append((s, e))
        s = e
    return out


class TableLayout:
    """Byte layout of the per-rank table arena that travels in the gather: everything rank 0
    needs from a shard, back to back in one buffer of a size all ranks agree on, so that the
    exchange is ONE fixed-size collective per step without any size negotiation or host sync.

        pose_a [pairs,12] f64   winner of a given hypothesis set
        pose_b [pairs,12] f64   result of the device-side RANSAC
        rows   [rows,4]  i32    inlier rows (prev_L, prev_R, cur_L, cur_R), dense in pair order
        loc    [rows,6]  f32    (xl, xr, yl) of both frames per inlier row
        n_inl  [pairs]   i32    inlier rows per pair
        n_cons [pairs]   i32    consensus rows per pair
    """
    FIELDS = (("pose_a", torch.float64, 12, True), ("pose_b", torch.float64, 12, True),
              ("rows", torch.int32, 4, False), ("loc", torch.float32, 6, False),
              ("n_inl", torch.int32, 1, True), ("n_cons", torch.int32, 1, True))

    def __init__(self, pairs, rows):
        self.pairs, self.rows = max(int(pairs), 1), max(int(rows), 1)
        self.offset, off = {}, 0
        for name, dtype, width, per_pair in self.FIELDS:
            n = (self.pairs if per_pair else self.rows) * width * torch.empty((), dtype=dtype).element_size()
            self.offset[name] = (off, n)
            off += (n + 15) // 16 * 16
        self.nbytes = off

    def views(self, arena):
        """name -> typed view of a uint8 arena of this layout."""
        out = {}
        for name, dtype, width, per_pair in self.FIELDS:
            off, n = self.offset[name]
            v = arena[off:off + n].view(dtype)
            out[name] = v.reshape(-1, width) if width > 1 else v
        return out


def agree_table_shape(pairs, rows, group=None, device=None):
    """(max pairs, max row capacity) over all ranks: the common arena shape (one all_reduce,
    done once before the workspaces are allocated)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(pairs), int(rows)
    on_gpu = dist.get_backend(group) != "gloo" and device is not None
    t = torch.tensor([int(pairs), int(rows)], dtype=torch.int64, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t[0]), int(t[1])


class TableGather:
    """Gather per-pair inlier tables from every rank to rank 0.

    `run()` sends the workspace's table arena (see TableLayout) with one fixed-size gather and
    no host synchronisation when every rank allocated the same arena shape
    (pipeline.FrontEnd.prepare(table_pairs=, table_rows=) from agree_table_shape); otherwise
    it falls back to `gather()`, which negotiates sizes and sends each table on its own.
    Works on whatever device the tensors live on (gloo ranks sharing one GPU stage through
    host memory: a rehearsal path).
    """

    def __init__(self, ws=None, world=1, rank=0, group=None, pose="best_Rt", collective=None):
        """pose: which per-pair pose of the workspace travels -- "best_Rt" (winner of a given
        hypothesis set) or "pose" (result of the device-side RANSAC).
        collective: go through the process group even with a single rank (default: only when
        world > 1); lets the collective path be exercised on a one-GPU box."""
        self.ws, self.world, self.rank, self.group, self.pose = ws, world, rank, group, pose
        self.collective = world > 1 if collective is None else bool(collective)
        self.result = None
        self.fixed = False
        if ws is not None and self.collective:
            gloo = dist.get_backend(group) == "gloo"
            dev = "cpu" if gloo else ws.arena.device
            mine = torch.tensor([ws.P, ws.layout.pairs, ws.layout.rows, getattr(ws, "max_nL", 0)], dtype=torch.int64, device=dev)
            every = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(every, mine, group=group)
            every = torch.stack(every).cpu().numpy()
            self.pairs_of = [int(v) for v in every[:, 0]]
            self.kp_cap = int(every[:, 3].max())
            self.fixed = bool((every[:, 1] == every[0, 1]).all() and (every[:, 2] == every[0, 2]).all())
            if self.fixed and rank == 0:
                self.recv = [torch.empty(ws.layout.nbytes, dtype=torch.uint8, device=dev) for _ in range(world)]

    def _shard(self, views, P):
        pose = views["pose_b" if self.pose == "pose" else "pose_a"]
        kp_cap = getattr(self, "kp_cap", None) or getattr(self.ws, "max_nL", None) or None
        return dict(n_inl=views["n_inl"][:P], n_cons=views["n_cons"][:P], pose=pose[:P], rows=views["rows"],
                    loc=views["loc"], kp_cap=kp_cap)

    def run(self):
        """Tables of every rank on rank 0 (None elsewhere): list of dicts n_inl [P_r], n_cons [P_r],
        pose [P_r,12], rows [>=T_r,4], loc [>=T_r,6]; the first sum(n_inl) rows are valid."""
        ws = self.ws
        P = ws.P
        ws.table_views["n_cons"][:P].copy_(ws.cons.count[:P])
        if not self.collective:
            self.result = [self._shard(ws.table_views, P)]
            return self.result
        if not self.fixed:
            v = ws.table_views
            pose = v["pose_b" if self.pose == "pose" else "pose_a"]
            self.result = self.gather(v["n_inl"][:P], v["rows"], v["loc"], pose[:P], v["n_cons"][:P])
            return self.result
        src = ws.arena
        if src.is_cuda and dist.get_backend(self.group) == "gloo":
            src = src.cpu()
        dist.gather(src, self.recv if self.rank == 0 else None, dst=0, group=self.group)
        if self.rank != 0:
            self.result = None
            return None
        self.result = [self._shard(ws.layout.views(self.recv[r]), self.pairs_of[r]) for r in range(self.world)]
        return self.result

    def gather(self, n_inl, rows, loc, pose, n_cons):
        """Variable-size gather of explicit tables (sizes are exchanged first; one host sync)."""
        world, rank = self.world, self.rank
        if self.collective and n_inl.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal on a box without one GPU per rank: stage through host memory
            n_inl, rows, loc, pose, n_cons = (t.cpu() for t in (n_inl, rows, loc, pose, n_cons))
        dev = n_inl.device
        if not self.collective:
            T = int(n_inl.sum().item())
            return [dict(n_inl=n_inl, rows=rows[:T], loc=loc[:T], pose=pose, n_cons=n_cons)]
        # 1. sizes: pairs per rank and total rows per rank (one small all_gather + host sync)
        mine = torch.stack([torch.tensor(n_inl.shape[0], device=dev, dtype=torch.int64), n_inl.sum().to(torch.int64)])
        sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(sizes, mine, group=self.group)
        sizes = torch.stack(sizes).cpu().numpy()
        maxP, maxT = int(sizes[:, 0].max()), int(sizes[:, 1].max())

        def padded(t, n):
            if t.shape[0] >= n:
                return t[:n].contiguous()
            pad = torch.zeros((n - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            return torch.cat([t, pad])

        # 2. payload: padded to the largest shard so that one gather per table suffices
        out = []
        for t, n in ((n_inl, maxP), (n_cons, maxP), (pose, maxP), (rows, maxT), (loc, maxT)):
            src = padded(t, n)
            dst = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
            dist.gather(src, dst, dst=0, group=self.group)
            out.append(dst)
        if rank != 0:
            return None
        res = []
        for r in range(world):
            Pr, Tr = int(sizes[r, 0]), int(sizes[r, 1])
            res.append(dict(n_inl=out[0][r][:Pr], n_cons=out[1][r][:Pr], pose=out[2][r][:Pr], rows=out[3][r][:Tr],
                            loc=out[4][r][:Tr]))
        return res


def _chain_poses_numpy(rel_poses, first=None):
    n = rel_poses.shape[0] + 1
    ex = np.zeros((n, 3, 4))
    ex[0] = np.hstack([np.eye(3), np.zeros((3, 1))]) if first is None else first
    cam = np.zeros((n, 3))
    for i in range(n - 1):
        Rt = rel_poses[i]
        R, t = Rt[:, :-1], Rt[:, -1]
        new_R = R @ ex[i][:, :-1]
        new_t = R @ ex[i][:, -1] + t
        ex[i + 1] = np.hstack((new_R, new_t[:, None]))
        cam[i + 1] = -new_R.T @ new_t
    return ex, cam


def _chain_poses_native(rel_poses, first=None):
    from .track_database import _native
    P = rel_poses.shape[0]
    ex, cam = np.empty((P + 1, 3, 4)), np.empty((P + 1, 3))
    start = np.hstack([np.eye(3), np.zeros((3, 1))]) if first is None else np.asarray(first, dtype=np.float64)
    _native.chain_poses(np.ascontiguousarray(rel_poses, dtype=np.float64).reshape(-1, 12),
                        np.ascontiguousarray(start.reshape(12)), ex, cam)
    return ex, cam


_native_chain_ok = None


def native_chain_matches_numpy():
    """Whether the C helper's pose chain is bit-identical to numpy's `@` chain on this host.  numpy's
    small products go through its BLAS, whose summation order (and use of fused multiply-add) is a
    property of that build and CPU, so the helper -- written to the order numpy 2.2 / OpenBLAS takes
    on x86-64 with FMA -- is trusted only after reproducing numpy on a seeded sample of 64 poses
    (every one of the 960 sums would have to round alike by chance otherwise).  Checked once."""
    global _native_chain_ok
    if _native_chain_ok is None:
        from .track_database import _native
        if _native is None:
            _native_chain_ok = False
        else:
            from .synthetic import rodrigues_batch
            rng = np.random.default_rng(20240607)
            rel = np.empty((64, 3, 4))
            rel[:, :, :3] = rodrigues_batch(rng.normal(size=(64, 3)) * 0.2)
            rel[:, :, 3] = rng.normal(size=(64, 3))
            first = np.hstack([rodrigues_batch(rng.normal(size=(1, 3)))[0], rng.normal(size=(3, 1))])
            a, b = _chain_poses_numpy(rel, first), _chain_poses_native(rel, first)
            _native_chain_ok = bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))
    return _native_chain_ok


def chain_poses(rel_poses, first=None):
    """Global extrinsics and camera centres from relative poses, as the reference
    composes them (utils/utils.py:283-288): R' = R @ R_prev, t' = R @ t_prev + t,
    centre = -R'^T t'.  rel_poses: [P,3,4]; first: extrinsic of frame 0 (the reference
    starts at M1, utils/utils.py:244-245; default [I|0]).  Returns ([P+1,3,4], [P+1,3]).
    The loop runs in the C helper when that reproduces numpy's products bit for bit on this
    host (native_chain_matches_numpy), else in numpy as the reference writes it: every path
    through this module -- host replay, device track arrays, the batched VO drivers -- gets
    the same poses either way."""
    rel_poses = np.asarray(rel_poses, dtype=np.float64).reshape(-1, 3, 4)
    if rel_poses.shape[0] > 8 and native_chain_matches_numpy():
        return _chain_poses_native(rel_poses, first)
    return _chain_poses_numpy(rel_poses, first)


def replay_tracks(shards, first_frame=0, first=None):
    """Sequential tail on rank 0: rebuild the track database from gathered shard tables
    (in rank order = frame order) exactly as the frame loop would have fed it
    (utils/utils.py:277-292).  first: extrinsic matrix of the first frame of the shards (the
    reference starts the chain at M1, utils/utils.py:244-245; default [I|0]); ex_matrices and
    camera_positions are indexed from that frame (row 0 = first_frame)."""
    db = TrackDatabase()
    poses = []
    frame = first_frame
    # Millions of small tuples are created and none of them can be part of a reference cycle;
    # the cyclic collector would rescan the growing store again and again (2-5x the run time).
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        frame = _replay_into(db, poses, shards, frame)
    finally:
        if gc_was_on:
            gc.enable()
    if poses:
        ex, cam = chain_poses(np.stack(poses), first=first)
        db.set_ex_matrices(ex)
        db.set_ex_camera_positions(cam)
    return db


def _fetch(tensors):
    """Device tensors -> numpy arrays through ONE pinned host buffer (asynchronous copies at the
    link's rate, one synchronisation); the arrays are views of that buffer and keep it alive.
    Pinned blocks come from torch's caching host allocator, so a buffer released by an earlier
    result is reused instead of being pinned again."""
    sizes = [t.numel() * t.element_size() for t in tensors]
    offs = np.concatenate([[0], np.cumsum([(n + 255) // 256 * 256 for n in sizes])]).astype(np.int64)
    buf = torch.empty(int(offs[-1]) or 1, dtype=torch.uint8, pin_memory=True)
    views = []
    for t, n, o in zip(tensors, sizes, offs[:-1].tolist()):
        v = buf[o:o + n].view(t.dtype).reshape(t.shape)
        v.copy_(t, non_blocking=True)
        views.append(v)
    torch.cuda.current_stream().synchronize()
    return [v.numpy() for v in views]


def replay_tracks_device(shards, first_frame=0, first=None, device=None):
    """replay_tracks with the sequential part done on the GPU: track ids, track lengths and the
    track-point table come from slam_track_assign over the gathered inlier tables (one pass of
    pointer jumping and two scans instead of a walk over every row), and the result is an
    ArrayTrackDatabase that answers the track store's getters from those arrays.  Same state as
    replay_tracks; falls back to it when the tables hold something the kernels do not cover (two rows
    of one pair continuing the same track).  Tables on the host (gloo rehearsal) are uploaded."""
    from . import ops
    from .track_database import ArrayTrackDatabase
    shards = [sh for sh in shards]
    if device is None:
        cuda = [sh["rows"].device for sh in shards if sh["rows"].is_cuda]
        device = cuda[0] if cuda else torch.device("cuda", torch.cuda.current_device())
    up = lambda t: t if t.device == device else t.to(device)
    # only the valid rows of every shard (the tables carry unused capacity behind them, whose contents
    # are stale or uninitialised), back to back: one size read per shard
    totals = [int(sh["n_inl"].sum().item()) for sh in shards]
    if len(shards) == 1:
        t = totals[0]
        n_inl_d, rows_d, loc_d = up(shards[0]["n_inl"]), up(shards[0]["rows"][:t]), up(shards[0]["loc"][:t])
    else:
        n_inl_d = torch.cat([up(sh["n_inl"]) for sh in shards])
        rows_d = torch.cat([up(sh["rows"][:t]) for sh, t in zip(shards, totals)])
        loc_d = torch.cat([up(sh["loc"][:t]) for sh, t in zip(shards, totals)])
    n_inl_d = n_inl_d.to(torch.int32).contiguous()
    rows_d, loc_d = rows_d.contiguous(), loc_d.contiguous()
    P = int(n_inl_d.shape[0])
    kp_caps = [sh.get("kp_cap") for sh in shards]
    if P == 0 or rows_d.shape[0] == 0:
        return replay_tracks(shards, first_frame, first)
    if any(k is None for k in kp_caps):
        kp_cap = int(torch.maximum(rows_d[:, 0].max(), rows_d[:, 2].max()).item()) + 1
    else:
        kp_cap = max(int(k) for k in kp_caps)
    try:
        out = ops.track_assign(rows_d, loc_d, n_inl_d, kp_cap, first_frame)
    except ops._lib.SlamError:
        return replay_tracks(shards, first_frame, first)      # e.g. keypoint indices beyond what the kernels index
    summary = out["summary"].cpu().numpy()
    T, n_tracks, N, longest, flags = (int(v) for v in summary[:5])
    if flags:
        return replay_tracks(shards, first_frame, first)
    row_track, row_depth, track_start, pt_kp, pt_frame, pt_loc = _fetch(
        [out["row_track"][:T], out["row_depth"][:T], out["track_start"][:n_tracks + 1], out["pt_kp"][:N],
         out["pt_frame"][:N], out["pt_loc"][:N]])
    db = ArrayTrackDatabase(first_frame, n_inl_d.cpu().numpy(), row_track, row_depth, track_start, pt_kp, pt_frame, pt_loc,
                            max_length=longest)
    n_inl = np.concatenate([sh["n_inl"].cpu().numpy() for sh in shards]).astype(np.int64)
    n_cons = np.concatenate([sh["n_cons"].cpu().numpy() for sh in shards]).astype(np.int64)
    db.set_frame_records(np.where(n_cons > 0, n_inl / np.maximum(n_cons, 1), 0.0), n_inl)
    poses = np.concatenate([sh["pose"].cpu().numpy().reshape(-1, 12) for sh in shards])
    ex, cam = chain_poses(poses.reshape(-1, 3, 4), first=first)
    db.set_ex_matrices(ex)
    db.set_ex_camera_positions(cam)
    return db


def _replay_into(db, poses, shards, frame):
    for sh in shards:
        n_inl = sh["n_inl"].cpu().numpy()
        n_cons = sh["n_cons"].cpu().numpy()
        T = int(n_inl.sum())                       # the tables may carry unused capacity behind the valid rows
        rows = sh["rows"][:T].cpu().numpy()
        loc = sh["loc"][:T].cpu().numpy()
        pose = sh["pose"].cpu().numpy().reshape(-1, 3, 4)
        db.update_tracks_from_pair_tables(frame, rows, loc, n_inl)
        for p in range(n_inl.shape[0]):
            m = int(n_inl[p])
            db.add_inliers_ratio(frame, (m / int(n_cons[p])) if n_cons[p] else 0.0)
            db.add_num_matches(frame, num_matches=np.float64(m))   # np.sum of find_supporters' float64 0/1 array (utils/utils.py:279,282,390)
            poses.append(pose[p])
            frame += 1
    return frame
