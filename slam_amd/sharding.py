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


class TableGather:
    """Gather per-pair inlier tables from every rank to rank 0.

    Per rank: n_inl [P_r] int32 (rows per pair), rows [T_r,4] int32 and loc [T_r,6]
    float32 dense in pair order (T_r = sum n_inl), pose [P_r,12] float64, ratio inputs
    n_cons [P_r] int32.  Works on whatever device the tensors live on.
    """

    def __init__(self, ws=None, world=1, rank=0, group=None, pose="best_Rt"):
        """pose: which per-pair pose of the workspace travels -- "best_Rt" (winner of a given
        hypothesis set) or "pose" (result of the device-side RANSAC)."""
        self.ws, self.world, self.rank, self.group, self.pose = ws, world, rank, group, pose
        self.result = None

    def run(self):
        ws = self.ws
        P = ws.P
        self.result = self.gather(ws.n_inl[:P], ws.inl_rows, ws.inl_loc, getattr(ws, self.pose)[:P], ws.cons.count[:P])
        return self.result

    def gather(self, n_inl, rows, loc, pose, n_cons):
        world, rank = self.world, self.rank
        if world > 1 and n_inl.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal on a box without one GPU per rank: stage through host memory
            n_inl, rows, loc, pose, n_cons = (t.cpu() for t in (n_inl, rows, loc, pose, n_cons))
        dev = n_inl.device
        if world == 1:
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


def chain_poses(rel_poses, first=None):
    """Global extrinsics and camera centres from relative poses, as the reference
    composes them (utils/utils.py:283-288): R' = R @ R_prev, t' = R @ t_prev + t,
    centre = -R'^T t'.  rel_poses: [P,3,4]; first: extrinsic of frame 0 (the reference
    starts at M1, utils/utils.py:244-245; default [I|0]).  Returns ([P+1,3,4], [P+1,3])."""
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


def replay_tracks(shards, first_frame=0):
    """Sequential tail on rank 0: rebuild the track database from gathered shard tables
    (in rank order = frame order) exactly as the frame loop would have fed it
    (utils/utils.py:277-292)."""
    db = TrackDatabase()
    poses = []
    frame = first_frame
    for sh in shards:
        n_inl = sh["n_inl"].cpu().numpy()
        n_cons = sh["n_cons"].cpu().numpy()
        rows = sh["rows"].cpu().numpy()
        loc = sh["loc"].cpu().numpy()
        pose = sh["pose"].cpu().numpy().reshape(-1, 3, 4)
        off = 0
        for p in range(n_inl.shape[0]):
            m = int(n_inl[p])
            db.update_tracks_from_tables(frame, rows[off:off + m], loc[off:off + m])
            db.add_inliers_ratio(frame, (m / int(n_cons[p])) if n_cons[p] else 0.0)
            db.add_num_matches(frame, num_matches=float(m))
            poses.append(pose[p])
            off += m
            frame += 1
    if poses:
        ex, cam = chain_poses(np.stack(poses))
        db.set_ex_matrices(ex)
        db.set_ex_camera_positions(cam)
    return db
