"""The collective path of the table gather over RCCL (backend "nccl" on ROCm) with a single rank: the
only form of it a one-GPU box can run.  The gathered shard must equal the local tables."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from slam_amd import pipeline, sharding
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_table_gather_over_rccl_with_one_rank(device):
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=device)
    try:
        k, m1, m2 = read_cameras()
        seq = SyntheticSequence(300, 4, seed=31)
        H = 40
        fe = pipeline.FrontEnd(k, m1, m2)
        tb = pipeline.FrameTables.from_frames(seq.frames, device)
        P = len(seq.frames) - 1
        tp, tr = sharding.agree_table_shape(P, P * 300, device=device)
        ws = fe.prepare(tb, n_hyp=H, table_pairs=tp, table_rows=tr)
        Rt = np.stack([seeded_hypotheses(seq.relative_pose(p, p + 1), H, seed=p) for p in range(P)])
        fe.run(ws, torch.from_numpy(Rt).to(device))
        g = sharding.TableGather(ws, 1, 0, collective=True)
        assert g.fixed                                   # one arena shape: the fixed-size gather
        shard = g.run()[0]
        torch.cuda.synchronize()
        T = int(ws.n_inl.sum())
        assert T > 100
        assert torch.equal(shard["n_inl"], ws.n_inl[:P]) and torch.equal(shard["n_cons"], ws.cons.count[:P])
        assert torch.equal(shard["rows"][:T], ws.inl_rows[:T]) and torch.equal(shard["loc"][:T], ws.inl_loc[:T])
        assert torch.equal(shard["pose"], ws.best_Rt[:P].reshape(P, -1))
        # the variable-size form (ranks with unequal arenas fall back to it)
        v = ws.table_views
        res = g.gather(v["n_inl"][:P], v["rows"], v["loc"], v["pose_a"][:P], v["n_cons"][:P])[0]
        assert torch.equal(res["rows"], ws.inl_rows[:T]) and torch.equal(res["loc"], ws.inl_loc[:T])
        db = sharding.replay_tracks([{kk: vv for kk, vv in shard.items()}])
        assert db.get_num_frames() == len(seq.frames)
    finally:
        dist.destroy_process_group()
