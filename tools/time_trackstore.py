#!/usr/bin/env python3
"""Where the time of sharding.replay_tracks_device goes on a full-length sequence.
    python tools/time_trackstore.py --frames 4541 --kp 2000"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import ops, pipeline, sharding                              # noqa: E402
from slam_amd.synthetic import SyntheticSequence, read_cameras             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4541)
    ap.add_argument("--kp", type=int, default=2000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    seq = SyntheticSequence(a.kp, a.frames)
    k, m1, m2 = read_cameras()
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seq.frames, dev)
    ws = fe.prepare(tb, n_hyp=1000)
    fe.run_ransac(ws, 6)
    torch.cuda.synchronize()
    shards = sharding.TableGather(ws, pose="pose").run()
    sh = shards[0]
    P = ws.P
    for rep in range(3):
        t = [time.perf_counter()]
        out = ops.track_assign(sh["rows"], sh["loc"], sh["n_inl"].to(torch.int32).contiguous(), ws.max_nL, 0)
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        summary = out["summary"].cpu().numpy()
        T, nt, N = int(summary[0]), int(summary[1]), int(summary[2])
        t.append(time.perf_counter())
        arrs = [out["row_track"][:T].cpu(), out["row_depth"][:T].cpu(), out["track_start"][:nt + 1].cpu(),
                out["pt_kp"][:N].cpu(), out["pt_frame"][:N].cpu(), out["pt_loc"][:N].cpu()]
        t.append(time.perf_counter())
        db = sharding.replay_tracks_device(shards, first=m1)
        t.append(time.perf_counter())
        ids = db.get_track_ids_for_frame(a.frames // 2)
        _ = [db.get_track_data(i) for i in ids]
        t.append(time.perf_counter())
        d = np.diff(t) * 1e3
        print(f"rep {rep}: kernels+alloc {d[0]:.2f} ms, summary {d[1]:.2f}, D2H of {sum(x.numel() * x.element_size() for x in arrs) / 1e6:.0f} MB "
              f"{d[2]:.2f}, whole replay_tracks_device {d[3]:.2f}, one window's queries {d[4]:.2f}")


if __name__ == "__main__":
    main()
