#!/usr/bin/env python3
"""Per-stage device time of the batched RANSAC (pipeline.FrontEnd.ransac) on the benchmark batch."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, pipeline                                        # noqa: E402
from slam_amd.pipeline import _p                                             # noqa: E402
from slam_amd.synthetic import SyntheticSequence, read_cameras               # noqa: E402

P, N = int(os.environ.get("AB_PAIRS", "128")), 2000
dev = torch.device("cuda:0")
seq = SyntheticSequence(N, P + 1)
k, m1, m2 = read_cameras()
fe = pipeline.FrontEnd(k, m1, m2)
tb = pipeline.FrameTables.from_frames(seq.frames, dev)
for H in (1000, 10000):
    ws = fe.prepare(tb, n_hyp=H)
    fe.prepare_ransac(ws)
    fe.run_ransac(ws, 6)
    torch.cuda.synchronize()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    c, lib = ws.cons, fe.lib
    stages = {
        "sample": lambda: lib.slam_ransac_sample(ctypes.c_uint64(6), ws.P, ws.H, _p(c.count), 0, _p(ws.samples), stream),
        "p3p": lambda: lib.slam_p3p_hypotheses(ws.hK, _p(c.X), _p(c.Y), _p(c.Z), _p(c.uL), _p(c.vL), ws.cap, _p(c.count), 0,
                                               _p(ws.samples), ws.P, ws.H, 0, ws.H, None, _p(ws.Rt_gen), stream),
        "score": lambda: fe.score(ws, ws.Rt_gen, stream),
        "select": lambda: lib.slam_ransac_select(_p(ws.counts), _p(ws.Rt_gen), ws.P, ws.H, ws.H, _p(c.count), 0, ws.cap, H,
                                                 _p(ws.best), _p(ws.best_count), _p(ws.best_Rt), _p(ws.iters),
                                                 _p(ws.stopped), stream),
        "first_slice_p3p": lambda: lib.slam_p3p_hypotheses(ws.hK, _p(c.X), _p(c.Y), _p(c.Z), _p(c.uL), _p(c.vL), ws.cap,
                                                           _p(c.count), 0, _p(ws.samples), ws.P, ws.H, 0, 128, None,
                                                           _p(ws.Rt_gen), stream),
        "first_slice_score": lambda: lib.slam_score_hypotheses_range(
            _p(ws.Rt_gen), ws.P, ws.H, 0, 128, None, ws.hK, ws.hT, _p(c.X), _p(c.Y), _p(c.Z), _p(c.uL), _p(c.vL), _p(c.uR),
            _p(c.vR), ws.cap, _p(c.count), 0, fe.pix_thresh, _p(ws.counts), None, 0, _p(ws.score_work), 0, stream),
        "refine_winners": lambda: lib.slam_refine_winners(ws.hK, ws.hT, fe.pix_thresh, _p(c.X), _p(c.Y), _p(c.Z), _p(c.uL),
                                                          _p(c.vL), _p(c.uR), _p(c.vR), ws.cap, _p(c.count), 0,
                                                          _p(ws.best_Rt), ws.P, 30, _p(ws.ref_Rt), _p(ws.ref_ok),
                                                          _p(ws.final_count), stream),
        "adopt": lambda: lib.slam_ransac_adopt(ws.P, _p(c.count), 0, ws.cap, _p(ws.best_count), _p(ws.final_count),
                                               _p(ws.best_Rt), _p(ws.ref_Rt), _p(ws.pose), _p(ws.ratio), _p(ws.found),
                                               stream),
        "winner_tables": lambda: lib.slam_winner_tables(
            _p(ws.pair), ws.P, _p(c.rows), _p(c.count), ws.cap, ws.hK, ws.hT, fe.pix_thresh, _p(ws.ref_Rt),
            _p(ws.final_count), _p(c.X), _p(c.Y), _p(c.Z), _p(c.uL), _p(c.vL), _p(c.uR), _p(c.vR), _p(ws.tb.xL), _p(ws.tb.yL),
            _p(ws.tb.xR), _p(ws.inl_rows), _p(ws.inl_loc), _p(ws.n_inl), stream),
        "whole_ransac_one_phase": lambda: fe.ransac(ws, 6, stream, first_slice=0),
        "whole_ransac_first_32_128": lambda: fe.ransac(ws, 6, stream, first_slice=(32, 128)),
        "whole_ransac_first_32": lambda: fe.ransac(ws, 6, stream, first_slice=32),
        "whole_ransac_first_16_64": lambda: fe.ransac(ws, 6, stream, first_slice=(16, 64)),
        "whole_ransac": lambda: fe.ransac(ws, 6, stream),
        "whole_front_end": lambda: fe.run_ransac(ws, 6),
    }
    out = {}
    for name, fn in stages.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / 20, 4)
    it = ws.iters.cpu().numpy()[:P]
    print(f"H={H} pairs={P}: ms per stage {out}; mean iterations the rule ran {it.mean():.1f}; "
          f"pairs/s whole front-end {P / (out['whole_front_end'] * 1e-3):.0f}")
