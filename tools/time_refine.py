#!/usr/bin/env python3
"""Time of the refinement kernel against its iteration cap (how much one LM iteration costs)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, pipeline                               # noqa: E402
from slam_amd.pipeline import _p                                  # noqa: E402
from slam_amd.synthetic import SyntheticSequence, read_cameras    # noqa: E402


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda:0")
    seq = SyntheticSequence(2000, pairs + 1)
    k, m1, m2 = read_cameras()
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seq.frames, dev)
    ws = fe.prepare(tb, n_hyp=1000)
    fe.prepare_ransac(ws)
    fe.run_ransac(ws, seed=6)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    fe._score_one(ws, ws.best_Rt, stream)          # supporter mask of the winner, as ransac() has it
    torch.cuda.synchronize()
    c = ws.cons
    for its in (0, 1, 2, 3, 4, 6, 8, 30):
        def go():
            _lib.check(fe.lib.slam_refine_poses(ws.hK, _p(c.X), _p(c.Y), _p(c.Z), _p(c.uL), _p(c.vL), ws.cap,
                                                _p(c.count), 0, _p(ws.final_mask), max(ws.mask_words, 1),
                                                _p(ws.best_Rt), ws.P, its, _p(ws.ref_Rt), _p(ws.ref_ok), stream),
                       "slam_refine_poses")
        for _ in range(3):
            go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            go()
        e1.record()
        torch.cuda.synchronize()
        print(f"pairs={pairs} iterations<={its}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
    # the fused tail of the loop (supporters, refinement, supporters of the result): left-image cost, and with
    # the right image's residuals added
    for name, entry in (("slam_refine_winners", fe.lib.slam_refine_winners),
                        ("slam_refine_winners_stereo", fe.lib.slam_refine_winners_stereo)):
        def go():
            _lib.check(entry(ws.hK, ws.hT, fe.pix_thresh, _p(c.X), _p(c.Y), _p(c.Z), _p(c.uL), _p(c.vL), _p(c.uR), _p(c.vR),
                             ws.cap, _p(c.count), 0, _p(ws.best_Rt), ws.P, 30, _p(ws.ref_Rt), _p(ws.ref_ok),
                             _p(ws.final_count), stream), name)
        for _ in range(3):
            go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            go()
        e1.record()
        torch.cuda.synchronize()
        print(f"pairs={pairs} {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)


if __name__ == "__main__":
    main()
