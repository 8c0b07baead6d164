"""Visual odometry over a frame range with the batched GPU front-end.

Produces what the reference's frame loop produces (utils/utils.py:236-294) -- camera
positions and a filled track database -- but runs stereo matching, triangulation,
temporal matching and the four-image join for the WHOLE range in four kernel launches
(slam_amd.pipeline), and only then walks the pairs in order for PnP-RANSAC, whose use of
the process-global random stream makes it sequential by definition.  Given the same
features and the same `random` state it returns exactly what
slam_amd.frontend.track_camera_for_many_images returns.
"""
import numpy as np
import torch

from . import frontend, pipeline
from .constants import PIXEL_THRESH, STEREO_ROW_THRESH, VO_MAX_ITERATIONS
from .sharding import TableGather, chain_poses, replay_tracks, replay_tracks_device
from .track_database import TrackDatabase


def front_end_tables(frames, cameras=None, thresh=1, device=None):
    """Stages 1-4 for all frames on the GPU; returns (workspace, list of per-pair host
    dicts with rows [M,4], X [M,3], uvL [M,2], uvR [M,2])."""
    k, m1, m2 = cameras if cameras is not None else frontend.read_cameras()
    if device is None:
        device = frontend._default_device()
    fe = pipeline.FrontEnd(k, m1, m2, row_thresh=STEREO_ROW_THRESH, ratio_thresh=thresh, pix_thresh=PIXEL_THRESH)
    tb = pipeline.FrameTables.from_frames(frames, device)
    ws = fe.prepare(tb, n_hyp=0)
    fe.run(ws)
    torch.cuda.synchronize()
    return ws, pipeline.consensus_to_host(ws)


def track_camera_batched(frames, cameras=None, thresh=1, max_iterations=VO_MAX_ITERATIONS, device=None):
    """(camera_positions [F,3], TrackDatabase) for precomputed features `frames`
    (objects with kpL/kpR [N,2] float32 and descL/descR [N,D] uint8)."""
    k, m1, m2 = cameras if cameras is not None else frontend.read_cameras()
    ws, pairs = front_end_tables(frames, (k, m1, m2), thresh=thresh, device=device)
    db = TrackDatabase()
    rel = []
    for p, tab in enumerate(pairs):
        Rt, ratio, mask = frontend.ransac_on_arrays(tab["X"], tab["uvL"], tab["uvR"], k, m2, thresh=PIXEL_THRESH,
                                                    max_iterations=max_iterations)
        assert mask is not None, f"RANSAC Couldn't find a transformation from frame {p} to frame {p + 1}..."
        sel = np.flatnonzero(mask)
        rows = tab["rows"][sel]
        f0, f1 = frames[p], frames[p + 1]
        loc = np.stack([f0.kpL[rows[:, 0], 0], f0.kpR[rows[:, 1], 0], f0.kpL[rows[:, 0], 1],
                        f1.kpL[rows[:, 2], 0], f1.kpR[rows[:, 3], 0], f1.kpL[rows[:, 2], 1]], axis=1) \
            if len(sel) else np.zeros((0, 6), np.float32)
        db.update_tracks_from_pair_tables(p, rows, loc, [len(rows)])
        db.add_inliers_ratio(p, ratio)
        db.add_num_matches(p, num_matches=np.sum(mask))
        rel.append(Rt)
    n = len(frames)
    ex, cam = chain_poses(np.stack(rel), first=m1) if rel else (np.zeros((n, 3, 4)), np.zeros((n, 3)))
    if rel:
        db.set_ex_camera_positions(cam)
        db.set_ex_matrices(ex)
    return cam, db


def track_camera_device(frames, seed=6, n_hyp=VO_MAX_ITERATIONS, cameras=None, thresh=1, device=None,
                        stereo_refine=False):
    """(camera_positions [F,3], TrackDatabase) with the whole RANSAC on the device as well:
    per pair `n_hyp` minimal samples from a generator keyed by (seed, pair, hypothesis), P3P,
    supporter counts, the reference's stopping rule, refinement of the winner, inlier tables
    (pipeline.FrontEnd.run_ransac).  The track store is an ArrayTrackDatabase built from device-made track ids.  The
    draws differ from the reference's process-global `random` stream, which chains the frames
    one after the other; everything downstream of the draws follows utils/utils.py:413-464.
    stereo_refine=True departs from the reference on purpose: the winner is refined on the left AND
    right residuals (slam_refine_winners_stereo), which removes the heading bias the left-only cost
    of utils/utils.py:452 leaves on triangulated points."""
    k, m1, m2 = cameras if cameras is not None else frontend.read_cameras()
    if device is None:
        device = frontend._default_device()
    fe = pipeline.FrontEnd(k, m1, m2, row_thresh=STEREO_ROW_THRESH, ratio_thresh=thresh, pix_thresh=PIXEL_THRESH)
    tb = pipeline.FrameTables.from_frames(frames, device)
    ws = fe.prepare(tb, n_hyp=n_hyp)
    fe.run_ransac(ws, seed, stereo_refine=stereo_refine)
    torch.cuda.synchronize()
    found = ws.found.cpu().numpy()[:ws.P]
    assert found.all(), f"RANSAC Couldn't find a transformation from frame {int(np.argmin(found))} to the next one"
    # track ids and the track-point table are made on the device too; the store answers from arrays
    db = replay_tracks_device(TableGather(ws, pose="pose").run(), first=m1)
    return np.asarray(db.camera_positions), db


def loop_closure_pairs(n_frames, keyframe_stride=20, min_gap=30):
    """[P,2] (reference keyframe, candidate keyframe) frame indices of an all-against-earlier
    sweep: keyframes every `keyframe_stride` frames (models/BundleAdjustment.py:205-210), each one
    against every earlier keyframe at least `min_gap` frames back (models/LoopClosure.py:214,234)."""
    kf = np.arange(0, n_frames, keyframe_stride)
    out = [(int(a), int(b)) for a in kf for b in kf if a - b >= min_gap]
    return np.asarray(out, dtype=np.int64).reshape(-1, 2)


def loop_closure_batched(frames, pairs, cameras=None, thresh=0.6, n_hyp=None, seed=6, device=None, stereo_refine=False):
    """Consensus matching of many keyframe pairs at once: what
    consensus_matching_of_general_two_frames (models/LoopClosure.py:112-139) does for one pair --
    stereo matching and triangulation of both keyframes, left-left 2-NN matching at ratio
    `thresh`, four-image join, PnP-RANSAC with 500 iterations -- as one launch per stage over
    all pairs, RANSAC included (pipeline.FrontEnd.run_ransac).

    frames: the keyframes' features, indexed by the entries of `pairs` [P,2] = (reference,
    candidate).  Returns a dict of host arrays per pair: valid (at least LC_MIN_CONSENSUS
    consensus matches and a pose was found: the reference returns six Nones otherwise), pose
    [P,3,4], inliers_ratio, n_consensus, n_filtered_matches (left-left matches that passed the
    ratio test: the caller's >= 30 gate, models/LoopClosure.py:264-266), n_inliers, and the
    workspace with the inlier tables (rows / (xl,xr,yl) locations of the supporters)."""
    from .constants import LC_MAX_ITERATIONS, LC_MIN_CONSENSUS
    k, m1, m2 = cameras if cameras is not None else frontend.read_cameras()
    if device is None:
        device = frontend._default_device()
    n_hyp = LC_MAX_ITERATIONS if n_hyp is None else int(n_hyp)
    fe = pipeline.FrontEnd(k, m1, m2, row_thresh=STEREO_ROW_THRESH, ratio_thresh=thresh, pix_thresh=PIXEL_THRESH)
    tb = pipeline.FrameTables.from_frames(frames, device)
    ws = fe.prepare(tb, n_hyp=n_hyp, pairs=pairs)
    fe.run_ransac(ws, seed, stereo_refine=stereo_refine)     # (option: left + right residuals, see track_camera_device)
    P = ws.P
    # kept left-left matches per pair: segment sums of the ratio-test mask
    csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=ws.t_keep.device), ws.t_keep.to(torch.int64).cumsum(0)])
    lo = torch.from_numpy(ws.t_off).to(csum.device)
    hi = lo + torch.from_numpy(np.diff(tb.offL)[ws.pairs_ab[:, 0]]).to(csum.device)
    n_filtered = (csum[hi] - csum[lo]).cpu().numpy() if P else np.zeros(0, np.int64)
    torch.cuda.synchronize()
    n_cons = ws.cons.count.cpu().numpy()[:P]
    found = ws.found.cpu().numpy()[:P].astype(bool)
    return dict(valid=found & (n_cons >= LC_MIN_CONSENSUS), pose=ws.pose.cpu().numpy()[:P].reshape(P, 3, 4),
                inliers_ratio=ws.ratio.cpu().numpy()[:P], n_consensus=n_cons, n_filtered_matches=n_filtered,
                n_inliers=ws.n_inl.cpu().numpy()[:P], workspace=ws)
