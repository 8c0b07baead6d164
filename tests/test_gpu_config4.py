"""GPU parity at BASELINE configs[4]'s own per-frame size: 8000 keypoints x 61-byte descriptors.

* a stretch of the stress sequence (64 consecutive frame pairs, one ninth of one GPU's share)
  through the whole front-end with the device RANSAC: size-independent properties on every row of
  every frame and pair, exact oracle parity on the first and last pair;
* the loop-closure all-vs-keyframe sweep at that size (12 keyframes, every keyframe against all
  keyframes at least two back = 55 pairs, ratio 0.6, 500-iteration RANSAC): properties on every
  pair, the oracle's per-pair pipeline on sampled pairs."""
import numpy as np
import pytest
import torch

from oracle import cpu as oracle
from oracle import numpy_twin as twin
from slam_amd import pipeline, vo
from slam_amd.synthetic import SyntheticSequence, read_cameras

from test_gpu_bench_sizes import _ham, _mask_bits, _oracle_pair, _oracle_stereo

pytestmark = pytest.mark.gpu

N_KP = 8000


@pytest.mark.timeout(600)
def test_8k_sequence_share_through_device_ransac(device):
    n_frames, H = 65, 1000
    k, m1, m2 = read_cameras()
    Pm, Qm = k @ m1, k @ m2
    seq = SyntheticSequence(N_KP, n_frames)
    P = n_frames - 1
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seq.frames, device)
    ws = fe.prepare(tb, n_hyp=H)
    fe.run_ransac(ws, seed=6)
    torch.cuda.synchronize()
    s_idx, s_dist, s_keep = ws.s_idx.cpu().numpy(), ws.s_dist.cpu().numpy(), ws.s_keep.cpu().numpy()
    t_idx, t_dist, t_keep = ws.t_idx.cpu().numpy(), ws.t_dist.cpu().numpy(), ws.t_keep.cpu().numpy()
    smap, valid, Z = ws.smap.cpu().numpy(), ws.valid.cpu().numpy(), ws.Z.cpu().numpy()
    n_cons, n_inl = ws.cons.count.cpu().numpy()[:P], ws.n_inl.cpu().numpy()[:P]
    found, iters = ws.found.cpu().numpy()[:P], ws.iters.cpu().numpy()[:P]
    pose = ws.pose.cpu().numpy()[:P].reshape(P, 3, 4)
    cons_rows, cap = ws.cons.rows.cpu().numpy(), ws.cap

    # ---- every row of every frame ------------------------------------------------------------
    for f, fr in enumerate(seq.frames):
        s = slice(f * N_KP, (f + 1) * N_KP)
        idx = s_idx[s, 0]
        assert idx.min() >= 0 and idx.max() < N_KP
        assert np.array_equal(_ham(fr.descL, fr.descR[idx]), s_dist[s, 0]), f
        dev = np.abs(fr.kpL[:, 1].astype(np.float64) - fr.kpR[idx, 1].astype(np.float64))
        assert np.array_equal(s_keep[s].astype(bool), dev <= 2.0), f
        assert np.array_equal(smap[s], np.where(s_keep[s] > 0, idx, -1)), f
        ok = valid[s] > 0
        assert not (ok & (s_keep[s] == 0)).any() and (Z[s][ok] > 0).all(), f
        true = fr.idL >= 0
        assert (fr.idR[idx[true]] == fr.idL[true]).mean() > 0.995, f
        if f < P:
            nx = seq.frames[f + 1]
            i2, d2 = t_idx[s], t_dist[s]
            assert np.array_equal(_ham(fr.descL, nx.descL[i2[:, 0]]), d2[:, 0]), f
            assert np.array_equal(_ham(fr.descL, nx.descL[i2[:, 1]]), d2[:, 1]), f
            assert ((d2[:, 0] < d2[:, 1]) | ((d2[:, 0] == d2[:, 1]) & (i2[:, 0] < i2[:, 1]))).all(), f
            assert np.array_equal(t_keep[s].astype(bool), d2[:, 0].astype(np.float64) < 1.0 * d2[:, 1]), f
            # every consensus row joins a kept temporal match to the stereo partners of its two ends
            r = cons_rows[f * cap:f * cap + n_cons[f]]
            assert (np.diff(r[:, 0]) > 0).all() and t_keep[s][r[:, 0]].all() and np.array_equal(i2[r[:, 0], 0], r[:, 2])
            assert np.array_equal(smap[s][r[:, 0]], r[:, 1]) and valid[s][r[:, 0]].all()
            assert np.array_equal(smap[(f + 1) * N_KP:(f + 2) * N_KP][r[:, 2]], r[:, 3])
    assert found.all() and (n_cons > 0.3 * 0.6 * N_KP).all() and (n_inl > 0.9 * n_cons).all() and (iters < H).all()
    gt = np.stack([seq.relative_pose(p, p + 1) for p in range(P)])
    assert np.abs(pose - gt).max() < 0.05

    # ---- the oracle's own pass over the first and last pair -------------------------------------
    counts, samples = ws.counts.cpu().numpy(), ws.samples.cpu().numpy()
    best, ref_Rt = ws.best.cpu().numpy(), ws.ref_Rt.cpu().numpy()
    inl_off = np.concatenate([[0], np.cumsum(n_inl)])
    inl_rows, inl_loc = ws.inl_rows.cpu().numpy(), ws.inl_loc.cpu().numpy()
    picks = [0, P - 1]
    assert np.array_equal(samples[picks], oracle.ransac_sample(6, P, H, n_cons)[picks])
    for p in picks:
        fa, fb = seq.frames[p], seq.frames[p + 1]
        sa, sb = _oracle_stereo(fa, Pm, Qm), _oracle_stereo(fb, Pm, Qm)
        s = slice(p * N_KP, (p + 1) * N_KP)
        assert np.array_equal(s_idx[s, 0], sa["idx"][:, 0]) and np.array_equal(s_dist[s, 0], sa["dist"][:, 0])
        assert np.array_equal(smap[s], sa["smap"]) and np.array_equal(valid[s], sa["valid"])
        tidx, tdist, tkeep, rows, pts = _oracle_pair(fa, fb, sa, sb, 1.0)
        assert np.array_equal(t_idx[s], tidx) and np.array_equal(t_dist[s], tdist) and np.array_equal(t_keep[s], tkeep)
        assert int(n_cons[p]) == len(rows) and np.array_equal(cons_rows[p * cap:p * cap + len(rows)], rows)
        first = 128
        with np.errstate(all="ignore"):
            rc = oracle.score_hypotheses(ws.Rt_gen[p, :first].cpu().numpy(), k, m2[:, 3], *pts, 2.0, threads=8)
        assert np.array_equal(counts[p, :first], rc)
        want_best, want_iters = twin.ransac_stop_replay(rc, len(rows), first)
        assert want_iters < first and (int(best[p]), int(iters[p])) == (want_best, want_iters)
        c, m = oracle.score_hypotheses(ref_Rt[p][None], k, m2[:, 3], *pts, 2.0, want_masks=True)
        sel = rows[_mask_bits(m[0], len(rows))]
        assert int(n_inl[p]) == int(c[0]) == len(sel)
        assert np.array_equal(inl_rows[inl_off[p]:inl_off[p + 1]], sel)
        loc = np.stack([fa.kpL[sel[:, 0], 0], fa.kpR[sel[:, 1], 0], fa.kpL[sel[:, 0], 1],
                        fb.kpL[sel[:, 2], 0], fb.kpR[sel[:, 3], 0], fb.kpL[sel[:, 2], 1]], axis=1)
        assert np.array_equal(inl_loc[inl_off[p]:inl_off[p + 1]], loc)


@pytest.mark.timeout(600)
def test_8k_loop_closure_sweep(device):
    k, m1, m2 = read_cameras()
    Pm, Qm = k @ m1, k @ m2
    n_kf, H = 12, 500
    seq = SyntheticSequence(N_KP, n_kf, seed=31, step=2.0)
    pairs = np.asarray([(i, j) for i in range(n_kf) for j in range(n_kf) if i - j >= 2], dtype=np.int64)
    res = vo.loop_closure_batched(seq.frames, pairs, thresh=0.6, n_hyp=H, seed=6)
    ws = res["workspace"]
    P = len(pairs)
    assert ws.P == P == 55
    cap = ws.cap
    cons_rows, counts = ws.cons.rows.cpu().numpy(), ws.counts.cpu().numpy()
    t_idx, t_dist, t_keep = ws.t_idx.cpu().numpy(), ws.t_dist.cpu().numpy(), ws.t_keep.cpu().numpy()
    ref_Rt = ws.ref_Rt.cpu().numpy()
    n_inl = ws.n_inl.cpu().numpy()[:P]
    inl_off = np.concatenate([[0], np.cumsum(n_inl)])
    inl_rows = ws.inl_rows.cpu().numpy()
    smap = ws.smap.cpu().numpy()
    # ---- every pair: distances, order, ratio test, join membership, the reference's gates ----------
    for p, (a, b) in enumerate(pairs):
        fa, fb = seq.frames[a], seq.frames[b]
        s = slice(int(ws.t_off[p]), int(ws.t_off[p]) + N_KP)
        i2, d2 = t_idx[s], t_dist[s]
        assert np.array_equal(_ham(fa.descL, fb.descL[i2[:, 0]]), d2[:, 0]) and np.array_equal(_ham(fa.descL, fb.descL[i2[:, 1]]), d2[:, 1])
        assert ((d2[:, 0] < d2[:, 1]) | ((d2[:, 0] == d2[:, 1]) & (i2[:, 0] < i2[:, 1]))).all()
        assert np.array_equal(t_keep[s].astype(bool), d2[:, 0].astype(np.float64) < 0.6 * d2[:, 1])
        assert int(res["n_filtered_matches"][p]) == int(t_keep[s].sum())
        n = int(res["n_consensus"][p])
        r = cons_rows[p * cap:p * cap + n]
        assert t_keep[s][r[:, 0]].all() and np.array_equal(i2[r[:, 0], 0], r[:, 2])
        assert np.array_equal(smap[a * N_KP:(a + 1) * N_KP][r[:, 0]], r[:, 1])
        assert np.array_equal(smap[b * N_KP:(b + 1) * N_KP][r[:, 2]], r[:, 3])
        assert not res["valid"][p] or n >= 10                      # the reference gives up below 10 consensus matches
        if res["valid"][p]:
            assert res["inliers_ratio"][p] == int(n_inl[p]) / n
            if n_inl[p] >= 30:
                assert np.abs(res["pose"][p][:, :3] - seq.relative_pose(a, b)[:, :3]).max() < 2e-2, (a, b)
    assert int(res["valid"].sum()) >= 15
    # ---- sampled pairs: the oracle's whole per-pair pipeline -------------------------------------------
    sample = [0, 1, 9, 20, 37, P - 1]
    frames_needed = sorted({int(f) for p in sample for f in pairs[p]})
    st = {f: _oracle_stereo(seq.frames[f], Pm, Qm) for f in frames_needed}
    for p in sample:
        a, b = (int(v) for v in pairs[p])
        fa, fb = seq.frames[a], seq.frames[b]
        tidx, tdist, tkeep, rows, pts = _oracle_pair(fa, fb, st[a], st[b], 0.6)
        s = slice(int(ws.t_off[p]), int(ws.t_off[p]) + N_KP)
        assert np.array_equal(t_idx[s], tidx) and np.array_equal(t_dist[s], tdist) and np.array_equal(t_keep[s], tkeep)
        assert int(res["n_consensus"][p]) == len(rows) and np.array_equal(cons_rows[p * cap:p * cap + len(rows)], rows)
        if len(rows) < 4:
            assert not res["valid"][p]
            continue
        first = min(128, H)
        with np.errstate(all="ignore"):
            rc = oracle.score_hypotheses(ws.Rt_gen[p, :first].cpu().numpy(), k, m2[:, 3], *pts, 2.0, threads=8)
        assert np.array_equal(counts[p, :first], rc), (a, b)
        c, m = oracle.score_hypotheses(ref_Rt[p][None], k, m2[:, 3], *pts, 2.0, want_masks=True)
        sel = rows[_mask_bits(m[0], len(rows))]
        assert int(n_inl[p]) == len(sel) and np.array_equal(inl_rows[inl_off[p]:inl_off[p + 1]], sel), (a, b)
