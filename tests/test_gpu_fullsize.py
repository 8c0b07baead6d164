"""GPU tests at the benchmark configurations' full per-frame sizes (2 000 and 8 000
keypoints of 61 bytes), checked through size-independent properties on every row plus
exact oracle parity on sampled rows (a full oracle pass would take minutes on the CPU)."""
import numpy as np
import pytest
import torch

from oracle import cpu as oracle
from slam_amd import pipeline
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses

pytestmark = pytest.mark.gpu

_POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.int32)


def _ham(a, b):
    return _POP8[np.bitwise_xor(a, b)].sum(axis=1)


def _check_frame_range(device, n_kp, n_frames, n_hyp, sample):
    k, m1, m2 = read_cameras()
    seq = SyntheticSequence(n_kp, n_frames)
    P = n_frames - 1
    Rt_host = np.stack([seeded_hypotheses(seq.relative_pose(p, p + 1), n_hyp, seed=6 + p) for p in range(P)])
    fe = pipeline.FrontEnd(k, m1, m2)
    tb = pipeline.FrameTables.from_frames(seq.frames, device)
    ws = fe.prepare(tb, n_hyp=n_hyp)
    Rt = torch.from_numpy(Rt_host).to(device)
    fe.run(ws, Rt)
    torch.cuda.synchronize()
    first = {n: getattr(ws, n).clone() for n in ("s_idx", "s_dist", "t_idx", "t_dist", "counts", "inl_rows", "n_inl")}
    fe.run(ws, Rt)                      # idempotence: the same buffers, the same results
    torch.cuda.synchronize()
    for n, v in first.items():
        assert torch.equal(v, getattr(ws, n)), n

    s_idx, s_dist, s_keep = ws.s_idx.cpu().numpy(), ws.s_dist.cpu().numpy(), ws.s_keep.cpu().numpy()
    t_idx, t_dist, t_keep = ws.t_idx.cpu().numpy(), ws.t_dist.cpu().numpy(), ws.t_keep.cpu().numpy()
    rng = np.random.default_rng(0)
    n_cons = ws.cons.count.cpu().numpy()
    counts = ws.counts.cpu().numpy()
    for f, fr in enumerate(seq.frames):
        s = slice(int(tb.offL[f]), int(tb.offL[f + 1]))
        idx, dist, keep = s_idx[s], s_dist[s], s_keep[s]
        # the reported distance is the Hamming distance of the reported pair, for every row
        assert np.array_equal(_ham(fr.descL, fr.descR[idx[:, 0]]), dist[:, 0])
        # row filter on every row, recomputed in float64
        dev = np.abs(fr.kpL[:, 1].astype(np.float64) - fr.kpR[idx[:, 0], 1].astype(np.float64))
        assert np.array_equal(keep.astype(bool), dev <= 2.0)
        # ground truth: a true point's nearest neighbour is its own right-image observation
        true = fr.idL >= 0
        assert (fr.idR[idx[true, 0]] == fr.idL[true]).mean() > 0.999
        # exact parity with the oracle's full scan on sampled queries
        pick = rng.choice(len(fr.descL), size=sample, replace=False)
        ridx, rdist = oracle.hamming_knn(fr.descL[pick], fr.descR, 1, threads=8)
        assert np.array_equal(idx[pick, 0], ridx[:, 0]) and np.array_equal(dist[pick, 0], rdist[:, 0])
        if f < P:
            nx = seq.frames[f + 1]
            idx2, dist2, keep2 = t_idx[s], t_dist[s], t_keep[s]
            assert np.array_equal(_ham(fr.descL, nx.descL[idx2[:, 0]]), dist2[:, 0])
            assert np.array_equal(_ham(fr.descL, nx.descL[idx2[:, 1]]), dist2[:, 1])
            # sortedness incl. the tie rule: (dist, idx) strictly increasing
            assert ((dist2[:, 0] < dist2[:, 1]) | ((dist2[:, 0] == dist2[:, 1]) & (idx2[:, 0] < idx2[:, 1]))).all()
            assert np.array_equal(keep2.astype(bool), dist2[:, 0].astype(np.float64) < 1.0 * dist2[:, 1])
            ridx, rdist = oracle.hamming_knn(fr.descL[pick], nx.descL, 2, threads=8)
            assert np.array_equal(idx2[pick], ridx) and np.array_equal(dist2[pick], rdist)
    # consensus + scoring: exact against the oracle on the first and last pair
    cons = pipeline.consensus_to_host(ws)
    smap = ws.smap.cpu().numpy()
    for p in (0, P - 1):
        tab = cons[p]
        assert len(tab["rows"]) == n_cons[p] > 0.3 * n_kp * 0.6
        r = tab["rows"]
        s0, s1 = int(tb.offL[p]), int(tb.offL[p + 1])
        assert np.array_equal(smap[s0 + r[:, 0]], r[:, 1]) and np.array_equal(smap[s1 + r[:, 2]], r[:, 3])
        assert (np.diff(r[:, 0]) > 0).all()                      # temporal-match order preserved
        rc = oracle.score_hypotheses(Rt_host[p], k, m2[:, 3], tab["X"][:, 0].copy(), tab["X"][:, 1].copy(),
                                     tab["X"][:, 2].copy(), tab["uvL"][:, 0].copy(), tab["uvL"][:, 1].copy(),
                                     tab["uvR"][:, 0].copy(), tab["uvR"][:, 1].copy(), 2.0, threads=8)
        assert np.array_equal(counts[p], rc)
        assert int(ws.best[p]) == int(np.argmax(rc)) and int(ws.n_inl[p]) == rc.max()
        assert rc.max() > 0.9 * n_cons[p]
    return seq, ws


def test_2k_keypoints_frame_range(device):
    _check_frame_range(device, 2000, 12, 256, sample=96)


def test_8k_keypoints_stress(device):
    _check_frame_range(device, 8000, 3, 128, sample=48)


def test_very_long_train_set(device):
    """300 000 train rows against a handful of queries (the library's limit is 2^22 rows per
    item): hundreds of key chunks, planted duplicates near the end, exact against the oracle."""
    from slam_amd import ops
    rng = np.random.default_rng(5)
    D, nt, nq = 61, 300_000, 200
    t = rng.integers(0, 256, (nt, D), dtype=np.uint8)
    t[:, 60] &= 0x3F
    q = rng.integers(0, 256, (nq, D), dtype=np.uint8)
    q[:, 60] &= 0x3F
    for i in range(50):                       # exact copies late in the scan, twice
        t[nt - 1 - 2 * i] = q[i]
        t[nt // 2 + 3 * i] = q[i]
    tq = ops.pack_descriptors(torch.from_numpy(q).to(device), D)
    tt = ops.pack_descriptors(torch.from_numpy(t).to(device), D)
    seg = torch.tensor([[0, nq, 0, nt, 0]], dtype=torch.int32, device=device)
    idx, dist, _ = ops.hamming_match(tq, tt, seg, nq, nt, D, k=2, n_rows=nq)
    ridx, rdist = oracle.hamming_knn(q, t, 2, threads=8)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(dist.cpu().numpy(), rdist)
    assert (ridx[:50, 0] == nt // 2 + 3 * np.arange(50)).all() and (rdist[:50] == 0).all()
