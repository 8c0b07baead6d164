"""GPU tests of the float L2 k-NN (cv2.BFMatcher() on SIFT descriptors, the reference's default
matcher: models/Matcher.py:15) against the C oracle and plain numpy.

cv2 is not available here.  For integer-valued descriptors (SIFT stores integers 0..255 in
float32) every partial sum of squared differences is an exactly representable integer, so any
summation order -- OpenCV's SIMD lanes, the kernel's dimension order, numpy's pairwise sum --
gives the same float32 value: parity on those is bit-exact by arithmetic, not by luck.  For
arbitrary floats the kernel and the oracle share one summation order (bit-exact to each other)
and differ from other orders by rounding only (checked to 1e-5 relative)."""
import numpy as np
import pytest
import torch

from oracle import cpu as oracle
from slam_amd import ops, pipeline
from slam_amd.matcher import HipBFMatcher
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses

pytestmark = pytest.mark.gpu


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _sift_like(rng, n, dim=128):
    return rng.integers(0, 256, (n, dim)).astype(np.float32)


def _gpu_l2(device, q, t, k, mode=0, thr=0.0):
    seg = torch.tensor([[0, len(q), 0, len(t), 0]], dtype=torch.int32, device=device)
    tt = _t(t, device) if len(t) else torch.zeros((1, q.shape[1]), dtype=torch.float32, device=device)
    idx, dist, keep = ops.l2_match(_t(q, device), tt, seg, len(q), len(t), k=k, mode=mode, thr=thr, n_rows=len(q))
    return idx.cpu().numpy(), dist.cpu().numpy(), None if keep is None else keep.cpu().numpy()


@pytest.mark.parametrize("dim", [128, 64])
def test_l2_knn_integer_descriptors_exact(device, dim):
    rng = np.random.default_rng(1)
    q, t = _sift_like(rng, 300, dim), _sift_like(rng, 777, dim)
    t[100] = q[5]; t[300] = q[5]                  # exact duplicates: distance 0 twice, lower index first
    t[401] = t[400]                                # equal distances to every query
    idx, dist, _ = _gpu_l2(device, q, t, 2)
    ridx, rdist = oracle.l2_knn(q, t, 2, threads=4)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert idx[5].tolist() == [100, 300] and dist[5].tolist() == [0.0, 0.0]
    # plain numpy: squared distances as exact integers, float32 sqrt
    d2 = ((q[:, None, :].astype(np.int64) - t[None, :, :].astype(np.int64)) ** 2).sum(-1)
    order = np.lexsort((np.broadcast_to(np.arange(len(t)), d2.shape), d2), axis=1)[:, :2]
    assert np.array_equal(idx, order)
    assert np.array_equal(dist, np.sqrt(np.take_along_axis(d2, order, 1).astype(np.float32)))
    # k = 1 and ragged / tiny train sets
    idx1, dist1, _ = _gpu_l2(device, q, t, 1)
    assert np.array_equal(idx1[:, 0], idx[:, 0]) and (idx1[:, 1] == -1).all()
    assert (dist1[:, 1] == np.finfo(np.float32).max).all()
    for nt in (0, 1, 2):
        i, d, _ = _gpu_l2(device, q[:7], t[:nt], 2)
        ri, rd = oracle.l2_knn(q[:7], t[:nt], 2)
        assert np.array_equal(i, ri) and np.array_equal(d, rd)


def test_l2_knn_float_descriptors_and_filters(device):
    rng = np.random.default_rng(2)
    q = rng.normal(0, 1, (500, 128)).astype(np.float32)
    t = rng.normal(0, 1, (650, 128)).astype(np.float32)
    t[:200] = q[:200] + rng.normal(0, 0.05, (200, 128)).astype(np.float32)     # true partners
    idx, dist, keep = _gpu_l2(device, q, t, 2, mode=1, thr=0.6)
    ridx, rdist = oracle.l2_knn(q, t, 2, threads=4)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)             # same summation order
    d = np.sqrt(((q[:, None, :].astype(np.float64) - t[None]) ** 2).sum(-1))     # any other order: rounding only
    assert np.array_equal(idx[:, 0], d.argmin(1))
    assert np.abs(dist[:, 0] - d.min(1)).max() < 1e-5 * d.max()
    # ratio test as Python evaluates it on the float32 distances
    want = np.array([float(a) < 0.6 * float(b) for a, b in dist], dtype=np.uint8)
    assert np.array_equal(keep, want) and keep[:200].all() and not keep[200:].any()
    # NaN descriptors are never admitted (cv2: d < best is false)
    qn = q[:4].copy(); qn[1, 7] = np.nan
    i, dd, _ = _gpu_l2(device, qn, t, 2)
    assert i[1].tolist() == [-1, -1] and np.array_equal(i[0], idx[0])


def test_hip_bf_matcher_l2_interface(device):
    rng = np.random.default_rng(3)
    q, t = _sift_like(rng, 40), _sift_like(rng, 55)
    m = HipBFMatcher(device)
    out = m.knnMatch(q, t, k=2)
    ridx, rdist = oracle.l2_knn(q, t, 2)
    assert len(out) == 40 and all(len(r) == 2 for r in out)
    for i, row in enumerate(out):
        assert [x.trainIdx for x in row] == ridx[i].tolist() and [x.queryIdx for x in row] == [i, i]
        assert [np.float32(x.distance) for x in row] == rdist[i].tolist()
    with pytest.raises(ValueError):
        m.knnMatch(q, t.astype(np.uint8), k=2)


def test_front_end_on_float_descriptors(device):
    """The batched pipeline with SIFT-like descriptors: same stages, L2 matching."""
    k, m1, m2 = read_cameras()
    Pm, Qm = k @ m1, k @ m2
    seq = SyntheticSequence(400, 3, seed=17)
    rng = np.random.default_rng(4)
    proj = rng.integers(0, 2, (61 * 8, 128)).astype(np.float32)

    def to_float(desc):            # a fixed random embedding of the binary descriptors into 128 integer dims
        bits = np.unpackbits(desc, axis=1).astype(np.float32)
        return np.minimum(bits @ proj, 255.0).astype(np.float32)

    frames = [type(f)(f.kpL, f.kpR, to_float(f.descL), to_float(f.descR), f.idL, f.idR, f.Rt) for f in seq.frames]
    H = 50
    fe = pipeline.FrontEnd(k, m1, m2, ratio_thresh=0.8)
    tb = pipeline.FrameTables.from_frames(frames, device)
    assert tb.i8L is not None and tb.i8R is not None          # integer-valued: the int8 matrix-core matcher
    # the same frames with one non-integer entry keep to the fp32 matcher, with the same matches
    odd = [type(f)(f.kpL, f.kpR, f.descL.copy(), f.descR.copy(), f.idL, f.idR, f.Rt) for f in frames]
    odd[1].descR[3, 5] += 0.25
    tb_f = pipeline.FrameTables.from_frames(odd, device)
    assert tb_f.i8L is None and tb_f.i8R is None
    ws = fe.prepare(tb, n_hyp=H)
    ws_f = fe.prepare(tb_f, n_hyp=H)
    import ctypes
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    fe.temporal_match(ws, stream)
    fe.temporal_match(ws_f, stream)
    torch.cuda.synchronize()
    nv = int(ws.seg_t[:, 1].sum())                            # rows of the last frame have no successor to match
    assert all(torch.equal(getattr(ws, n)[:nv], getattr(ws_f, n)[:nv]) for n in ("t_idx", "t_dist", "t_keep"))
    Rt_host = np.stack([seeded_hypotheses(seq.relative_pose(p, p + 1), H, seed=p) for p in range(2)])
    fe.run(ws, _t(Rt_host, device))
    torch.cuda.synchronize()
    cons = pipeline.consensus_to_host(ws)
    st = []
    for f in frames:
        idx, _ = oracle.l2_knn(f.descL, f.descR, 1, threads=4)
        keep = oracle.row_filter(idx, f.kpL[:, 1], f.kpR[:, 1], 2.0)
        j = np.where(keep > 0, idx[:, 0], 0)
        X, Y, Z, valid = oracle.triangulate_dlt(Pm, Qm, f.kpL[:, 0], f.kpL[:, 1], f.kpR[j, 0], f.kpR[j, 1])
        st.append((np.where(keep > 0, idx[:, 0], -1).astype(np.int32), (valid * (keep > 0)).astype(np.uint8)))
    for p in range(2):
        tidx, tdist = oracle.l2_knn(frames[p].descL, frames[p + 1].descL, 2, threads=4)
        tkeep = np.array([float(a) < 0.8 * float(b) for a, b in tdist], dtype=np.uint8)
        rows = oracle.join(tidx, tkeep, st[p][0], st[p][1], st[p + 1][0])
        assert np.array_equal(cons[p]["rows"], rows) and len(rows) > 60
        assert int(ws.n_inl[p]) > 0.8 * len(rows)


# ---- integer-valued descriptors on the int8 matrix cores ---------------------------------------
def _gpu_l2_int8(device, q, t, k, mode=0, thr=0.0, yq=None, yt=None):
    seg = torch.tensor([[0, len(q), 0, len(t), 0]], dtype=torch.int32, device=device)
    dim = q.shape[1]
    tq = ops.l2_pack_int8(_t(q, device))
    tt = ops.l2_pack_int8(_t(t, device) if len(t) else torch.zeros((0, dim), dtype=torch.float32, device=device))
    assert tq.usable() and tt.usable()
    idx, dist, keep = ops.l2_match_int8(tq, tt, seg, len(q), len(t), k=k, mode=mode, thr=thr,
                                        yq=None if yq is None else _t(yq, device),
                                        yt=None if yt is None else _t(yt, device), n_rows=len(q))
    return idx.cpu().numpy(), dist.cpu().numpy(), None if keep is None else keep.cpu().numpy()


def test_l2_pack_int8_values_norms_and_rejects(device):
    rng = np.random.default_rng(3)
    d = _sift_like(rng, 37, 128)
    d[0, :] = 0.0
    d[1, :] = 255.0
    tb = ops.l2_pack_int8(_t(d, device))
    assert tb.usable()
    want = (d.astype(np.int32) - 128)
    assert np.array_equal(tb.desc.cpu().numpy().astype(np.int32), want)
    assert np.array_equal(tb.norm2.cpu().numpy()[:37], (want * want).sum(1))
    for badval in (0.5, -1.0, 256.0, np.nan, np.inf):
        e = d.copy()
        e[5, 77] = badval
        assert not ops.l2_pack_int8(_t(e, device)).usable()
    assert ops.l2_pack_int8(torch.zeros((0, 64), dtype=torch.float32, device=device)).usable()


@pytest.mark.parametrize("dim", [128, 64])
def test_l2_int8_equals_oracle_and_float_kernel(device, dim):
    rng = np.random.default_rng(5 + dim)
    q, t = _sift_like(rng, 333, dim), _sift_like(rng, 1500, dim)
    # ties: duplicates of one train row at a tile edge, a super-tile edge and both sides of a key
    # chunk edge; an exact copy of a query twice; extreme rows (all 0 / all 255: largest norms)
    for j in (31, 32, 63, 64, 255, 256, 257, 1023, 1024):
        t[j] = t[7]
    t[700] = q[9]; t[1300] = q[9]
    t[500] = 0.0; t[501] = 255.0; q[3] = 255.0; q[4] = 0.0
    ridx, rdist = oracle.l2_knn(q, t, 2, threads=4)
    for k in (2, 1):
        idx, dist, _ = _gpu_l2_int8(device, q, t, k)
        fidx, fdist, _ = _gpu_l2(device, q, t, k)
        assert np.array_equal(idx, fidx) and np.array_equal(dist, fdist)
        if k == 2:
            assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert idx is not None and _gpu_l2_int8(device, q, t, 2)[0][9].tolist() == [700, 1300]
    # ragged / tiny / empty train sets, sizes around the tile edges
    for nt in (0, 1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257):
        i, d, _ = _gpu_l2_int8(device, q[:40], t[:nt], 2)
        ri, rd = oracle.l2_knn(q[:40], t[:nt], 2)
        assert np.array_equal(i, ri) and np.array_equal(d, rd), nt
    # fused filters
    yq, yt = rng.uniform(0, 300, len(q)).astype(np.float32), rng.uniform(0, 300, len(t)).astype(np.float32)
    for mode, thr in ((1, 0.8), (1, 1.0), (2, 20.0)):
        a = _gpu_l2_int8(device, q, t, 2 if mode == 1 else 1, mode, thr, yq, yt)
        seg = torch.tensor([[0, len(q), 0, len(t), 0]], dtype=torch.int32, device=device)
        b = ops.l2_match(_t(q, device), _t(t, device), seg, len(q), len(t), k=2 if mode == 1 else 1, mode=mode, thr=thr,
                         yq=_t(yq, device), yt=_t(yt, device), n_rows=len(q))
        assert np.array_equal(a[0], b[0].cpu().numpy()) and np.array_equal(a[2], b[2].cpu().numpy())


def test_l2_int8_batched_ragged_wide_and_narrow(device):
    """Many ragged items in one launch (the four-tiles-per-wave variant once the grid is large
    enough, the one-tile variant otherwise) against the float kernel."""
    rng = np.random.default_rng(17)
    for batch, hi in ((3, 700), (600, 300)):
        nq = rng.integers(0, hi, batch)
        nt = rng.integers(0, hi, batch)
        nq[0], nt[0] = hi, hi + 13
        if batch > 2:
            nq[1], nt[2] = 0, 0
        q, t = _sift_like(rng, int(nq.sum()), 128), _sift_like(rng, int(nt.sum()), 128)
        qo, to = np.concatenate([[0], np.cumsum(nq)]), np.concatenate([[0], np.cumsum(nt)])
        seg = torch.from_numpy(np.stack([qo[:-1], nq, to[:-1], nt, qo[:-1]], 1).astype(np.int32)).to(device)
        tq, tt = ops.l2_pack_int8(_t(q, device)), ops.l2_pack_int8(_t(t, device))
        a = ops.l2_match_int8(tq, tt, seg, int(nq.max()), int(nt.max()), k=2, mode=1, thr=0.7, n_rows=len(q))
        b = ops.l2_match(_t(q, device), _t(t, device), seg, int(nq.max()), int(nt.max()), k=2, mode=1, thr=0.7,
                         n_rows=len(q))
        for x, y in zip(a, b):
            assert torch.equal(x, y)
