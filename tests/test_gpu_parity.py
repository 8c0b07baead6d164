"""GPU parity tests: every HIP kernel, through the C ABI, against the CPU oracle.

Bars: bit-exact for match indices / distances / filter masks / join rows / supporter
counts and masks; triangulated points within 1e-4 relative of the LAPACK solution
(the contract) and within 1e-9 of the C oracle that runs the same Jacobi iteration.
"""
import numpy as np
import pytest
import torch

from oracle import cpu as oracle
from oracle import numpy_twin as twin
from slam_amd import ops
from slam_amd.synthetic import SyntheticSequence, read_cameras, seeded_hypotheses

from helpers import gpu_match, oracle_match

pytestmark = pytest.mark.gpu


def _rand_desc(rng, n, D):
    return rng.integers(0, 256, size=(n, D), dtype=np.uint8)


# --------------------------------------------------------------------------- knn
@pytest.fixture(params=["fp4", "valu", "wide", "mfma"], autouse=True)
def kernel_impl(request):
    """Every test in this file runs against both implementations of the k-NN kernel (fp4 matrix
    cores, VALU popcount) and the forms of the filtered scoring kernel (fp32 first level with the
    points of a set split over a block or not, fp64 only, fp32 first level on the matrix cores)."""
    from slam_amd import _lib
    knn, score = {"fp4": ("fp4", "split"), "valu": ("valu", "f64"), "wide": ("fp4", "wide"),
                  "mfma": ("fp4", "mfma")}[request.param]
    _lib.select_kernels(knn, score)
    yield request.param
    _lib.select_kernels("fp4", "auto")


@pytest.mark.parametrize("D,stride", [(61, 64), (64, 64), (32, 32), (32, 64), (17, 32)])
@pytest.mark.parametrize("k", [1, 2])
def test_knn_random_ragged(device, D, stride, k):
    rng = np.random.default_rng(100 + D + k)
    sizes = [(1, 1), (3, 2), (64, 64), (65, 257), (300, 513), (1000, 777), (257, 1)]
    items = [(_rand_desc(rng, a, D), _rand_desc(rng, b, D)) for a, b in sizes]
    idx, dist, _ = gpu_match(items, D, stride, k, device)
    ridx, rdist = oracle_match(items, D, k)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist, rdist)


def test_knn_large_batch_selects_wide_kernel(device):
    # enough blocks that the 4-queries-per-lane variant is chosen
    rng = np.random.default_rng(7)
    items = [(_rand_desc(rng, 500 + 7 * b, 61), _rand_desc(rng, 480 + 5 * b, 61)) for b in range(80)]
    idx, dist, _ = gpu_match(items, 61, 64, 2, device)
    ridx, rdist = oracle_match(items, 61, 2)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist, rdist)
    # the 2-queries-per-lane variant
    items = items[:24]
    idx, dist, _ = gpu_match(items, 61, 64, 2, device)
    ridx, rdist = oracle_match(items, 61, 2)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)


def test_knn_ties_lowest_index_wins(device):
    rng = np.random.default_rng(3)
    D = 61
    base = _rand_desc(rng, 40, D)
    # train set made of repeated rows: every distance value occurs many times
    t = np.concatenate([base, base, base[::-1], base])
    q = np.concatenate([base, _rand_desc(rng, 30, D)])
    # all-equal distances: train rows identical
    t_same = np.repeat(base[:1], 300, axis=0)
    # few-bit descriptors: massive distance collisions
    q_few = (_rand_desc(rng, 200, D) & 1).astype(np.uint8)
    t_few = (_rand_desc(rng, 700, D) & 1).astype(np.uint8)
    items = [(q, t), (q, t_same), (q_few, t_few), (np.zeros((5, D), np.uint8), np.zeros((600, D), np.uint8))]
    for k in (1, 2):
        idx, dist, _ = gpu_match(items, D, 64, k, device)
        ridx, rdist = oracle_match(items, D, k)
        assert np.array_equal(idx, ridx)
        assert np.array_equal(dist, rdist)
    # spot-check the rule itself, independent of the oracle
    idx, dist, _ = gpu_match([(base, t)], D, 64, 2, device)
    assert np.array_equal(idx[:, 0], np.arange(40))       # first copy wins
    assert np.array_equal(idx[:, 1], np.arange(40) + 40)  # second copy is the runner-up
    assert (dist == 0).all()


def test_knn_small_train_sets(device):
    rng = np.random.default_rng(5)
    D = 32
    q = _rand_desc(rng, 70, D)
    items = [(q, _rand_desc(rng, 0, D)), (q, _rand_desc(rng, 1, D)), (q, _rand_desc(rng, 2, D)),
             (_rand_desc(rng, 0, D), _rand_desc(rng, 9, D))]
    idx, dist, _ = gpu_match(items, D, 32, 2, device)
    ridx, rdist = oracle_match(items, D, 2)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert (idx[:70] == -1).all() and (idx[70:140, 1] == -1).all()


def test_knn_matches_python_twin(device):
    # the slow statement-by-statement twin of cv2's scan agrees with both
    rng = np.random.default_rng(11)
    q = (_rand_desc(rng, 20, 8) & 3).astype(np.uint8)
    t = (_rand_desc(rng, 33, 8) & 3).astype(np.uint8)
    ref = twin.knn_match(q, t, 2)
    idx, dist, _ = gpu_match([(q, t)], 8, 32, 2, device)
    for i, lst in enumerate(ref):
        assert [m.trainIdx for m in lst] == list(idx[i])
        assert [int(m.distance) for m in lst] == list(dist[i])


def test_ratio_and_row_filters(device):
    seq = SyntheticSequence(700, 2)
    f0, f1 = seq.frames
    D = 61
    for thr in (1.0, 0.6, 0.2):
        idx, dist, keep = gpu_match([(f0.descL, f1.descL)], D, 64, 2, device, mode=1, thr=thr)
        ridx, rdist = oracle_match([(f0.descL, f1.descL)], D, 2)
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
        assert np.array_equal(keep, oracle.ratio_filter(ridx, rdist, thr))
    for thr in (2.0, 1.0):
        idx, dist, keep = gpu_match([(f0.descL, f0.descR)], D, 64, 1, device, mode=2, thr=thr,
                                    yq=f0.kpL[:, 1], yt=f0.kpR[:, 1])
        ridx, _ = oracle_match([(f0.descL, f0.descR)], D, 1)
        assert np.array_equal(idx, ridx)
        assert np.array_equal(keep, oracle.row_filter(ridx, f0.kpL[:, 1], f0.kpR[:, 1], thr))


def test_ratio_filter_is_python_float_compare(device):
    # every (d1, d2) in a grid, against Python's own float compare (models/Matcher.py:182)
    D = 64
    d_vals = list(range(0, 200, 3)) + [255, 256, 300, 488, 511, 512]
    zero = np.zeros((1, D), np.uint8)

    def with_bits(n):
        r = np.zeros(D * 8, np.uint8)
        r[:n] = 1
        return np.packbits(r, bitorder="little")[None, :]

    items = []
    pairs = []
    for a in d_vals:
        for b in d_vals:
            if a <= b:
                items.append((zero, np.concatenate([with_bits(a), with_bits(b)])))
                pairs.append((a, b))
    for thr in (0.2, 0.4, 0.6, 0.7, 1.0):
        idx, dist, keep = gpu_match(items, D, 64, 2, device, mode=1, thr=thr)
        assert [tuple(r) for r in dist] == pairs
        expect = [float(np.float32(a)) < thr * float(np.float32(b)) for a, b in pairs]
        assert list(keep.astype(bool)) == expect


def test_row_filter_edges(device):
    D = 32
    q = np.zeros((6, D), np.uint8)
    t = np.zeros((1, D), np.uint8)
    two = np.float32(2.0)
    yq = np.array([10.0, 10.0, 10.0, 10.0, 10.0, 10.0], np.float32)
    up = np.nextafter(np.float32(12.0), np.float32(20.0))
    dn = np.nextafter(np.float32(12.0), np.float32(0.0))
    for yt_val, expect in ((np.float32(12.0), 1), (up, 0), (dn, 1), (np.float32(8.0), 1), (np.float32(7.9999), 0)):
        idx, dist, keep = gpu_match([(q, t)], D, 32, 1, device, mode=2, thr=float(two), yq=yq,
                                    yt=np.array([yt_val], np.float32))
        assert (keep == expect).all(), (yt_val, keep)


def test_pack_descriptors(device):
    rng = np.random.default_rng(2)
    for D, stride in ((61, 64), (32, 32), (17, 32), (64, 64)):
        d = _rand_desc(rng, 333, D)
        out = ops.pack_descriptors(torch.from_numpy(d).to(device), D, stride).cpu().numpy()
        assert out.shape == (333, stride)
        assert np.array_equal(out[:, :D], d) and (out[:, D:] == 0).all()
    # a strided source: take the first 61 of 80 bytes
    d = _rand_desc(rng, 50, 80)
    out = ops.pack_descriptors(torch.from_numpy(d).to(device), 61, 64).cpu().numpy()
    assert np.array_equal(out[:, :61], d[:, :61]) and (out[:, 61:] == 0).all()


# --------------------------------------------------------------------------- DLT
def _stereo_inputs(n_kp=1500):
    seq = SyntheticSequence(n_kp, 1)
    f0 = seq.frames[0]
    idx, _ = oracle.hamming_knn(f0.descL, f0.descR, 1, threads=4)
    keep = oracle.row_filter(idx, f0.kpL[:, 1], f0.kpR[:, 1], 2.0)
    return seq, f0, idx, keep


def test_dlt_parity(device):
    seq, f0, idx, keep = _stereo_inputs()
    k, m1, m2 = read_cameras()
    P, Q = k @ m1, k @ m2
    sel = np.flatnonzero(keep)
    x1, y1 = f0.kpL[sel, 0].copy(), f0.kpL[sel, 1].copy()
    x2, y2 = f0.kpR[idx[sel, 0], 0].copy(), f0.kpR[idx[sel, 0], 1].copy()
    t = lambda a: torch.from_numpy(a).to(device)
    X, Y, Z, valid = ops.triangulate_dlt(P, Q, t(x1), t(y1), t(x2), t(y2))
    got = np.stack([X.cpu().numpy(), Y.cpu().numpy(), Z.cpu().numpy()], axis=1)
    oX, oY, oZ, ovalid = oracle.triangulate_dlt(P, Q, x1, y1, x2, y2)
    ref = np.stack([oX, oY, oZ], axis=1)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert (np.abs(got - ref) / scale).max() < 1e-9
    assert np.array_equal(valid.cpu().numpy(), ovalid)
    # the reference's own arithmetic: np.linalg.svd per point (utils/utils.py:553)
    for n in range(0, len(sel), 7):
        s = twin.least_squares((float(x1[n]), float(y1[n])), (float(x2[n]), float(y2[n])), P, Q)
        assert np.abs(got[n] - s).max() / np.abs(s).max() < 1e-4


def test_dlt_extreme_disparities(device):
    # disparity from 0.05 px (Z ~ 7.6 km) to 200 px, noisy rows, general cameras
    rng = np.random.default_rng(4)
    k, m1, m2 = read_cameras()
    P, Q = k @ m1, k @ m2
    n = 2000
    disp = np.exp(rng.uniform(np.log(0.05), np.log(200.0), n))
    x1 = rng.uniform(250, 1200, n).astype(np.float32)
    y1 = rng.uniform(0, 370, n).astype(np.float32)
    x2 = (x1 - disp).astype(np.float32)
    y2 = (y1 + rng.normal(0, 0.7, n)).astype(np.float32)
    x2[:50] = x1[:50] + 3.0  # negative disparity: rejected by x1 < x2 / Z <= 0
    t = lambda a: torch.from_numpy(a).to(device)
    X, Y, Z, valid = ops.triangulate_dlt(P, Q, t(x1), t(y1), t(x2), t(y2))
    got = np.stack([X.cpu().numpy(), Y.cpu().numpy(), Z.cpu().numpy()], axis=1)
    oX, oY, oZ, ovalid = oracle.triangulate_dlt(P, Q, x1, y1, x2, y2)
    ref = np.stack([oX, oY, oZ], axis=1)
    assert (np.abs(got - ref) / np.abs(ref).max(axis=1, keepdims=True)).max() < 1e-9
    assert np.array_equal(valid.cpu().numpy(), ovalid)
    assert (valid.cpu().numpy()[:50] == 0).all()
    for i in range(0, n, 11):
        s = twin.least_squares((float(x1[i]), float(y1[i])), (float(x2[i]), float(y2[i])), P, Q)
        assert np.abs(got[i] - s).max() / np.abs(s).max() < 1e-4


# --------------------------------------------------------------------- pipeline pieces
def _oracle_frame(f):
    idx, _ = oracle.hamming_knn(f.descL, f.descR, 1, threads=4)
    keep = oracle.row_filter(idx, f.kpL[:, 1], f.kpR[:, 1], 2.0)
    k, m1, m2 = read_cameras()
    P, Q = k @ m1, k @ m2
    j = np.where(keep > 0, idx[:, 0], 0)
    X, Y, Z, valid = oracle.triangulate_dlt(P, Q, f.kpL[:, 0], f.kpL[:, 1], f.kpR[j, 0], f.kpR[j, 1])
    valid = valid * (keep > 0)
    smap = np.where(keep > 0, idx[:, 0], -1).astype(np.int32)
    return smap, X, Y, Z, valid.astype(np.uint8)


def test_matched_triangulation_and_join(device):
    seq = SyntheticSequence(900, 3)
    k, m1, m2 = read_cameras()
    P, Q = k @ m1, k @ m2
    F = seq.frames
    nL = [f.descL.shape[0] for f in F]
    offL = np.concatenate([[0], np.cumsum(nL)])
    pad = lambda d: np.concatenate([d, np.zeros((d.shape[0], 3), np.uint8)], axis=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    dL = t(np.concatenate([pad(f.descL) for f in F]))
    dR = t(np.concatenate([pad(f.descR) for f in F]))
    xL = t(np.concatenate([f.kpL[:, 0] for f in F])); yL = t(np.concatenate([f.kpL[:, 1] for f in F]))
    xR = t(np.concatenate([f.kpR[:, 0] for f in F])); yR = t(np.concatenate([f.kpR[:, 1] for f in F]))
    seg_s = ops.make_segments(offL[:-1], nL, offL[:-1], nL, offL[:-1], device)
    idx, dist, keep = ops.hamming_match(dL, dR, seg_s, max(nL), max(nL), 61, k=1, mode=2, thr=2.0, yq=yL, yt=yR)
    X, Y, Z, valid, smap = ops.triangulate_matched(P, Q, xL, yL, xR, yR, seg_s, max(nL), idx, keep)
    ref = [_oracle_frame(f) for f in F]
    rsmap = np.concatenate([r[0] for r in ref])
    rvalid = np.concatenate([r[4] for r in ref])
    assert np.array_equal(smap.cpu().numpy(), rsmap)
    assert np.array_equal(valid.cpu().numpy(), rvalid)
    m = rsmap >= 0
    for got, r in ((X, 1), (Y, 2), (Z, 3)):
        rr = np.concatenate([x[r] for x in ref])
        assert np.allclose(got.cpu().numpy()[m], rr[m], rtol=1e-9, atol=0)
    assert torch.isnan(X[torch.from_numpy(~m).to(device)]).all()

    # temporal matching of pairs (0,1), (1,2) + join
    seg_t = ops.make_segments(offL[:-2], nL[:-1], offL[1:-1], nL[1:], offL[:-2], device)
    tidx, tdist, tkeep = ops.hamming_match(dL, dL, seg_t, max(nL), max(nL), 61, k=2, mode=1, thr=1.0)
    pair = torch.tensor([[offL[p], nL[p], offL[p + 1], nL[p + 1], offL[p], offL[p + 1], offL[p], 0] for p in range(2)],
                        dtype=torch.int32, device=device)
    cap = max(nL)
    cons = ops.join_consensus(pair, max(nL), tidx, tkeep, smap, valid, X, Y, Z, xL, yL, xR, yR, cap)
    counts = cons.count.cpu().numpy()
    for p in range(2):
        ridx, rdist = oracle.hamming_knn(F[p].descL, F[p + 1].descL, 2, threads=4)
        rkeep = oracle.ratio_filter(ridx, rdist, 1.0)
        rows = oracle.join(ridx, rkeep, ref[p][0], ref[p][4], ref[p + 1][0])
        assert counts[p] == rows.shape[0] and counts[p] > 100
        got = cons.rows.cpu().numpy()[p * cap:p * cap + counts[p]]
        assert np.array_equal(got, rows)
        sl = slice(p * cap, p * cap + counts[p])
        assert np.allclose(cons.X.cpu().numpy()[sl], ref[p][1][rows[:, 0]], rtol=1e-9, atol=0)
        assert np.array_equal(cons.uL.cpu().numpy()[sl], F[p + 1].kpL[rows[:, 2], 0])
        assert np.array_equal(cons.vL.cpu().numpy()[sl], F[p + 1].kpL[rows[:, 2], 1])
        assert np.array_equal(cons.uR.cpu().numpy()[sl], F[p + 1].kpR[rows[:, 3], 0])
        assert np.array_equal(cons.vR.cpu().numpy()[sl], F[p + 1].kpR[rows[:, 3], 1])
        # the statement-by-statement twin of utils/utils.py:338-386 gives the same rows
        tw_matches = [[twin.DMatchLike(q, ridx[q, 0], rdist[q, 0])] for q in range(nL[p]) if rkeep[q]]
        prev_map = {int(q): int(v) for q, v in enumerate(ref[p][0]) if v >= 0}
        cur_map = {int(q): int(v) for q, v in enumerate(ref[p + 1][0]) if v >= 0}
        d3 = {int(q): 0 for q in np.flatnonzero(ref[p][4])}
        cm, _ = twin.matches_in_all_4_images(tw_matches, prev_map, cur_map, d3)
        assert [c[:4] for c in cm] == [tuple(r) for r in rows]


# ----------------------------------------------------------------------- scoring
def _consensus_case(n_kp=1200, H=600, seed=6):
    seq = SyntheticSequence(n_kp, 2, seed=seed)
    f0, f1 = seq.frames
    s0, s1 = _oracle_frame(f0), _oracle_frame(f1)
    ridx, rdist = oracle.hamming_knn(f0.descL, f1.descL, 2, threads=4)
    rkeep = oracle.ratio_filter(ridx, rdist, 1.0)
    rows = oracle.join(ridx, rkeep, s0[0], s0[4], s1[0])
    X, Y, Z = s0[1][rows[:, 0]], s0[2][rows[:, 0]], s0[3][rows[:, 0]]
    uL, vL = f1.kpL[rows[:, 2], 0], f1.kpL[rows[:, 2], 1]
    uR, vR = f1.kpR[rows[:, 3], 0], f1.kpR[rows[:, 3], 1]
    Rt = seeded_hypotheses(seq.relative_pose(0, 1), H)
    return seq, rows, (X, Y, Z, uL, vL, uR, vR), Rt


def _gpu_score(device, Rt, k, tr, pts, thresh, want_masks=True, cap=None):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    X, Y, Z, uL, vL, uR, vR = pts
    n = len(X)
    cap = n if cap is None else cap
    padf = lambda a, dt: np.concatenate([a.astype(dt), np.zeros(cap - n, dt)])
    args = (t(Rt), k, tr, t(padf(X, np.float64)), t(padf(Y, np.float64)), t(padf(Z, np.float64)),
            t(padf(uL, np.float32)), t(padf(vL, np.float32)), t(padf(uR, np.float32)), t(padf(vR, np.float32)))
    out = ops.score_hypotheses(*args, thresh=thresh, cap=cap, n_host=n, want_masks=want_masks, fast=False)
    # the filtered fast path must give the very same counts as the plain evaluation
    fast = ops.score_hypotheses(*args, thresh=thresh, cap=cap, n_host=n, want_masks=False, fast=True)
    torch.cuda.synchronize()
    assert torch.equal(fast, out[0] if want_masks else out), "fast path disagrees with the exact kernel"
    return out


def test_score_exact_counts_and_masks(device):
    seq, rows, pts, Rt = _consensus_case()
    k, m1, m2 = read_cameras()
    tr = m2[:, 3]
    counts, masks = _gpu_score(device, Rt, k, tr, pts, 2.0)
    rc, rm = oracle.score_hypotheses(Rt, k, tr, *pts, 2.0, want_masks=True, threads=4)
    assert np.array_equal(counts.cpu().numpy()[0], rc)
    assert np.array_equal(masks.cpu().numpy()[0].view(np.uint64), rm)
    assert rc.max() > 0.8 * len(pts[0]) and rc.min() < 0.5 * len(pts[0])
    # the statement-by-statement numpy twin agrees on a sample of hypotheses
    f1 = seq.frames[1]
    cm = [(r[0], r[1], r[2], r[3], np.array([pts[0][i], pts[1][i], pts[2][i]])) for i, r in enumerate(rows)]
    for h in (0, 1, 2, 50, 333):
        good, num = twin.find_supporters(Rt[h], m2, cm, k, f1.kpL, f1.kpR, thresh=2)
        assert num == rc[h]


def test_score_threshold_knife_edge(device):
    # points whose reprojection distance lands within a few ulp of the threshold
    k, m1, m2 = read_cameras()
    tr = m2[:, 3]
    rng = np.random.default_rng(9)
    n = 640
    X = rng.uniform(-20, 20, n); Y = rng.uniform(-3, 2, n); Z = rng.uniform(5, 80, n)
    Rt = np.hstack([np.eye(3), np.zeros((3, 1))])[None].repeat(3, axis=0).copy()
    Rt[1, :, 3] = [0.01, -0.02, 0.03]
    Rt[2, :, 3] = [-0.3, 0.1, 0.2]
    uL = k[0, 0] * X / Z + k[0, 2]; vL = k[1, 1] * Y / Z + k[1, 2]
    uR = k[0, 0] * (X + tr[0]) / Z + k[0, 2]; vR = vL.copy()
    # displace the observation by exactly-2-ish pixels in assorted directions; float32
    # storage rounds them to either side of the threshold
    ang = rng.uniform(0, 2 * np.pi, n)
    uL32 = (uL + 2.0 * np.cos(ang)).astype(np.float32)
    vL32 = (vL + 2.0 * np.sin(ang)).astype(np.float32)
    uR32 = uR.astype(np.float32); vR32 = vR.astype(np.float32)
    # integer-pixel geometry: distances that are exactly 2.0 in fp64
    X[:64] = 0.0; Y[:64] = 0.0; Z[:64] = 10.0
    uL32[:64] = np.float32(k[0, 2]); vL32[:64] = np.float32(k[1, 2] + 2.0)
    pts = (X, Y, Z, uL32, vL32, uR32, vR32)
    for thresh in (2.0, 1.9999999999999998, 2.0000000000000004, 0.5, 1e-9, 1e9):
        counts, masks = _gpu_score(device, Rt, k, tr, pts, thresh)
        rc, rm = oracle.score_hypotheses(Rt, k, tr, *pts, thresh, want_masks=True)
        assert np.array_equal(counts.cpu().numpy()[0], rc), thresh
        assert np.array_equal(masks.cpu().numpy()[0].view(np.uint64), rm), thresh


def test_score_degenerate_inputs(device):
    # zero depth, negative depth, NaN/Inf coordinates: same verdicts as IEEE on the host
    k, m1, m2 = read_cameras()
    tr = m2[:, 3]
    X = np.array([0.0, 1.0, 0.0, np.nan, 1.0, 1e300, 0.0, 2.0])
    Y = np.array([0.0, 1.0, 0.0, 0.0, np.inf, 0.0, 0.0, -1.0])
    Z = np.array([0.0, -5.0, 1e-300, 3.0, 3.0, 1e-300, 5.0, 7.0])
    uL = np.array([601.8873, 460.0, 601.0, 600.0, 600.0, 0.0, 601.8873, 803.9], np.float32)
    vL = np.array([183.1104, 41.0, 183.0, 183.0, 183.0, 0.0, 183.1104, 82.1], np.float32)
    uR = (uL - np.float32(50.0)).astype(np.float32)
    vR = vL.copy()
    Rt = np.hstack([np.eye(3), np.zeros((3, 1))])[None].repeat(2, axis=0).copy()
    Rt[1, 2, 3] = -5.0
    pts = (X, Y, Z, uL, vL, uR, vR)
    counts, masks = _gpu_score(device, Rt, k, tr, pts, 2.0)
    with np.errstate(all="ignore"):
        rc, rm = oracle.score_hypotheses(Rt, k, tr, *pts, 2.0, want_masks=True)
    assert np.array_equal(counts.cpu().numpy()[0], rc)
    assert np.array_equal(masks.cpu().numpy()[0].view(np.uint64), rm)


def test_score_batched_sets_device_counts(device):
    # several point sets with device-side counts, cap > n, H not a multiple of the block
    k, m1, m2 = read_cameras()
    tr = m2[:, 3]
    cases = [_consensus_case(500, 37, seed=s) for s in (6, 7, 8)]
    cap = max(len(c[2][0]) for c in cases) + 13
    H = 37
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    cat = lambda j, dt: np.concatenate([np.concatenate([c[2][j].astype(dt), np.full(cap - len(c[2][j]), 7, dt)])
                                        for c in cases])
    Rt = np.stack([c[3] for c in cases])
    n_dev = torch.tensor([len(c[2][0]) for c in cases], dtype=torch.int32, device=device)
    args = (t(Rt), k, tr, t(cat(0, np.float64)), t(cat(1, np.float64)), t(cat(2, np.float64)),
            t(cat(3, np.float32)), t(cat(4, np.float32)), t(cat(5, np.float32)), t(cat(6, np.float32)))
    counts, masks = ops.score_hypotheses(*args, thresh=2.0, cap=cap, n_dev=n_dev, want_masks=True)
    fast = ops.score_hypotheses(*args, thresh=2.0, cap=cap, n_dev=n_dev, fast=True)
    assert torch.equal(fast, counts)
    words = (cap + 63) // 64
    for s, c in enumerate(cases):
        rc, rm = oracle.score_hypotheses(c[3], k, tr, *c[2], 2.0, want_masks=True)
        assert np.array_equal(counts.cpu().numpy()[s], rc)
        gm = masks.cpu().numpy()[s].view(np.uint64)
        assert np.array_equal(gm[:, :rm.shape[1]], rm)
        assert (gm[:, rm.shape[1]:] == 0).all() and gm.shape[1] == words


def test_score_empty_and_tiny(device):
    k, m1, m2 = read_cameras()
    tr = m2[:, 3]
    Rt = seeded_hypotheses(np.hstack([np.eye(3), np.zeros((3, 1))]), 5)
    z64, z32 = np.zeros(0), np.zeros(0, np.float32)
    counts = _gpu_score(device, Rt, k, tr, (z64, z64, z64, z32, z32, z32, z32), 2.0, want_masks=False, cap=4)
    assert (counts.cpu().numpy() == 0).all()
    one = (np.array([0.0]), np.array([0.0]), np.array([10.0]), np.array([601.8873], np.float32),
           np.array([183.1104], np.float32), np.array([601.8873 - 37.98145], np.float32),
           np.array([183.1104], np.float32))
    counts = _gpu_score(device, Rt[:1] * 0 + np.hstack([np.eye(3), np.zeros((3, 1))]), k, tr, one, 2.0,
                        want_masks=False)
    assert counts.cpu().numpy().tolist() == [[1]]


def test_score_fast_path_general_k_and_scale(device):
    seq, rows, pts, Rt = _consensus_case(1500, 2000)
    k, m1, m2 = read_cameras()
    tr = m2[:, 3]
    rc = oracle.score_hypotheses(Rt, k, tr, *pts, 2.0, threads=8)
    counts = _gpu_score(device, Rt, k, tr, pts, 2.0, want_masks=False)
    assert np.array_equal(counts.cpu().numpy()[0], rc)
    # skewed intrinsics: not a pinhole pattern, the library must take the plain kernel
    k2 = k.copy()
    k2[0, 1] = 0.37
    k2[2, 0] = 1e-6
    rc2 = oracle.score_hypotheses(Rt[:300], k2, tr, *pts, 2.0, threads=8)
    counts2 = _gpu_score(device, Rt[:300], k2, tr, pts, 2.0, want_masks=False)
    assert np.array_equal(counts2.cpu().numpy()[0], rc2)
    # a right camera that is not purely sideways (t_right[1], t_right[2] != 0): the other filter variant
    tr2 = np.array([tr[0], 0.013, -0.021])
    rc4 = oracle.score_hypotheses(Rt[:700], k, tr2, *pts, 2.0, threads=8)
    c4 = _gpu_score(device, Rt[:700], k, tr2, pts, 2.0, want_masks=False)
    assert np.array_equal(c4.cpu().numpy()[0], rc4) and rc4.max() > 0
    # other thresholds, including sub-pixel ones where most evaluations sit near the band
    for thresh in (0.05, 0.4, 1.0, 3.5, 50.0):
        rc3 = oracle.score_hypotheses(Rt[:500], k, tr, *pts, thresh, threads=8)
        c3 = _gpu_score(device, Rt[:500], k, tr, pts, thresh, want_masks=False)
        assert np.array_equal(c3.cpu().numpy()[0], rc3), thresh


def _band_case(rng, n, k, tr, thresh, scale=1.0):
    """Points observed by an identity pose with both observations displaced from the exact
    projection by thresh*(1+delta), delta swept over +-[1e-9, 1e-2]: every scale of the guard
    bands of the filtered kernels gets hit from both sides."""
    X = rng.uniform(-20, 20, n) * scale
    Y = rng.uniform(-3, 2, n) * scale
    Z = np.exp(rng.uniform(np.log(4), np.log(120), n)) * scale
    uL = k[0, 0] * X / Z + k[0, 2]
    vL = k[1, 1] * Y / Z + k[1, 2]
    uR = k[0, 0] * (X + tr[0]) / Z + k[0, 2]
    delta = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-9, -2, n)
    ang = rng.uniform(0, 2 * np.pi, n)
    r = thresh * (1.0 + delta)
    right = rng.random(n) < 0.5            # which camera carries the near-threshold residual
    uL32 = (uL + np.where(right, 0.3, r) * np.cos(ang)).astype(np.float32)
    vL32 = (vL + np.where(right, 0.3, r) * np.sin(ang)).astype(np.float32)
    uR32 = (uR + np.where(right, r, 0.3) * np.cos(ang)).astype(np.float32)
    vR32 = (vL + np.where(right, r, 0.3) * np.sin(ang)).astype(np.float32)
    return X, Y, Z, uL32, vL32, uR32, vR32


def test_score_guard_bands_adversarial(device):
    k, m1, m2 = read_cameras()
    tr = m2[:, 3]
    rng = np.random.default_rng(77)
    eye = np.hstack([np.eye(3), np.zeros((3, 1))])
    for thresh, scale, n in ((2.0, 1.0, 901), (2.0, 1e3, 400), (0.7, 1e-3, 333), (5.0, 1.0, 2)):
        pts = _band_case(rng, n, k, tr * scale, thresh, scale)
        # tight cloud of poses around the identity (most residuals stay within the sweep), a few
        # wild ones, and non-finite / degenerate poses that must take the exact path
        Rt = seeded_hypotheses(eye, 300, seed=3)
        Rt[:200] = eye + (Rt[:200] - eye) * 1e-3
        Rt[:, :, 3] *= scale
        Rt[0] = eye
        Rt[290] = 0.0
        Rt[291, 1, 2] = np.nan
        Rt[292, 0, 3] = np.inf
        Rt[293] *= 1e200
        Rt[294] *= 1e-200
        Rt[295, 2, :] = [0.0, 0.0, -1.0, 50.0 * scale]     # depths through zero inside the cloud
        with np.errstate(all="ignore"):
            rc = oracle.score_hypotheses(Rt, k, tr * scale, *pts, thresh, threads=8)
        counts = _gpu_score(device, Rt, k, tr * scale, pts, thresh, want_masks=False)
        assert np.array_equal(counts.cpu().numpy()[0], rc), (thresh, scale)
        assert 0 < rc[0] < n or n < 8      # the sweep really straddles the threshold
    # a principal point far outside the image switches the fp32 stage off; counts stay exact
    k3 = k.copy()
    k3[0, 2] = 3.0e6
    pts = _band_case(rng, 257, k3, tr, 2.0)
    Rt = seeded_hypotheses(eye, 70, seed=5)
    rc = oracle.score_hypotheses(Rt, k3, tr, *pts, 2.0, threads=8)
    counts = _gpu_score(device, Rt, k3, tr, pts, 2.0, want_masks=False)
    assert np.array_equal(counts.cpu().numpy()[0], rc)


@pytest.mark.parametrize("D,stride", [(61, 64), (64, 64), (32, 32), (29, 32)])
def test_knn_ties_across_tile_and_chunk_boundaries(device, D, stride):
    """Exact duplicates of the queries planted on both sides of every internal boundary of the
    matrix-core kernel's scan (32-row tiles, 64-row super-tiles, 1024-row key chunks), with and
    without the row index riding in the descriptor's pad slots: equal distances must resolve to
    the lower train index, and the runner-up must be the next copy."""
    rng = np.random.default_rng(31 + D)
    nt = 2300
    t = _rand_desc(rng, nt, D)
    q = _rand_desc(rng, 90, D)
    spots = [(31, 32), (63, 64), (1023, 1024), (1022, 1025), (2047, 2048), (0, nt - 1), (1026, 2050), (95, 96),
             (1055, 1056), (5, 2298)]
    for i, (a, b) in enumerate(spots):
        t[a] = q[i]
        t[b] = q[i]
    # triples: the winner and the runner-up both come before a third copy in a later chunk
    t[700] = q[40]; t[1500] = q[40]; t[2200] = q[40]
    # a near copy (distance 1) ahead of an exact copy behind a chunk boundary
    near = q[41].copy(); near[0] ^= 1
    t[1000] = near; t[1030] = q[41]
    idx, dist, _ = gpu_match([(q, t)], D, stride, 2, device)
    ridx, rdist = oracle_match([(q, t)], D, 2)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    for i, (a, b) in enumerate(spots):
        assert idx[i].tolist() == [a, b] and dist[i].tolist() == [0, 0]
    assert idx[40].tolist() == [700, 1500] and idx[41].tolist() == [1030, 1000] and dist[41].tolist() == [0, 1]


def test_knn_long_train_set_crosses_key_chunks(device):
    # more than 8192 train rows: the matrix-core kernel scans them chunk by chunk;
    # duplicated rows put exact ties on both sides of chunk boundaries
    rng = np.random.default_rng(21)
    D = 61
    base = _rand_desc(rng, 300, D)
    t = np.concatenate([_rand_desc(rng, 8000, D), base[:150], base[:150], _rand_desc(rng, 700, D), base])
    q = np.concatenate([base, _rand_desc(rng, 40, D)])
    idx, dist, _ = gpu_match([(q, t)], D, 64, 2, device)
    ridx, rdist = oracle_match([(q, t)], D, 2)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert (idx[:150, 0] == 8000 + np.arange(150)).all() and (idx[:150, 1] == 8150 + np.arange(150)).all()
