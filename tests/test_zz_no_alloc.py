"""The scoring entry point that allocates nothing (slam_score_hypotheses_static): its counts
equal those of slam_score_hypotheses with a caller-owned workspace, bit for bit, on two calls
in a row (the second smaller, so stale workspace contents of the first would show), and equal
the CPU oracle's; sizes beyond its built-in workspace are refused."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import cpu as oracle
from slam_amd import _lib
from slam_amd.synthetic import read_cameras

STATIC_MAX_POINTS = 65536   # SLAM_SCORE_STATIC_MAX_POINTS in include/slam_amd.h


def _case(rng, n_sets, cap, H, counts_n):
    """Random KITTI-like consensus sets and hypotheses near a true pose."""
    k, _, m2 = read_cameras()
    X = rng.uniform(-25, 25, (n_sets, cap))
    Y = rng.uniform(-3, 2, (n_sets, cap))
    Z = np.exp(rng.uniform(np.log(4), np.log(120), (n_sets, cap)))
    Rt = np.zeros((n_sets, H, 3, 4))
    uL = np.zeros((n_sets, cap), np.float32)
    vL, uR, vR = np.zeros_like(uL), np.zeros_like(uL), np.zeros_like(uL)
    for s in range(n_sets):
        w = rng.normal(0, 0.01, 3)
        th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R = np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx
        t = np.array([0.02, 0.01, -0.8]) + rng.normal(0, 0.02, 3)
        P = np.stack([X[s], Y[s], Z[s]], 1)
        L = P @ R.T + t
        Rr = L + m2[:, 3]
        pl, pr = L @ k.T, Rr @ k.T
        noise = rng.normal(0, 0.7, (cap, 4))
        uL[s], vL[s] = pl[:, 0] / pl[:, 2] + noise[:, 0], pl[:, 1] / pl[:, 2] + noise[:, 1]
        uR[s], vR[s] = pr[:, 0] / pr[:, 2] + noise[:, 2], pr[:, 1] / pr[:, 2] + noise[:, 3]
        for h in range(H):
            dw = rng.normal(0, 2e-3, 3)
            Kd = np.array([[0, -dw[2], dw[1]], [dw[2], 0, -dw[0]], [-dw[1], dw[0], 0]])
            Rt[s, h, :, :3] = (np.eye(3) + Kd) @ R
            Rt[s, h, :, 3] = t + rng.normal(0, 0.03, 3)
    n = np.asarray(counts_n, dtype=np.int32)
    return k, m2[:, 3].copy(), Rt, X, Y, Z, uL, vL, uR, vR, n


def _run(lib, dev, case, cap, static):
    k, tr, Rt, X, Y, Z, uL, vL, uR, vR, n = case
    n_sets, H = Rt.shape[:2]
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    dRt = t(Rt.reshape(n_sets, H, 12), torch.float64)
    d = [t(a.reshape(-1), torch.float64) for a in (X, Y, Z)] + [t(a.reshape(-1), torch.float32) for a in (uL, vL, uR, vR)]
    dn = t(n, torch.int32)
    counts = torch.full((n_sets, H), -7, dtype=torch.int32, device=dev)
    K9 = (ctypes.c_double * 9)(*k.reshape(-1))
    t3 = (ctypes.c_double * 3)(*tr)
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if static:
        st = lib.slam_score_hypotheses_static(p(dRt), n_sets, H, K9, t3, *[p(x) for x in d], cap, p(dn), 0, 2.0,
                                              p(counts), stream)
    else:
        work = torch.empty(16 * n_sets * max(cap, 1), dtype=torch.float64, device=dev)
        st = lib.slam_score_hypotheses(p(dRt), n_sets, H, K9, t3, *[p(x) for x in d], cap, p(dn), 0, 2.0, p(counts),
                                       None, 0, p(work), stream)
    _lib.check(st, "score")
    torch.cuda.synchronize()
    return counts.cpu().numpy()


@pytest.mark.gpu
def test_static_workspace_scoring_matches_the_workspace_entry_point(device):
    lib = _lib.load()
    rng = np.random.default_rng(77)
    # first call: 24 sets x 700 slots, ragged fills (incl. an empty and a full set); second: smaller
    big = _case(rng, 24, 700, 384, [700, 0, 3, 64, 65, 699] + list(rng.integers(100, 700, 18)))
    small = _case(rng, 5, 130, 70, [130, 1, 77, 128, 0])
    for case, cap in ((big, 700), (small, 130)):
        want = _run(lib, device, case, cap, static=False)
        got = _run(lib, device, case, cap, static=True)
        assert np.array_equal(got, want)
        k, tr, Rt, X, Y, Z, uL, vL, uR, vR, n = case
        for s in (0, 2, len(n) - 1):
            m = int(n[s])
            ref = oracle.score_hypotheses(Rt[s], k, tr, X[s, :m], Y[s, :m], Z[s, :m], uL[s, :m], vL[s, :m], uR[s, :m],
                                          vR[s, :m], 2.0)
            assert np.array_equal(got[s], ref)
    # and once more the big one after the small one
    assert np.array_equal(_run(lib, device, big, 700, static=True), _run(lib, device, big, 700, static=False))


def test_static_workspace_scoring_refuses_what_does_not_fit():
    """No GPU needed: the entry point checks its sizes before its first HIP call.  The library itself
    still has to load, which needs the HIP runtime's shared objects on the machine."""
    try:
        lib = _lib.load()
    except (OSError, _lib.SlamError) as exc:
        pytest.skip(f"libslam_amd.so does not load here: {exc}")
    null = ctypes.c_void_p(0)
    K9 = (ctypes.c_double * 9)(700.0, 0.0, 600.0, 0.0, 700.0, 180.0, 0.0, 0.0, 1.0)
    t3 = (ctypes.c_double * 3)(-0.5, 0.0, 0.0)
    args = lambda n_sets, cap: [null, n_sets, 8, K9, t3] + [null] * 7 + [cap, null, cap, 2.0, null, null]
    assert lib.slam_score_hypotheses_static(*args(STATIC_MAX_POINTS // 1000 + 1, 1000)) == -3
    assert b"exceed" in lib.slam_last_error()
    assert lib.slam_score_hypotheses_static(*args(-1, 10)) == -1
    assert lib.slam_score_hypotheses_static(*args(0, 10)) == 0
