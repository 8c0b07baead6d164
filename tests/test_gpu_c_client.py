"""The C ABI used by a program that has neither Python nor PyTorch in it: examples/stereo_pair
(plain C++, hipMalloc / hipMemcpy, built by slam_amd/csrc/Makefile) does the stereo matching and
triangulation of one frame; its output file is compared with the CPU oracle bit for bit.  The
process started here loads nothing but the shared library and the HIP runtime."""
import os
import subprocess

import numpy as np
import pytest

from oracle import cpu as oracle
from slam_amd.synthetic import SyntheticSequence, read_cameras

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIENT = os.path.join(ROOT, "examples", "stereo_pair")


def _run(tmp_path, descL, descR, kpL, kpR, P, Q, thr):
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    nL, nR, D = len(descL), len(descR), descL.shape[1]
    with open(src, "wb") as f:
        f.write(np.array([nL, nR, D, 0], np.int32).tobytes())
        for a, dt in ((descL, np.uint8), (descR, np.uint8), (kpL, np.float32), (kpR, np.float32), (P, np.float64),
                      (Q, np.float64), (np.array([thr]), np.float64)):
            f.write(np.ascontiguousarray(a, dtype=dt).tobytes())
    r = subprocess.run([CLIENT, str(src), str(dst)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    raw = open(dst, "rb").read()
    out, off = {}, 0
    for name, dt, shape in (("idx", np.int32, (nL, 2)), ("dist", np.int32, (nL, 2)), ("keep", np.uint8, (nL,)),
                            ("X", np.float64, (nL,)), ("Y", np.float64, (nL,)), ("Z", np.float64, (nL,)),
                            ("valid", np.uint8, (nL,)), ("map", np.int32, (nL,))):
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        out[name] = np.frombuffer(raw[off:off + n], dtype=dt).reshape(shape)
        off += n
    assert off == len(raw)
    return out, r.stdout


@pytest.mark.parametrize("n_kp,desc_bytes", [(700, 61), (333, 32), (1, 61)])
def test_c_client_matches_the_oracle(tmp_path, n_kp, desc_bytes):
    assert os.path.exists(CLIENT), "examples/stereo_pair is not built (make -C slam_amd/csrc)"
    k, m1, m2 = read_cameras()
    P, Q = k @ m1, k @ m2
    fr = SyntheticSequence(n_kp, 1, seed=3 + n_kp).frames[0]
    descL, descR = fr.descL[:, :desc_bytes], fr.descR[:, :desc_bytes]
    got, text = _run(tmp_path, descL, descR, fr.kpL, fr.kpR, P, Q, 2.0)
    idx, dist = oracle.hamming_knn(descL, descR, 1)
    keep = oracle.row_filter(idx, fr.kpL[:, 1], fr.kpR[:, 1], 2.0)
    assert np.array_equal(got["idx"][:, 0], idx[:, 0]) and np.array_equal(got["dist"][:, 0], dist[:, 0])
    assert (got["idx"][:, 1] == -1).all() and (got["dist"][:, 1] == np.iinfo(np.int32).max).all()   # k = 1: no second neighbour
    assert np.array_equal(got["keep"], keep)
    j = np.where(keep > 0, idx[:, 0], 0)
    X, Y, Z, valid = oracle.triangulate_dlt(P, Q, fr.kpL[:, 0], fr.kpL[:, 1], fr.kpR[j, 0], fr.kpR[j, 1])
    kept = keep > 0
    assert np.array_equal(got["map"], np.where(kept, idx[:, 0], -1))
    assert np.array_equal(got["valid"], (valid * kept).astype(np.uint8))
    for name, want in (("X", X), ("Y", Y), ("Z", Z)):
        assert np.array_equal(got[name][kept], want[kept]), name          # bit for bit
        assert np.isnan(got[name][~kept]).all()
    assert f"matches kept {int(kept.sum())}," in text


def test_c_client_with_an_empty_right_image(tmp_path):
    """No right keypoints at all: every left row reports no neighbour, nothing is kept, nothing triangulated."""
    k, m1, m2 = read_cameras()
    fr = SyntheticSequence(50, 1, seed=9).frames[0]
    got, text = _run(tmp_path, fr.descL, fr.descR[:0], fr.kpL, fr.kpR[:0], k @ m1, k @ m2, 2.0)
    assert (got["idx"] == -1).all() and (got["dist"] == np.iinfo(np.int32).max).all()
    assert not got["keep"].any() and not got["valid"].any() and (got["map"] == -1).all()
    assert np.isnan(got["X"]).all() and np.isnan(got["Y"]).all() and np.isnan(got["Z"]).all()
    assert "matches kept 0," in text
