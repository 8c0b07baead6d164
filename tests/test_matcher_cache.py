"""The matcher's cache and its serialised form (models/Matcher.py:174-176,226-274), and the
ingestion of detector output into the batched tables."""
import pickle

import numpy as np
import pytest
import torch

from oracle import cpu as oracle
from slam_amd import pipeline
from slam_amd.constants import CONSECUTIVE, FILTERED_MATCHES, FRAMES, LEFT, MATCHES, RIGHT
from slam_amd.matcher import Matcher, PrecomputedFeatures
from slam_amd.synthetic import SyntheticSequence, read_cameras
from slam_amd.types import DMatch, KeyPoint


def _three_frame_cache(gpu):
    seq = SyntheticSequence(120, 3, seed=12)
    m = Matcher(algo=PrecomputedFeatures(seq.frames))
    for f in range(3):
        m.detect_and_compute(f)
        if gpu:
            m.find_matching_features(with_significance_test=False)
        else:                                   # no GPU here: a hand-made k = 1 match list stands in
            ms = [[DMatch(i, (7 * i) % 120, float(i % 50))] for i in range(120)]
            m.cache[f][MATCHES] = m.cache[f][FILTERED_MATCHES] = ms
        m.cache[f][CONSECUTIVE] = [[DMatch(1, 2, 3.0)]]
    return seq, m


def _check_schema(cache):
    assert sorted(cache) == [0, 1, 2]
    for f, entry in cache.items():
        assert set(entry) == {FRAMES, LEFT, RIGHT, MATCHES, FILTERED_MATCHES, CONSECUTIVE}
        assert entry[CONSECUTIVE] is None                                         # models/Matcher.py:246
        for side in (LEFT, RIGHT):
            kps, dsc = entry[side]
            assert isinstance(kps, list) and len(kps) == 120 and dsc.shape == (120, 61) and dsc.dtype == np.uint8
            assert set(kps[0]) == {"pt", "size", "angle", "response", "octave", "class_id"}   # models/Matcher.py:262-270
            assert isinstance(kps[0]["pt"], tuple) and len(kps[0]["pt"]) == 2
        for key in (MATCHES, FILTERED_MATCHES):
            assert len(entry[key]) == 120
            assert set(entry[key][0]) == {"queryIdx", "trainIdx", "distance"}          # models/Matcher.py:273-274
            assert isinstance(entry[key][0]["queryIdx"], int) and isinstance(entry[key][0]["distance"], float)


def test_cache_serialisation_schema_and_round_trip():
    seq, m = _three_frame_cache(gpu=False)
    before = {f: ([kp.pt for kp in m.cache[f][LEFT][0]], [(x[0].queryIdx, x[0].trainIdx, x[0].distance)
                                                           for x in m.cache[f][MATCHES]]) for f in range(3)}
    cache = m.get_matcher_cache()
    assert cache is m.cache
    _check_schema(cache)
    assert m.get_matcher_cache() is cache                                         # a second call leaves the dict form alone
    blob = pickle.dumps(cache)
    m2 = Matcher(algo=PrecomputedFeatures(seq.frames))
    m2._packed[0] = {LEFT: "stale", RIGHT: "stale"}                                # device copies of an earlier cache
    m2._stereo_keep[0] = "stale"
    m2._synced_frame = 0
    m2.set_cache(pickle.loads(blob))
    assert m2._packed == {} and m2._stereo_keep == {} and m2._synced_frame is None
    for f in range(3):
        kpl, kpr = m2.get_kp(f)
        assert isinstance(kpl[0], KeyPoint) and [kp.pt for kp in kpl] == before[f][0]
        assert m2.get_feature_location_frame(f, 5, LEFT) == before[f][0][5]
        got = [(x[0].queryIdx, x[0].trainIdx, x[0].distance) for x in m2.get_matches(f)]
        assert got == before[f][1] and isinstance(m2.get_matches(f)[0][0], DMatch)
        assert np.array_equal(m2.cache[f][LEFT][1], seq.frames[f].descL)
    # the track store carries the same dict along (models/TrackDatabase.py:300-304)
    from slam_amd.track_database import TrackDatabase
    db = TrackDatabase()
    db.set_matcher(cache)
    assert db.get_matcher_cache() is cache


def test_apply_thresholds_on_float_distances_needs_no_gpu():
    """An L2 matcher's distances are floats: the ratio test is the reference's own compare
    (models/Matcher.py:178-184), whatever the hardware."""
    ms = [[DMatch(0, 3, 10.5), DMatch(0, 4, 20.0)], [DMatch(1, 5, 12.1), DMatch(1, 6, 20.0)],
          [DMatch(2, 7, 12.0), DMatch(2, 8, 20.0)]]
    kept = Matcher.apply_thresholds(ms, 0.6)
    assert [k[0].queryIdx for k in kept] == [0]                                  # 12.0 < 0.6 * 20.0 is false (strict)
    assert Matcher.apply_thresholds(ms, 0.61)[-1][0].queryIdx == 2
    with pytest.raises(ValueError):
        Matcher.apply_thresholds([[DMatch(0, 1, 2.5)]], 0.6)                       # `for m, n in` on a one-element list


@pytest.mark.gpu
def test_cache_round_trip_then_match_and_stale_tables(device):
    seq, m = _three_frame_cache(gpu=True)
    want = m.match_between_consecutive_frames(0, 1, thresh=0.8)
    stereo = [(x[0].trainIdx, x[0].distance) for x in m.get_matches(2)]
    m2 = Matcher(algo=PrecomputedFeatures(seq.frames))
    m2.set_cache(pickle.loads(pickle.dumps(m.get_matcher_cache())))
    got = m2.match_between_consecutive_frames(0, 1, thresh=0.8)
    assert [(x[0].queryIdx, x[0].trainIdx, x[0].distance) for x in got] == \
        [(x[0].queryIdx, x[0].trainIdx, x[0].distance) for x in want] and len(got) > 30
    assert [(x[0].trainIdx, x[0].distance) for x in m2.get_matches(2)] == stereo
    # adopting ANOTHER cache under the same frame ids must not match the old descriptors
    other = SyntheticSequence(120, 3, seed=77)
    m3 = Matcher(algo=PrecomputedFeatures(other.frames))
    for f in range(3):
        m3.detect_and_compute(f)
        m3.find_matching_features()
    m2.set_cache(m3.get_matcher_cache())
    got = m2.match_between_consecutive_frames(0, 1, thresh=0.8)
    idx, dist = oracle.hamming_knn(other.frames[0].descL, other.frames[1].descL, 2)
    keep = oracle.ratio_filter(idx, dist, 0.8)
    assert [(x[0].queryIdx, x[0].trainIdx) for x in got] == [(int(i), int(idx[i, 0])) for i in np.flatnonzero(keep)]


@pytest.mark.gpu
def test_l2_find_matching_features_with_significance_test(device):
    """The reference's default pair -- float descriptors, cv2.BFMatcher() -- through
    find_matching_features(True) (models/Matcher.py:126-129): the ratio test on float distances."""
    rng = np.random.default_rng(3)
    n, dim = 150, 128
    base = rng.uniform(0, 255, (n, dim)).astype(np.float32) + 0.25                # non-integer: the fp32 kernel
    noisy = (base + rng.normal(0, 6, base.shape)).astype(np.float32)[rng.permutation(n)]
    kp = rng.uniform(0, 370, (n, 2)).astype(np.float32)

    class F:
        kpL, kpR, descL, descR = kp, kp, base, noisy
    m = Matcher(algo=PrecomputedFeatures([F]), threshold=0.6)
    m.detect_and_compute(0)
    m.find_matching_features(with_significance_test=True)
    idx, dist = oracle.l2_knn(base, noisy, 2)
    want = [i for i in range(n) if float(dist[i, 0]) < 0.6 * float(dist[i, 1])]
    got = m.get_filtered_matches(0)
    assert [g[0].queryIdx for g in got] == want and len(want) > 100
    assert [g[0].trainIdx for g in got] == [int(idx[i, 0]) for i in want]
    assert all(abs(g[0].distance - float(dist[i, 0])) <= 1e-5 * float(dist[i, 0]) for g, i in zip(got, want))
    assert len(m.get_matches(0)) == n and len(m.get_matches(0)[0]) == 2
    # a plugged matcher without the array interface: distances stay floats, same outcome

    class Plain:
        def __init__(self, inner):
            self.inner = inner

        def knnMatch(self, q, t, k=2):
            return self.inner.knnMatch(q, t, k)
    m2 = Matcher(algo=PrecomputedFeatures([F]), matcher=Plain(m.matcher), threshold=0.6)
    m2.detect_and_compute(0)
    m2.find_matching_features(with_significance_test=True)
    assert [g[0].queryIdx for g in m2.get_filtered_matches(0)] == want
    m2.cache[1] = m2.cache[0]
    assert [g[0].queryIdx for g in m2.match_between_consecutive_frames(0, 1, thresh=0.6)] == \
        [i for i in range(n) if float(oracle.l2_knn(base, base, 2)[1][i, 0]) < 0.6 * float(oracle.l2_knn(base, base, 2)[1][i, 1])]


@pytest.mark.gpu
@pytest.mark.parametrize("D", [32, 61])
def test_detector_output_ingestion(device, D):
    """ORB-32 / AKAZE-61 detector output (lists of keypoint objects + uint8 descriptor arrays, ragged,
    one image without keypoints) -> FrameTables -> stereo matching equal to the oracle's."""
    rng = np.random.default_rng(D)
    per_frame, raw = [], []
    for f, (nl, nr) in enumerate([(140, 131), (90, 200), (0, 60), (75, 75)]):
        dl = rng.integers(0, 256, (nl, D), dtype=np.uint8)
        dr = rng.integers(0, 256, (nr, D), dtype=np.uint8)
        m = min(nl, nr) // 2
        dr[:m] = dl[:m] ^ (rng.random((m, D)) < 0.02).astype(np.uint8)            # true partners
        xl = rng.uniform(0, 370, (nl, 2)).astype(np.float32)
        xr = rng.uniform(0, 370, (nr, 2)).astype(np.float32)
        xr[:m, 1] = xl[:m, 1] + rng.uniform(-1.5, 1.5, m).astype(np.float32)
        kl = [KeyPoint(x, y) for x, y in xl]
        kr = xr                                                                    # an [N,2] array is accepted too
        per_frame.append(((kl, dl if nl else None), (kr, dr)))                    # cv2 returns None without keypoints
        raw.append((xl, dl, xr, dr))
    tb = pipeline.FrameTables.from_detector_output(per_frame, device)
    assert tb.n_frames == 4 and tb.desc_bytes == D and tb.descL.shape[1] == (32 if D == 32 else 64)
    assert np.diff(tb.offL).tolist() == [140, 90, 0, 75] and np.diff(tb.offR).tolist() == [131, 200, 60, 75]
    k, m1, m2 = read_cameras()
    fe = pipeline.FrontEnd(k, m1, m2)
    ws = fe.prepare(tb, n_hyp=0)
    fe.run(ws)
    torch.cuda.synchronize()
    s_idx, s_keep = ws.s_idx.cpu().numpy(), ws.s_keep.cpu().numpy()
    for f, (xl, dl, xr, dr) in enumerate(raw):
        s = slice(int(tb.offL[f]), int(tb.offL[f + 1]))
        if dl.shape[0] == 0:
            continue
        idx, _ = oracle.hamming_knn(dl, dr, 1)
        assert np.array_equal(s_idx[s, 0], idx[:, 0])
        assert np.array_equal(s_keep[s], oracle.row_filter(idx, xl[:, 1], xr[:, 1], 2.0))
    # the same tables from a matcher's cache, live and serialised
    seq = SyntheticSequence(80, 3, seed=5, desc_bytes=D)
    mt = Matcher(algo=PrecomputedFeatures(seq.frames))
    for f in range(3):
        mt.detect_and_compute(f)
    a = pipeline.FrameTables.from_matcher_cache(mt.cache, [0, 1, 2], device)
    b = pipeline.FrameTables.from_matcher_cache(pickle.loads(pickle.dumps(mt.get_matcher_cache())), [0, 1, 2], device)
    c = pipeline.FrameTables.from_frames(seq.frames, device)
    for name in ("descL", "descR", "xL", "yL", "xR", "yR"):
        assert torch.equal(getattr(a, name), getattr(c, name)) and torch.equal(getattr(b, name), getattr(c, name))
