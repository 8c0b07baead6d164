"""KITTI-shaped synthetic stereo sequences (host side, numpy only).

The reference runs on KITTI odometry images through a cv2 detector
(models/Matcher.py:111-119); neither the dataset nor cv2 travels with this
package, so tests and benchmarks feed the front-end with synthetic keypoints and
binary descriptors of the same shape:

* camera: fx = fy = 707.0912, cx = 601.8873, cy = 183.1104, right camera
  P1[0,3] = -379.8145 (VAN_ex/dataset/sequences/05/calib.txt:1-2), image 1226x370;
* per frame 0.6*N keypoints are projections of persistent 3-D points
  (X in [-25,25], Y in [-3,2], Z log-uniform in [4,120] m when spawned) with
  N(0, 0.3 px) noise per coordinate and view, stored as float32; the other
  0.4*N are distractors with random positions;
* descriptors are 61 bytes (486 used bits, the top two bits of byte 60 are zero):
  a random base per 3-D point, each view flips bits i.i.d. with p = 0.08;
  distractors get independent random descriptors; rows are shuffled per image;
* motion: 0.8 m forward plus N(0, 0.5 deg) yaw per frame.

Frame f is generated from numpy.random.default_rng(seed + f) given the world
state, so a sequence is reproducible from (seed, n_kp, n_frames).
"""
import numpy as np

IMG_W, IMG_H = 1226, 370
FX = 707.0912
CX, CY = 601.8873, 183.1104
P1_TX = -379.8145
SEED = 6


def read_cameras():
    """K, M1, M2 exactly as utils/utils.py:75-86 derives them from calib.txt
    (K^-1 @ P0 and K^-1 @ P1, in numpy float64)."""
    m1 = np.array([[FX, 0.0, CX, 0.0], [0.0, FX, CY, 0.0], [0.0, 0.0, 1.0, 0.0]])
    m2 = m1.copy()
    m2[0, 3] = P1_TX
    k = m1[:, :3]
    return k, np.linalg.inv(k) @ m1, np.linalg.inv(k) @ m2


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def rodrigues(rv):
    """Rotation vector -> matrix."""
    th = float(np.linalg.norm(rv))
    if th < 1e-300:
        return np.eye(3)
    k = np.asarray(rv, dtype=np.float64) / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


class Frame:
    """One stereo frame: float32 keypoints [N,2], uint8 descriptors [N,D] per image,
    ground-truth 3-D point id per keypoint (-1 for distractors) and world->camera pose."""

    __slots__ = ("kpL", "kpR", "descL", "descR", "idL", "idR", "Rt")

    def __init__(self, kpL, kpR, descL, descR, idL, idR, Rt):
        self.kpL, self.kpR, self.descL, self.descR = kpL, kpR, descL, descR
        self.idL, self.idR, self.Rt = idL, idR, Rt


class SyntheticSequence:
    def __init__(self, n_kp, n_frames, seed=SEED, desc_bytes=61, flip=0.08, true_frac=0.6, noise=0.3,
                 step=0.8, yaw_sigma_deg=0.5, keep_world=False):
        self.n_kp, self.n_frames, self.seed = int(n_kp), int(n_frames), int(seed)
        self.desc_bytes, self.flip, self.true_frac, self.noise = int(desc_bytes), flip, true_frac, noise
        self.step, self.yaw_sigma = step, np.deg2rad(yaw_sigma_deg)
        self.k, self.m1, self.m2 = read_cameras()
        self.t_right = self.m2[:, 3].copy()
        self.used_bits = 486 if desc_bytes == 61 else 8 * desc_bytes
        self._tail_mask = np.uint8(0xFF >> (8 * desc_bytes - self.used_bits))
        self.frames = []
        self._world = [] if keep_world else None      # every 3-D point ever spawned, by id (world_points)
        self._generate()

    # -- helpers -----------------------------------------------------------
    def _random_desc(self, rng, n):
        d = rng.integers(0, 256, size=(n, self.desc_bytes), dtype=np.uint8)
        d[:, -1] &= self._tail_mask
        return d

    def _flip_mask(self, rng, n):
        bits = rng.random((n, 8 * self.desc_bytes)) < self.flip
        m = np.packbits(bits, axis=1, bitorder="little")
        m[:, -1] &= self._tail_mask
        return m

    def _project(self, Xc):
        z = Xc[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            uL = FX * Xc[:, 0] / z + CX
            v = FX * Xc[:, 1] / z + CY
            uR = FX * (Xc[:, 0] + self.t_right[0]) / z + CX
        ok = (z >= 4.0) & (z <= 120.0) & (uL >= 0) & (uL < IMG_W) & (uR >= 0) & (uR < IMG_W) & (v >= 0) & (v < IMG_H)
        return uL, uR, v, ok

    def _spawn(self, rng, n):
        """n new points in the current camera frame, all visible in both images."""
        out = np.zeros((0, 3))
        while out.shape[0] < n:
            m = 2 * (n - out.shape[0]) + 16
            X = rng.uniform(-25.0, 25.0, m)
            Y = rng.uniform(-3.0, 2.0, m)
            Z = np.exp(rng.uniform(np.log(4.0), np.log(120.0), m))
            P = np.stack([X, Y, Z], axis=1)
            ok = self._project(P)[3]
            out = np.concatenate([out, P[ok]], axis=0)
        return out[:n]

    # -- generation --------------------------------------------------------
    def _generate(self):
        n_true = int(round(self.true_frac * self.n_kp))
        n_dis = self.n_kp - n_true
        R, t = np.eye(3), np.zeros(3)          # world -> camera of the current frame
        Pw = np.zeros((0, 3))
        base = np.zeros((0, self.desc_bytes), dtype=np.uint8)
        prio = np.zeros(0)
        ids = np.zeros(0, dtype=np.int64)
        next_id = 0
        for f in range(self.n_frames):
            rng = np.random.default_rng(self.seed + f)
            if f > 0:
                yaw = rng.normal(0.0, self.yaw_sigma)
                Rrel = _rot_y(-yaw)
                c = np.array([0.0, 0.0, self.step])
                R, t = Rrel @ R, Rrel @ (t - c)
            Xc = Pw @ R.T + t
            uL, uR, v, ok = self._project(Xc)
            alive = Xc[:, 2] >= 4.0 if Xc.size else np.zeros(0, dtype=bool)
            # forget what fell behind the camera for good
            Pw, base, prio, ids = Pw[alive], base[alive], prio[alive], ids[alive]
            Xc, uL, uR, v, ok = Xc[alive], uL[alive], uR[alive], v[alive], ok[alive]
            vis = np.flatnonzero(ok)
            if vis.size > n_true:
                vis = vis[np.argsort(prio[vis], kind="stable")[:n_true]]
            need = n_true - vis.size
            if need > 0:
                newc = self._spawn(rng, need)
                neww = (newc - t) @ R          # R^T (Xc - t)
                nb = self._random_desc(rng, need)
                start = Pw.shape[0]
                Pw = np.concatenate([Pw, neww], axis=0)
                base = np.concatenate([base, nb], axis=0)
                prio = np.concatenate([prio, rng.random(need)])
                ids = np.concatenate([ids, np.arange(next_id, next_id + need)])
                next_id += need
                if self._world is not None:
                    self._world.append(neww)
                Xc = np.concatenate([Xc, newc], axis=0)
                nuL, nuR, nv, _ = self._project(newc)
                uL, uR, v = np.concatenate([uL, nuL]), np.concatenate([uR, nuR]), np.concatenate([v, nv])
                vis = np.concatenate([vis, np.arange(start, start + need)])
            # observations of the true points
            nz = rng.normal(0.0, self.noise, size=(4, vis.size))
            kpL_t = np.stack([uL[vis] + nz[0], v[vis] + nz[1]], axis=1)
            kpR_t = np.stack([uR[vis] + nz[2], v[vis] + nz[3]], axis=1)
            dL_t = base[vis] ^ self._flip_mask(rng, vis.size)
            dR_t = base[vis] ^ self._flip_mask(rng, vis.size)
            # distractors
            kpL_d = np.stack([rng.uniform(0, IMG_W, n_dis), rng.uniform(0, IMG_H, n_dis)], axis=1)
            kpR_d = np.stack([rng.uniform(0, IMG_W, n_dis), rng.uniform(0, IMG_H, n_dis)], axis=1)
            dL_d, dR_d = self._random_desc(rng, n_dis), self._random_desc(rng, n_dis)
            pL, pR = rng.permutation(self.n_kp), rng.permutation(self.n_kp)
            idt = ids[vis]
            idd = np.full(n_dis, -1, dtype=np.int64)
            kpL = np.concatenate([kpL_t, kpL_d])[pL].astype(np.float32)
            kpR = np.concatenate([kpR_t, kpR_d])[pR].astype(np.float32)
            descL = np.ascontiguousarray(np.concatenate([dL_t, dL_d])[pL])
            descR = np.ascontiguousarray(np.concatenate([dR_t, dR_d])[pR])
            idL = np.concatenate([idt, idd])[pL]
            idR = np.concatenate([idt, idd])[pR]
            self.frames.append(Frame(kpL, kpR, descL, descR, idL, idR, np.hstack([R, t[:, None]])))

    # -- ground truth ------------------------------------------------------
    def world_points(self, ids):
        """[n,3] world coordinates of the 3-D points with these ids (Frame.idL / idR >= 0); needs keep_world=True."""
        if self._world is None:
            raise ValueError("SyntheticSequence(..., keep_world=True) keeps the points")
        if not isinstance(self._world, np.ndarray):
            self._world = np.concatenate(self._world, axis=0) if self._world else np.zeros((0, 3))
        return self._world[np.asarray(ids, dtype=np.int64)]

    def relative_pose(self, i, j):
        """[3,4] transform taking points from camera i's frame to camera j's."""
        Ri, ti = self.frames[i].Rt[:, :3], self.frames[i].Rt[:, 3]
        Rj, tj = self.frames[j].Rt[:, :3], self.frames[j].Rt[:, 3]
        Rr = Rj @ Ri.T
        return np.hstack([Rr, (tj - Rr @ ti)[:, None]])


def rodrigues_batch(rv):
    """[H,3] rotation vectors -> [H,3,3] matrices."""
    rv = np.asarray(rv, dtype=np.float64)
    th = np.linalg.norm(rv, axis=1)
    safe = np.where(th < 1e-300, 1.0, th)
    k = rv / safe[:, None]
    Kx = np.zeros((rv.shape[0], 3, 3))
    Kx[:, 0, 1], Kx[:, 0, 2] = -k[:, 2], k[:, 1]
    Kx[:, 1, 0], Kx[:, 1, 2] = k[:, 2], -k[:, 0]
    Kx[:, 2, 0], Kx[:, 2, 1] = -k[:, 1], k[:, 0]
    s, c = np.sin(th)[:, None, None], (1 - np.cos(th))[:, None, None]
    return np.eye(3)[None] + s * Kx + c * (Kx @ Kx)


def seeded_hypotheses(Rt_gt, n_hyp, seed=SEED):
    """A fixed, seeded hypothesis set around a ground-truth relative pose:
    rotation-vector noise sigma in {1e-3, 1e-2, 5e-2} rad and translation noise
    sigma in {0.01, 0.1, 0.5} m, cycled over the hypotheses.  Returns [H,3,4] f64."""
    rng = np.random.default_rng(seed)
    sr = np.array([1e-3, 1e-2, 5e-2])
    st = np.array([0.01, 0.1, 0.5])
    h = np.arange(n_hyp)
    rv = rng.normal(size=(n_hyp, 3)) * sr[h % 3][:, None]
    tv = rng.normal(size=(n_hyp, 3)) * st[(h // 3) % 3][:, None]
    out = np.empty((n_hyp, 3, 4))
    out[:, :, :3] = rodrigues_batch(rv) @ Rt_gt[:, :3]
    out[:, :, 3] = Rt_gt[:, 3] + tv
    return out
