"""Host-side pose solvers used around the GPU scoring kernel.

The reference obtains RANSAC hypotheses from cv2.solvePnP(flags=SOLVEPNP_AP3P) on four
sampled correspondences and refines the winner with SOLVEPNP_ITERATIVE on all supporters
(utils/utils.py:428,452,481-497).  OpenCV is not a dependency of this package, so the
same two jobs are done here in numpy:

* `p3p_4pt`  : Grunert's three-point resection (quartic in the depth ratio) on the
               first three correspondences, the fourth picks among the up-to-four
               solutions -- the role AP3P's fourth point plays in cv2;
* `refine_pnp`: Levenberg-Marquardt on the left-image reprojection error over all
               given correspondences, started from a supplied pose (optionally on both
               images' residuals: not the reference's cost, kept to study its drift).

They are small scalar host code (a few poses per frame); every hypothesis they produce
is scored on the GPU.  Their outputs agree with cv2's to solver tolerance, not bit for
bit: poses are "parity unpinned", the supporter counts computed from a given pose are exact.
"""
import numpy as np

SOLVEPNP_ITERATIVE = 0   # cv2's enum values, so callers may pass either
SOLVEPNP_AP3P = 5


def _rigid_fit(P, Q):
    """R, t with Q ~ R P + t (Kabsch), P,Q [n,3]."""
    cp, cq = P.mean(axis=0), Q.mean(axis=0)
    H = (P - cp).T @ (Q - cq)
    U, _, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    D = np.diag([1.0, 1.0, d])
    R = Vt.T @ D @ U.T
    return R, cq - R @ cp


def p3p_grunert(P, f):
    """P [3,3] world points (rows), f [3,3] unit bearing vectors (rows).
    Returns a list of (R, t) with  lambda_i f_i = R P_i + t,  lambda_i > 0."""
    a = np.linalg.norm(P[1] - P[2])
    b = np.linalg.norm(P[0] - P[2])
    c = np.linalg.norm(P[0] - P[1])
    if min(a, b, c) < 1e-12:
        return []
    ca, cb, cg = f[1] @ f[2], f[0] @ f[2], f[0] @ f[1]
    a2, b2, c2 = a * a, b * b, c * c
    q = (a2 - c2) / b2
    A4 = (q - 1.0) ** 2 - 4.0 * c2 / b2 * ca * ca
    A3 = 4.0 * (q * (1.0 - q) * cb - (1.0 - (a2 + c2) / b2) * ca * cg + 2.0 * c2 / b2 * ca * ca * cb)
    A2 = 2.0 * (q * q - 1.0 + 2.0 * q * q * cb * cb + 2.0 * (b2 - c2) / b2 * ca * ca
                - 4.0 * (a2 + c2) / b2 * ca * cb * cg + 2.0 * (b2 - a2) / b2 * cg * cg)
    A1 = 4.0 * (-q * (1.0 + q) * cb + 2.0 * a2 / b2 * cg * cg * cb - (1.0 - (a2 + c2) / b2) * ca * cg)
    A0 = (1.0 + q) ** 2 - 4.0 * a2 / b2 * cg * cg
    coeffs = np.array([A4, A3, A2, A1, A0])
    if not np.all(np.isfinite(coeffs)) or np.abs(coeffs).max() == 0.0:
        return []
    roots = np.roots(coeffs)
    out = []
    for v in roots:
        if abs(v.imag) > 1e-7 * max(1.0, abs(v.real)):
            continue
        v = v.real
        if v <= 0:
            continue
        den = 2.0 * (cg - v * ca)
        if abs(den) < 1e-14:
            continue
        u = ((q - 1.0) * v * v - 2.0 * q * cb * v + 1.0 + q) / den
        if u <= 0:
            continue
        s1sq = c2 / (1.0 + u * u - 2.0 * u * cg)
        if not (s1sq > 0):
            continue
        s1 = np.sqrt(s1sq)
        Q = np.stack([s1 * f[0], u * s1 * f[1], v * s1 * f[2]])
        R, t = _rigid_fit(P, Q)
        if np.abs(R @ P.T + t[:, None] - Q.T).max() > 1e-6 * max(1.0, np.abs(Q).max()):
            continue
        out.append((R, t))
    return out


def p3p_4pt(points_3d, points_2d, K):
    """Pose from four 3-D/2-D correspondences: P3P on the first three, the fourth
    disambiguates.  Returns [3,4] or None (degenerate sample)."""
    points_3d = np.asarray(points_3d, dtype=np.float64).reshape(-1, 3)
    points_2d = np.asarray(points_2d, dtype=np.float64).reshape(-1, 2)
    if points_3d.shape[0] < 4:
        return None
    K = np.asarray(K, dtype=np.float64)
    Kinv = np.linalg.inv(K)
    rays = (Kinv @ np.hstack([points_2d, np.ones((points_2d.shape[0], 1))]).T).T
    f = rays / np.linalg.norm(rays, axis=1, keepdims=True)
    sols = p3p_grunert(points_3d[:3], f[:3])
    best, best_err = None, np.inf
    for R, t in sols:
        x = K @ (R @ points_3d[3] + t)
        if x[2] <= 0:
            continue
        err = np.linalg.norm(x[:2] / x[2] - points_2d[3])
        if err < best_err:
            best, best_err = np.hstack([R, t[:, None]]), err
    if best is not None:
        # The quartic loses digits on thin triangles; a few Gauss-Newton steps on the three
        # defining points (6 equations, 6 unknowns) restore an exact fit to them.  Most samples
        # fit their three points to rounding already and are left as they are.
        Y3 = points_3d[:3] @ best[:, :3].T + best[:, 3]
        fit = np.abs(np.stack([K[0, 0] * Y3[:, 0] / Y3[:, 2] + K[0, 2], K[1, 1] * Y3[:, 1] / Y3[:, 2] + K[1, 2]],
                              axis=1) - points_2d[:3]).max()
        polished = refine_pnp(points_3d[:3], points_2d[:3], K, best, iterations=4) if not fit < 1e-9 else None
        if polished is not None:
            x = K @ (polished[:, :3] @ points_3d[3] + polished[:, 3])
            if x[2] > 0 and np.linalg.norm(x[:2] / x[2] - points_2d[3]) <= best_err + 1e-6:
                best = polished
    return best


def _exp_so3(w):
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Kx
    return np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / (th * th) * (Kx @ Kx)


def refine_pnp(points_3d, points_2d, K, Rt0, iterations=30, points_2d_right=None, t_right=None):
    """Levenberg-Marquardt on left-image reprojection error from the initial pose Rt0 -- the cost
    cv2.solvePnP(SOLVEPNP_ITERATIVE) minimises for the reference (utils/utils.py:452,481-485: left
    keypoints only).  Option, off by default and not the reference's behaviour: with points_2d_right
    [n,2] and t_right [3] (the right camera's offset, M2[:, 3]) the right image's residuals are
    minimised as well (4 per point)."""
    X = np.asarray(points_3d, dtype=np.float64).reshape(-1, 3)
    uv = np.asarray(points_2d, dtype=np.float64).reshape(-1, 2)
    if X.shape[0] < 3 or Rt0 is None:
        return None
    K = np.asarray(K, dtype=np.float64)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    R, t = np.array(Rt0[:, :3], dtype=np.float64), np.array(Rt0[:, 3], dtype=np.float64)
    stereo = points_2d_right is not None
    if stereo:
        uvr = np.asarray(points_2d_right, dtype=np.float64).reshape(-1, 2)
        tr = np.asarray(t_right, dtype=np.float64).reshape(3)
    cams = [(uv, np.zeros(3))] + ([(uvr, tr)] if stereo else [])

    def residual(R, t):
        Y = X @ R.T + t
        cols = []
        for obs, off in cams:
            Yc = Y + off
            cols += [fx * Yc[:, 0] / Yc[:, 2] + cx - obs[:, 0], fy * Yc[:, 1] / Yc[:, 2] + cy - obs[:, 1]]
        return Y, np.stack(cols, axis=1)

    lam = 1e-3
    Y, r = residual(R, t)
    cost = float((r * r).sum())
    if not np.isfinite(cost):
        return None
    for _ in range(iterations):
        n = X.shape[0]
        J = np.zeros((n, 2 * len(cams), 6))
        # Y' = exp(w) Y + d  ->  dY/dw = -[Y]x, dY/dd = I
        Yx = np.zeros((n, 3, 3))
        Yx[:, 0, 1], Yx[:, 0, 2] = -Y[:, 2], Y[:, 1]
        Yx[:, 1, 0], Yx[:, 1, 2] = Y[:, 2], -Y[:, 0]
        Yx[:, 2, 0], Yx[:, 2, 1] = -Y[:, 1], Y[:, 0]
        for c, (_, off) in enumerate(cams):
            Yc = Y + off
            z = Yc[:, 2]
            # d(proj)/dY
            du = np.stack([fx / z, np.zeros(n), -fx * Yc[:, 0] / (z * z)], axis=1)
            dv = np.stack([np.zeros(n), fy / z, -fy * Yc[:, 1] / (z * z)], axis=1)
            J[:, 2 * c, :3] = -np.einsum("ni,nij->nj", du, Yx)
            J[:, 2 * c + 1, :3] = -np.einsum("ni,nij->nj", dv, Yx)
            J[:, 2 * c, 3:] = du
            J[:, 2 * c + 1, 3:] = dv
        Jm = J.reshape(-1, 6)
        g = Jm.T @ r.reshape(-1)
        Hm = Jm.T @ Jm
        try:
            step = np.linalg.solve(Hm + lam * np.diag(np.diag(Hm) + 1e-12), -g)
        except np.linalg.LinAlgError:
            return None
        if np.abs(step[:3]).max() + np.abs(step[3:]).max() / (1.0 + np.abs(t).max()) < 1e-11:
            break                                  # converged: the step no longer moves the pose
        dR = _exp_so3(step[:3])
        R2, t2 = dR @ R, dR @ t + step[3:]
        Y2, r2 = residual(R2, t2)
        c2 = float((r2 * r2).sum())
        if np.isfinite(c2) and c2 < cost:
            R, t, Y, r = R2, t2, Y2, r2
            if cost - c2 < 1e-14 * max(cost, 1e-300):
                cost = c2
                break
            cost = c2
            lam = max(lam * 0.3, 1e-12)
        elif np.isfinite(c2) and c2 - cost <= 1e-12 * cost:
            break                                  # the step moves the cost by less than its rounding noise
        else:
            lam *= 10.0
            if lam > 1e12:
                break
    # re-orthonormalise against drift
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    return np.hstack([R, t[:, None]])
