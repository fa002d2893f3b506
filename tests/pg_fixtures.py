"""Pose-graph fixtures: a drifting loop with loop-closure edges, and an independent numpy
Gauss-Newton (scipy rotations, numeric Jacobians, dense solve) used to pin the oracle."""
import numpy as np
from scipy.spatial.transform import Rotation as Rot


def pose7(R, t):
    q = Rot.from_matrix(R).as_quat()  # x y z w
    return np.r_[t, q]


def drifting_loop(n=60, radius=20.0, yaw_drift=2e-3, scale_drift=1.003, seed=0, laps=1):
    """Ground truth: a circle in the x-z plane.  Estimate: the same relative motions with a
    small yaw and scale error per step, so the end does not meet the start."""
    rng = np.random.default_rng(seed)
    gt = []
    for i in range(n):
        a = 2 * np.pi * laps * i / n  # vertex i + n/laps revisits vertex i
        R = Rot.from_euler("y", -a).as_matrix()
        gt.append((R, np.array([radius * np.sin(a), 0.0, radius * (1 - np.cos(a))])))
    est = [gt[0]]
    for i in range(1, n):
        Rrel = gt[i - 1][0].T @ gt[i][0]
        trel = gt[i - 1][0].T @ (gt[i][1] - gt[i - 1][1])
        Rrel = Rrel @ Rot.from_euler("y", yaw_drift + rng.normal(0, 2e-4)).as_matrix()
        trel = trel * scale_drift + rng.normal(0, 2e-3, 3)
        est.append((est[-1][0] @ Rrel, est[-1][1] + est[-1][0] @ trel))
    return gt, [pose7(R, t) for R, t in est]


# ---- independent numpy restatement of the g2o semantics ---------------------------------------
def _T(p):
    T = np.eye(4)
    T[:3, :3] = Rot.from_quat(p[3:]).as_matrix()
    T[:3, 3] = p[:3]
    return T


def _err(Xi, Xj, Z):
    E = np.linalg.inv(_T(Z)) @ np.linalg.inv(_T(Xi)) @ _T(Xj)
    q = Rot.from_matrix(E[:3, :3]).as_quat()
    if q[3] < 0:
        q = -q
    return np.r_[E[:3, 3], q[:3]]


def _oplus(X, d):
    w = 1 - d[3:] @ d[3:]
    D = np.eye(4)
    if w >= 0:
        D[:3, :3] = Rot.from_quat(np.r_[d[3:], np.sqrt(w)]).as_matrix()
    D[:3, 3] = d[:3]
    T = _T(X) @ D
    return pose7(T[:3, :3], T[:3, 3])


def numpy_gauss_newton(poses, edges, iters):
    """poses: (V,7); edges: [(i, j, Z7)].  Vertex 0 fixed.  Returns [poses after each iteration], chi2s."""
    poses = np.array(poses, np.float64)
    V = len(poses)
    hist, chis = [], []
    h = 1e-6
    for it in range(iters + 1):
        H = np.zeros((6 * V, 6 * V))
        b = np.zeros(6 * V)
        chi = 0.0
        for i, j, Z in edges:
            e = _err(poses[i], poses[j], Z)
            chi += e @ e
            Ji, Jj = np.zeros((6, 6)), np.zeros((6, 6))
            for k in range(6):
                d = np.zeros(6)
                d[k] = h
                Ji[:, k] = (_err(_oplus(poses[i], d), poses[j], Z) - _err(_oplus(poses[i], -d), poses[j], Z)) / (2 * h)
                Jj[:, k] = (_err(poses[i], _oplus(poses[j], d), Z) - _err(poses[i], _oplus(poses[j], -d), Z)) / (2 * h)
            for (a, Ja) in ((i, Ji), (j, Jj)):
                b[6 * a:6 * a + 6] += Ja.T @ e
                for (c, Jc) in ((i, Ji), (j, Jj)):
                    H[6 * a:6 * a + 6, 6 * c:6 * c + 6] += Ja.T @ Jc
        chis.append(chi)
        if it == iters:
            break
        dx = np.zeros(6 * V)
        dx[6:] = np.linalg.solve(H[6:, 6:], -b[6:])
        poses = np.array([_oplus(poses[v], dx[6 * v:6 * v + 6]) for v in range(V)])
        hist.append(poses.copy())
    return hist, np.array(chis)
