"""Shared synthetic two-view / PnP fixtures (seeded, analytic ground truth)."""
import numpy as np
from scipy.spatial.transform import Rotation as Rot

K4 = (718.856, 718.856, 607.1928, 185.2157)
K = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1.0]])
BASELINE = 0.54


def project(X, R=np.eye(3), t=np.zeros(3)):
    Xc = X @ np.asarray(R).T + np.asarray(t)
    x = Xc @ K.T
    return x[:, :2] / x[:, 2:]


def scene_points(n, seed=0):
    rng = np.random.default_rng(seed)
    return np.c_[rng.uniform(-8, 8, n), rng.uniform(-2, 1.6, n), rng.uniform(5, 40, n)]


def two_view(n=400, n_out=80, seed=0, rotvec=(0.01, -0.03, 0.005), t=(0.05, -0.02, 0.9), noise=0.0):
    """Correspondences of a moving camera (x2 = K (R X + t)) with gross outliers."""
    rng = np.random.default_rng(seed + 1000)
    X = scene_points(n, seed)
    R = Rot.from_rotvec(rotvec).as_matrix()
    x1, x2 = project(X), project(X, R, np.array(t))
    if noise:
        x1 = x1 + rng.normal(0, noise, x1.shape)
        x2 = x2 + rng.normal(0, noise, x2.shape)
    out = rng.choice(n, n_out, replace=False)
    x2 = x2.copy()
    x2[out] += rng.uniform(20, 60, (n_out, 2)) * rng.choice([-1, 1], (n_out, 2))
    gt = np.ones(n, bool)
    gt[out] = False
    return x1.astype(np.float32), x2.astype(np.float32), gt, X, R, np.array(t)
