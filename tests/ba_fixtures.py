"""Shared by the BundleAdjust3d2d tests: a synthetic 3D-2D problem and an independent dense
Levenberg-Marquardt on all 6 + 3N unknowns (numeric Jacobians, no Schur complement) that follows
g2o's OptimizationAlgorithmLevenberg control flow -- the check the oracle's Schur form is held to."""
import numpy as np
from scipy.spatial.transform import Rotation as Rot

K4 = (718.856, 718.856, 607.1928, 185.2157)


def problem(n, seed=0, noise=0.3, pose_err=1.0):
    rng = np.random.default_rng(seed)
    Rw = Rot.from_rotvec([0.01, -0.02, 0.005]).as_matrix()
    tw = np.array([0.1, -0.05, 0.3])
    X = np.c_[rng.uniform(-8, 8, n), rng.uniform(-2, 2, n), rng.uniform(6, 40, n)].astype(np.float32)
    xc = (Rw @ X.astype(np.float64).T).T + tw
    uv = np.c_[xc[:, 0] / xc[:, 2] * K4[0] + K4[2], xc[:, 1] / xc[:, 2] * K4[0] + K4[3]]
    uv = (uv + rng.normal(0, noise, (n, 2))).astype(np.float32)
    R0 = Rot.from_rotvec(np.array([0.01, -0.02, 0.005]) + pose_err * np.array([0.002, 0.002, -0.001])).as_matrix()
    t0 = tw + pose_err * np.array([0.05, -0.02, 0.04])
    return uv, X, R0, t0, (Rw, tw)


def se3_exp(d):
    w, u = np.asarray(d[:3], float), np.asarray(d[3:], float)
    th = np.linalg.norm(w)
    O = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-5:
        R = np.eye(3) + O + O @ O
        V = R
    else:
        R = np.eye(3) + np.sin(th) / th * O + (1 - np.cos(th)) / th ** 2 * O @ O
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * O + (th - np.sin(th)) / th ** 3 * O @ O
    return R, V @ u


def dense_lm(uv, X, K4, R, t, iterations=10):
    f, cx, cy = K4[0], K4[2], K4[3]
    uv = np.asarray(uv, np.float64)
    X = np.asarray(X, np.float64).copy()
    R, t = np.asarray(R, float).copy(), np.asarray(t, float).copy()
    n = len(X)

    def err(R, t, X):
        xc = (R @ X.T).T + t
        return (uv - np.c_[xc[:, 0] / xc[:, 2] * f + cx, xc[:, 1] / xc[:, 2] * f + cy]).ravel()

    def oplus(R, t, X, dx):
        E, u = se3_exp(dx[:6])
        return E @ R, E @ t + u, X + dx[6:].reshape(n, 3)

    lam, ni, trials = 0.0, 2.0, 0
    chi_first = chi_last = None
    for it in range(iterations):
        e = err(R, t, X)
        chi = e @ e
        if it == 0:
            chi_first = chi_last = chi
        J = np.zeros((2 * n, 6 + 3 * n))
        h = 1e-6
        for k in range(6 + 3 * n):
            d = np.zeros(6 + 3 * n)
            d[k] = h
            ep = err(*oplus(R, t, X, d))
            d[k] = -h
            em = err(*oplus(R, t, X, d))
            J[:, k] = (ep - em) / (2 * h)
        H, b = J.T @ J, -J.T @ e
        if it == 0:
            lam, ni = 1e-5 * np.max(np.abs(np.diag(H))), 2.0
        rho, qmax = 0.0, 0
        while True:
            dx = np.linalg.solve(H + lam * np.eye(len(b)), b)
            Rn, tn, Xn = oplus(R, t, X, dx)
            en = err(Rn, tn, Xn)
            temp = en @ en
            trials += 1
            rho = (chi - temp) / (dx @ (lam * dx + b) + 1e-3)
            if rho > 0 and np.isfinite(temp):
                lam *= max(1 / 3, min(2 / 3, 1 - (2 * rho - 1) ** 3))
                ni = 2.0
                R, t, X = Rn, tn, Xn
                chi_last = temp
            else:
                lam *= ni
                ni *= 2
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        if qmax == 10 or rho == 0:
            break
    return t, R, X, dict(chi2_before=chi_first, chi2_after=chi_last, lambda_final=lam, trials=trials)
