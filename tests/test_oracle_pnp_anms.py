"""Pins the oracle's PnP-RANSAC (src/keyFrameManagement.cpp:84,88) and ANMS (src/ANMS.cpp:18-67)
restatements with analytic known answers and brute-force numpy restatements."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from geom_fixtures import K4, project, scene_points


def _pose():
    return Rot.from_rotvec([0.02, -0.05, 0.01]).as_matrix(), np.array([0.1, -0.05, -0.8])


def test_jacobi_eigen_matches_numpy(orc):
    rng = np.random.default_rng(0)
    for n in (3, 4, 5, 12):
        A = rng.normal(size=(n, n))
        A = A @ A.T
        w, V = orc.jacobi_eigen_sym(A)
        assert np.abs(np.sort(w) - np.linalg.eigvalsh(A)).max() < 1e-11 * np.abs(w).max()
        assert np.abs(V @ np.diag(w) @ V.T - A).max() < 1e-11 * np.abs(A).max()
        assert np.abs(V.T @ V - np.eye(n)).max() < 1e-13


@pytest.mark.parametrize("n", [5, 6, 8, 12])
def test_epnp_exact_on_noise_free_points(orc, n):
    X = scene_points(40, 1)
    R, t = _pose()
    x = project(X, R, t)
    rc, Re, te = orc.epnp(X[:n], x[:n], K4)
    assert rc == 0
    assert np.abs(Re - R).max() < 1e-9 and np.abs(te - t).max() < 1e-8


def test_epnp_rejects_coplanar_sample(orc):
    X = scene_points(5, 2)
    X[:, 1] = 1.65  # all on the ground plane: no control-point basis
    R, t = _pose()
    rc, _, _ = orc.epnp(X, project(X, R, t), K4)
    assert rc != 0


def test_pnp_ransac_recovers_pose_and_inliers(orc):
    rng = np.random.default_rng(4)
    X = scene_points(600, 3)
    R, t = _pose()
    x = project(X, R, t).astype(np.float32)
    # noise free: pose to ~float32 pixel precision, every point an inlier
    cnt, rvec, tvec, inl, iters = orc.pnp_ransac(X, x, K4, seed=1)
    assert cnt == 600 and np.array_equal(inl, np.arange(600))
    assert np.abs(rvec - [0.02, -0.05, 0.01]).max() < 1e-5 and np.abs(tvec - t).max() < 2e-4
    # outliers + noise: exact inlier index list, pose within noise
    out = rng.choice(600, 150, replace=False)
    xo = x + rng.normal(0, 0.15, x.shape).astype(np.float32)
    xo[out] += rng.uniform(10, 50, (150, 2)).astype(np.float32)
    cnt, rvec, tvec, inl, iters = orc.pnp_ransac(X, xo, K4, seed=1)
    gt = np.setdiff1d(np.arange(600), out)
    assert len(np.setdiff1d(inl, gt)) == 0            # no outlier accepted at 1 px
    assert len(inl) >= 0.97 * len(gt)
    assert np.all(np.diff(inl) > 0)
    assert np.abs(rvec - [0.02, -0.05, 0.01]).max() < 2e-4 and np.abs(tvec - t).max() < 3e-3
    assert iters < 100


def test_pnp_refine_reaches_least_squares_optimum(orc):
    """LM result must be a stationary point: compare with scipy.optimize.least_squares."""
    from scipy.optimize import least_squares
    rng = np.random.default_rng(5)
    X = scene_points(200, 6).astype(np.float32)
    R, t = _pose()
    x = (project(X.astype(np.float64), R, t) + rng.normal(0, 0.3, (200, 2))).astype(np.float32)
    idx = np.arange(200)
    rms, rvec, tvec = orc.pnp_refine(X, x, idx, K4, [0.0, 0.0, 0.0], t + 0.05)

    def res(p):
        return (project(X.astype(np.float64), Rot.from_rotvec(p[:3]).as_matrix(), p[3:]) - x).ravel()

    sol = least_squares(res, np.r_[rvec, tvec], xtol=1e-14, ftol=1e-14, gtol=1e-14)
    assert np.abs(sol.x - np.r_[rvec, tvec]).max() < 1e-6
    assert abs(rms - np.sqrt(np.mean(np.sum(res(sol.x).reshape(-1, 2) ** 2, axis=1)))) < 1e-8


def test_pnp_degenerate_inputs(orc):
    X = scene_points(4, 1)
    R, t = _pose()
    cnt, *_ = orc.pnp_ransac(X, project(X, R, t), K4)
    assert cnt == 0


def _np_anms(xy, resp, keep):
    """Brute-force restatement of src/ANMS.cpp:18-67 (stable sort, clamped index)."""
    n = len(xy)
    order = sorted(range(n), key=lambda i: (-resp[i], i))
    if n <= keep:
        return np.array(order, np.int32)
    radii = []
    for s, i in enumerate(order):
        r = np.float32(resp[i]) * np.float32(1.11)
        best = np.finfo(np.float64).max
        for j in order[:s]:
            if not (resp[j] > r):
                break
            d = xy[i] - xy[j]
            best = min(best, np.sqrt(float(d[0]) ** 2 + float(d[1]) ** 2))
        radii.append(best)
    dec = sorted(radii, reverse=True)[keep]
    return np.array([i for i, r in zip(order, radii) if r >= dec], np.int32)


@pytest.mark.parametrize("n,keep", [(300, 100), (300, 299), (50, 50), (50, 80), (200, 0)])
def test_anms_matches_bruteforce(orc, n, keep):
    rng = np.random.default_rng(n + keep)
    xy = rng.uniform(0, 500, (n, 2)).astype(np.float32)
    resp = rng.uniform(0, 1, n).astype(np.float32)
    resp[rng.integers(0, n, n // 5)] = resp[0]  # ties
    got, _ = orc.anms(xy, resp, keep)
    assert np.array_equal(got, _np_anms(xy, resp, keep))


def test_anms_zero_response_keeps_everything(orc):
    """The reference's grid keypoints have response 0 (src/tracking.cpp:8): every radius is
    DBL_MAX and nothing is suppressed."""
    xy = np.stack(np.meshgrid(np.arange(10.0), np.arange(10.0)), -1).reshape(-1, 2).astype(np.float32)
    got, radii = orc.anms(xy, np.zeros(100, np.float32), 30)
    assert len(got) == 100 and np.array_equal(got, np.arange(100))


# ---- cv::solvePnP (ITERATIVE, no guess): the last rung of the older ladder, src/bundleAdjust.cpp:470-477 ----
def test_solve_pnp_recovers_a_known_pose_and_is_a_least_squares_optimum(orc):
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation as Rot

    from geom_fixtures import K4, project, scene_points

    R = Rot.from_rotvec([0.02, -0.05, 0.01]).as_matrix()
    t = np.array([0.1, -0.05, -0.8])
    for n in (6, 7, 60):
        X = scene_points(n, 3)
        rc, rv, tv, rms = orc.solve_pnp(X, project(X, R, t), K4)
        assert rc == 0 and np.abs(rv - Rot.from_matrix(R).as_rotvec()).max() < 1e-6 and np.abs(tv - t).max() < 2e-5
    # with noise the answer is the stationary point of the reprojection error over ALL points
    X = scene_points(300, 4).astype(np.float32)
    x = (project(X, R, t) + np.random.default_rng(0).normal(0, 0.4, (300, 2))).astype(np.float32)
    rc, rv, tv, rms = orc.solve_pnp(X, x, K4)
    assert rc == 0

    def res(p):
        Xc = X.astype(np.float64) @ Rot.from_rotvec(p[:3]).as_matrix().T + p[3:]
        return np.c_[K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2] - x[:, 0], K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3] - x[:, 1]].ravel()

    sol = least_squares(res, np.r_[rv, tv], xtol=1e-15, ftol=1e-15, gtol=1e-15)
    assert np.abs(sol.x - np.r_[rv, tv]).max() < 1e-6
    assert rms == pytest.approx(np.sqrt(np.sum(res(np.r_[rv, tv]) ** 2) / 300), rel=1e-6)


def test_solve_pnp_refuses_what_upstream_cannot_solve_by_dlt(orc):
    from geom_fixtures import K4, project, scene_points

    X = scene_points(100, 1)
    assert orc.solve_pnp(X[:5], project(X[:5]), K4)[0] == -1          # fewer than 6 points
    Xl = X.copy()
    Xl[:, 1] = 1.0
    Xl[:, 2] = 10.0                                                   # collinear: no DLT, no homography
    assert orc.solve_pnp(Xl, project(Xl), K4)[0] == -2


def test_oracle_solve_pnp_planar_branch_recovers_the_pose(orc):
    """cv::solvePnP's planar (homography) initialisation, restated in oracle/pnp.c: object points in one plane --
    fronto-parallel, the ground plane, a tilted wall -- projected with a known pose come back to 1e-6 noise-free, and the
    result is a stationary point of the reprojection error (scipy least_squares started there does not move)."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation as Rot

    from geom_fixtures import K4, project

    rng = np.random.default_rng(23)
    R, t = Rot.from_rotvec([0.02, -0.05, 0.01]).as_matrix(), np.array([0.1, -0.05, -0.8])
    n = 300
    a, b = rng.uniform(-5, 5, n), rng.uniform(-3, 3, n)
    planes = [np.c_[a, b, np.full(n, 12.0)], np.c_[rng.uniform(-6, 6, n), np.full(n, 1.65), rng.uniform(6, 40, n)],
              np.c_[a, b, 15 + 0.4 * a - 0.25 * b]]
    for X in planes:
        X = X.astype(np.float32)
        x0 = project(X, R, t)
        rc, rv, tv, rms = orc.solve_pnp(X, x0.astype(np.float32), K4)
        assert rc == 0 and np.abs(tv - t).max() < 2e-4 and np.abs(Rot.from_rotvec(rv).as_matrix() - R).max() < 2e-5
        x = (x0 + rng.normal(0, 0.3, x0.shape)).astype(np.float32)
        rc, rv, tv, rms = orc.solve_pnp(X, x, K4)
        assert rc == 0

        def res(p):
            return (project(X, Rot.from_rotvec(p[:3]).as_matrix(), p[3:]) - x).ravel()

        sol = least_squares(res, np.r_[rv, tv], method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15)
        assert np.abs(sol.x - np.r_[rv, tv]).max() < 1e-5
        assert np.sqrt(np.mean(res(np.r_[rv, tv]).reshape(-1, 2) ** 2)) <= np.sqrt(np.mean(sol.fun.reshape(-1, 2) ** 2)) * (1 + 1e-9) + 1e-12
