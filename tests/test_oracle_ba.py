"""Oracle of BundleAdjust3d2d (src/bundleAdjust.cpp:551-613) against an independent dense
Levenberg-Marquardt over all 6 + 3N unknowns with numeric Jacobians (tests/ba_fixtures.py): same LM
control flow (g2o's), no Schur complement, no analytic derivatives."""
import numpy as np
import pytest

from ba_fixtures import K4, dense_lm, problem


@pytest.mark.parametrize("n,seed", [(12, 1), (40, 2), (100, 5)])
def test_schur_form_follows_the_dense_lm(orc, n, seed):
    uv, X, R0, t0, _ = problem(n, seed)
    # six iterations: chi2 is still far above the rounding floor, every accept / reject decision and the
    # damping schedule must agree with the dense solver's
    t, R, Xo, info = orc.ba_3d2d(uv, X, K4, R0, t0, iterations=6)
    td, Rd, Xd, infod = dense_lm(uv, X, K4, R0, t0, iterations=6)
    assert info["iterations"] == 6 and info["trials"] == infod["trials"]
    assert info["chi2_before"] == pytest.approx(infod["chi2_before"], rel=1e-12)
    assert info["chi2_after"] == pytest.approx(infod["chi2_after"], rel=1e-3)
    assert info["lambda_final"] == pytest.approx(infod["lambda_final"], rel=1e-6)
    assert np.abs(t - td).max() < 1e-9 and np.abs(R - Rd).max() < 1e-10
    assert np.abs(Xo - Xd).max() < 1e-7
    # the reference's ten: past iteration 7 chi2 sits at the rounding floor (1e-25) where accept / reject
    # is noise in both solvers; the estimates no longer move
    t10, R10, X10, info10 = orc.ba_3d2d(uv, X, K4, R0, t0, iterations=10)
    td10, Rd10, _, _ = dense_lm(uv, X, K4, R0, t0, iterations=10)
    assert np.abs(t10 - td10).max() < 1e-9 and np.abs(R10 - Rd10).max() < 1e-10
    # every point has one observation: the points absorb their residuals, chi2 collapses
    assert info10["chi2_after"] < 1e-12 * info10["chi2_before"]


def test_only_t_is_the_output_and_it_moves_towards_the_true_pose(orc):
    uv, X, R0, t0, (Rw, tw) = problem(440, 3, noise=0.2)
    t, R, Xo, info = orc.ba_3d2d(uv, X, K4, R0, t0)
    assert np.linalg.norm(t - tw) < 0.5 * np.linalg.norm(t0 - tw)
    # fy is not read (upstream builds CameraParameters from K(0,0) alone, :588-590)
    t2, *_ = orc.ba_3d2d(uv, X, (K4[0], 123.0, K4[2], K4[3]), R0, t0)
    assert np.array_equal(t, t2)
    # zero iterations: nothing moves
    t3, R3, X3, info3 = orc.ba_3d2d(uv, X, K4, R0, t0, iterations=0)
    assert np.array_equal(t3, t0) and np.array_equal(X3, X.astype(np.float64))
