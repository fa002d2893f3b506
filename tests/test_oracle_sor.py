"""Oracle of visualSLAM::SORcloud (src/rosFuncs.cpp:9-39) against an independent numpy/scipy
restatement of pcl::StatisticalOutlierRemoval (cKDTree neighbours, the same float/double steps),
and known answers: a planted far cluster goes, a regular lattice stays."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from oracle import orc


def cloud(n, seed=0, outliers=0):
    rng = np.random.default_rng(seed)
    pts = rng.normal(0, 5.0, (n, 3)).astype(np.float32)
    pts[:, 2] = -np.abs(pts[:, 2]) * 10 - 1          # the reference's map lives at negative z
    if outliers:
        pts[:outliers] += rng.normal(0, 1, (outliers, 3)).astype(np.float32) * 200
    col = rng.integers(0, 256, (n, 3)).astype(np.float32)
    return pts, col


def numpy_sor(xyz, mean_k, mul, z_limit):
    keep0 = ~(-xyz[:, 2] > z_limit) if z_limit > 0 else np.ones(len(xyz), bool)
    p = xyz[keep0]
    m = len(p)
    kk = min(mean_k, m - 1)
    if m == 0:
        return keep0, np.zeros(0, np.float32), np.zeros(0, bool)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2)              # float32 throughout, as the float kd-tree
    d2 = (d2[..., 0] + d2[..., 1]) + d2[..., 2]
    np.fill_diagonal(d2, np.inf)
    d2.sort(axis=1)
    dist = (np.sqrt(d2[:, :kk]).astype(np.float64).sum(axis=1) / max(kk, 1)).astype(np.float32) if kk else \
        np.zeros(m, np.float32)
    s = float(np.sum(dist.astype(np.float64)))
    sq = float(np.sum((dist * dist).astype(np.float64)))
    if m > 1:
        var = max((sq - s * s / m) / (m - 1), 0.0)
        thr = s / m + mul * np.sqrt(var)
    else:
        thr = np.inf
    return keep0, dist, dist <= thr


@pytest.mark.parametrize("n,k", [(1500, 200), (300, 200), (150, 200), (64, 8), (2, 200), (1, 200)])
def test_matches_numpy_restatement(n, k):
    xyz, col = cloud(n, seed=n, outliers=n // 50)
    xo, co, md = orc.sor_filter(xyz, col, mean_k=k, stddev_mul=0.01, z_limit=500.0)
    keep0, dist, keep = numpy_sor(xyz, k, 0.01, 500.0)
    assert len(md) == keep0.sum()
    # sequential sum order vs numpy's pairwise sum: the double sums are exact here, the floats equal
    assert np.array_equal(md, dist)
    assert np.array_equal(xo, xyz[keep0][keep])
    assert np.array_equal(co, col[keep0][keep])


def test_neighbours_agree_with_a_kd_tree():
    xyz, _ = cloud(800, seed=3)
    _, _, md = orc.sor_filter(xyz, None, mean_k=50, z_limit=0.0)
    d, _ = cKDTree(xyz.astype(np.float64)).query(xyz.astype(np.float64), k=51)
    assert np.allclose(md, d[:, 1:].mean(axis=1), rtol=1e-5)


def test_planted_outliers_are_removed_and_far_points_dropped_first():
    rng = np.random.default_rng(1)
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(6), indexing="ij"), -1).reshape(-1, 3)
    xyz = g.astype(np.float32) + rng.normal(0, 0.01, (len(g), 3)).astype(np.float32)
    xyz[:, 2] = -xyz[:, 2] - 1
    far = np.array([[0, 0, -900.0], [5, 5, -501.0]], np.float32)          # -z > 500: dropped before the filter
    lone = np.array([[100, 100, -100.0], [-80, 40, -60.0]], np.float32)    # isolated: removed by the filter
    allp = np.concatenate([far, xyz, lone])
    xo, _, md = orc.sor_filter(allp, None, mean_k=20, stddev_mul=1.0)
    assert len(md) == len(allp) - 2
    kept = {tuple(p) for p in xo}
    assert not any(tuple(p) in kept for p in lone) and not any(tuple(p) in kept for p in far)
    assert len(xo) > 0.8 * len(xyz)


def test_empty_and_all_far():
    xo, co, md = orc.sor_filter(np.zeros((0, 3), np.float32), None)
    assert len(xo) == 0 and len(md) == 0
    xo, _, md = orc.sor_filter(np.array([[0, 0, -600.0]] * 5, np.float32), None)
    assert len(xo) == 0 and len(md) == 0
