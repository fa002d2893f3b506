"""What "parity" of the LK tracker rests on.  tests/lk_numpy.py restates
cv::calcOpticalFlowPyrLK from SURVEY.md appendix A.1 ALONE -- float32 accumulation of the normal matrix and of the
mismatch vector as upstream does it, in the order of its scalar build and of its 4- and 8-lane SIMD builds -- without
looking at oracle/lk.c or lk.hip.  The oracle (and the GPU, which equals it bit for bit) instead sums the integer
products exactly and rounds once (deviation 1).  Measured here:
  * the oracle agrees with the float32 restatement to ~0.01 px with IDENTICAL status -- on analytic shifts and on
    frames of the benchmark stream (temporal and stereo pairs);
  * two upstream builds differ from EACH OTHER by as much: the exact sums sit inside upstream's cross-build spread;
  * against the analytic shift the restatement makes the same error as the oracle (0.07 px max at the texture's 9 px
    wavelength, 0.17-0.24 px at 20 px): that error is the tracker's on that texture (0.01 px stop rule, fixed-point
    patches), not the oracle's -- why tests/test_oracle_lk.py bounds it by 0.12 px and not by SURVEY 8c's 0.05 px."""
import numpy as np
import pytest

import lk_numpy as L
from ros_stereo_slam_amd import synth

ORDERS = ("scalar", "simd4", "simd8")


def _compare(orc, a, b, pts):
    o, st, err, _ = orc.lk_track(a, b, pts)
    r = {k: L.calc_optical_flow_pyr_lk(a, b, pts, k) for k in ORDERS}
    for k in ORDERS:
        assert np.array_equal(st, r[k][1]), f"status differs between the oracle and the {k} build"
    ok = st == 1
    d_oracle = max(float(np.abs(o - r[k][0])[ok].max()) for k in ORDERS)            # oracle vs any upstream build
    d_builds = max(float(np.abs(r["scalar"][0] - r[k][0])[ok].max()) for k in ("simd4", "simd8"))
    d_err = float(np.abs(err - r["scalar"][2])[ok].max())
    return o, r, ok, d_oracle, d_builds, d_err


@pytest.mark.parametrize("shift,wavelength", [((2.3, -1.4), 9.0), ((-6.75, 3.2), 9.0), ((11.5, 7.25), 14.0),
                                              ((2.3, -1.4), 20.0)])
def test_exact_sums_sit_inside_upstreams_cross_build_spread_on_analytic_shifts(orc, shift, wavelength):
    a, b = synth.textured_pair(320, 200, 3, shift=shift, seed=3, wavelength=wavelength)
    pts = orc.grid_keypoints(200, 320, 20)
    o, r, ok, d_oracle, d_builds, d_err = _compare(orc, a, b, pts)
    assert d_oracle < 0.02 and d_builds < 0.02                       # pixels; measured 0.001 .. 0.013
    assert d_oracle <= 2.0 * d_builds + 3e-3                         # no further from a build than builds are apart
    assert d_err < 0.01                                              # the err output (mean |J - I| / 32)
    inner = ok & (pts[:, 0] > 45) & (pts[:, 0] < 275) & (pts[:, 1] > 45) & (pts[:, 1] < 155)
    e_oracle = np.abs((o - pts)[inner] - np.array(shift, np.float32)).max()
    e_numpy = np.abs((r["scalar"][0] - pts)[inner] - np.array(shift, np.float32)).max()
    assert abs(e_oracle - e_numpy) < 5e-3, (e_oracle, e_numpy)      # the SAME error against the truth
    assert e_oracle < (0.08 if wavelength <= 9.0 else 0.26)          # 0.07 at 9 px, 0.13 at 14 px, 0.17-0.24 at 20 px


def test_exact_sums_sit_inside_upstreams_cross_build_spread_on_the_benchmark_stream(orc):
    """Two consecutive frames and one stereo pair of bench.py's stream at full size, a 40 px lattice."""
    scene = synth.bench_scene()
    poses = synth.loop_trajectory(2, **synth.BENCH_LOOP)
    (l0, r0), (l1, _) = scene.stereo(*poses[0])[:2], scene.stereo(*poses[1])[:2]
    pts = orc.grid_keypoints(376, 1241, 40)
    for a, b, name in ((l0, l1, "t -> t+1"), (l0, r0, "left -> right")):
        o, r, ok, d_oracle, d_builds, d_err = _compare(orc, a, b, pts)
        assert ok.sum() > 150, name
        assert d_oracle < 0.03 and d_builds < 0.03, (name, d_oracle, d_builds)      # measured 0.0003 .. 0.012
        assert d_oracle <= 2.0 * d_builds + 3e-3, (name, d_oracle, d_builds)
        print(f"\\n{name}: {int(ok.sum())} of {len(pts)} tracked, oracle vs upstream builds {d_oracle:.4f} px, "
              f"builds among themselves {d_builds:.4f} px, status identical")


def test_the_restatement_reproduces_the_building_blocks(orc):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    assert np.array_equal(L.pyr_down(img), orc.pyr_down(img))
    dx, dy = L.scharr(img)
    got = orc.scharr(img)
    assert np.array_equal(got[..., 0], dx) and np.array_equal(got[..., 1], dy)


def test_the_oracles_channel_loops_on_colour_images(orc):
    """every 3-channel parity image used to have R = G = B, so the oracle's own channel loops
    (interleaved BGR rows, sums over the channels, minEig not divided by C) were checked by nothing but the oracle.
    Colour textures (three different channels) and a colour stereo / temporal pair of the benchmark's scene: the numpy
    restatement of appendix A.1 against oracle/lk.c -- status identical, positions inside upstream's cross-build
    spread -- and the building blocks (pyrDown, Scharr) channel by channel."""
    a, b = synth.textured_pair(320, 200, 3, shift=(2.3, -1.4), seed=3, colour=True)
    assert min(np.abs(a[..., i].astype(int) - a[..., j].astype(int)).mean() for i, j in ((0, 1), (0, 2), (1, 2))) > 5
    pts = orc.grid_keypoints(200, 320, 20)
    o, r, ok, d_oracle, d_builds, d_err = _compare(orc, a, b, pts)
    assert d_oracle < 0.02 and d_builds < 0.02 and d_oracle <= 2.0 * d_builds + 3e-3 and d_err < 0.01
    inner = ok & (pts[:, 0] > 45) & (pts[:, 0] < 275) & (pts[:, 1] > 45) & (pts[:, 1] < 155)
    assert np.abs((o - pts)[inner] - np.array((2.3, -1.4), np.float32)).max() < 0.12
    # a channel permutation of ONE image must change the answer of both implementations alike
    b_rot = np.ascontiguousarray(b[..., [1, 2, 0]])
    o2, r2, ok2, d2, _, _ = _compare(orc, a, b_rot, pts)
    assert d2 < 0.05
    assert (ok != ok2).any() or np.abs(o - o2)[ok & ok2].max() > 0.05
    assert np.array_equal(L.pyr_down(a), orc.pyr_down(a))
    dx, dy = L.scharr(a)
    got = orc.scharr(a)
    assert np.array_equal(got[..., 0], dx) and np.array_equal(got[..., 1], dy)
    scene = synth.bench_scene(colour=True)
    poses = synth.loop_trajectory(2, **synth.BENCH_LOOP)
    (l0, r0), (l1, _) = scene.stereo(*poses[0])[:2], scene.stereo(*poses[1])[:2]
    pts = orc.grid_keypoints(376, 1241, 40)
    for x, y, name in ((l0, l1, "t -> t+1"), (l0, r0, "left -> right")):
        o, r, ok, d_oracle, d_builds, d_err = _compare(orc, x, y, pts)
        assert ok.sum() > 120, name
        assert d_oracle < 0.03 and d_builds < 0.03 and d_oracle <= 2.0 * d_builds + 3e-3, (name, d_oracle, d_builds)
        print(f"\ncolour {name}: {int(ok.sum())} of {len(pts)} tracked, oracle vs upstream builds {d_oracle:.4f} px, "
              f"builds among themselves {d_builds:.4f} px, status identical")
