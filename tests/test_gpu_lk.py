"""GPU parity: HIP pyramid + LK (through the C ABI) against the CPU oracle -- bit-exact,
as the path is integer arithmetic plus order-independent exact sums (DESIGN.md numerics)."""
import numpy as np
import pytest

from ros_stereo_slam_amd import synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("shape", [(376, 1241, 3), (376, 1241, 1), (97, 131, 3), (50, 64, 1)])
def test_pyramid_bit_exact(ctx, orc, shape):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    pyr = ctx.pyramid(shape[1], shape[0], shape[2]).build(img)
    ref = img
    for l in range(4):
        got = pyr.level(l)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), f"level {l}"
        ref = orc.pyr_down(ref)
    pyr.close()


def _compare_lk(ctx, orc, a, b, pts):
    h, w, c = a.shape
    pa, pb = ctx.pyramid(w, h, c).build(a), ctx.pyramid(w, h, c).build(b)
    out, st, err, me = ctx.lk_track(pa, pb, pts)
    ro, rs, re, rm = orc.lk_track(a, b, pts)
    pa.close()
    pb.close()
    assert np.array_equal(st, rs), f"status differs at {np.nonzero(st != rs)[0][:10]}"
    assert np.array_equal(_bits(me), _bits(rm)), "minEig bits differ"
    bad = np.nonzero((_bits(out) != _bits(ro)).any(1))[0]
    assert bad.size == 0, f"{bad.size} points differ, first {bad[:5]}: {out[bad[:5]]} vs {ro[bad[:5]]}"
    assert np.array_equal(_bits(err), _bits(re)), "err bits differ"
    return out, st


@pytest.mark.parametrize("c", [1, 3])
@pytest.mark.parametrize("shift", [(0.0, 0.0), (2.3, -1.4), (-9.6, 5.2)])
def test_lk_textured_bit_exact(ctx, orc, c, shift):
    a, b = synth.textured_pair(320, 200, c, shift=shift, seed=3)
    pts = orc.grid_keypoints(200, 320, 10)
    out, st = _compare_lk(ctx, orc, a, b, pts)
    inner = (pts[:, 0] > 45) & (pts[:, 0] < 275) & (pts[:, 1] > 45) & (pts[:, 1] < 155)
    assert np.abs((out - pts)[inner] - np.array(shift, np.float32)).max() < 0.12


def test_lk_edge_cases_bit_exact(ctx, orc):
    a, b = synth.textured_pair(200, 160, 3, shift=(1.0, 0.5), seed=9)
    a[60:110, 70:130] = 100
    b[60:110, 70:130] = 100
    rng = np.random.default_rng(2)
    pts = np.concatenate([
        np.array([[100, 85], [30, 30], [-40, 50], [199, 159], [400, 80], [0, 0], [-10.5, -10.5],
                  [199.99, 0.01], [10.0, 159.5], [-11.0, 80.0], [210.0, 80.0], [100.0, 170.5]], np.float32),
        rng.uniform([-15, -15], [215, 175], (500, 2)).astype(np.float32)])
    _compare_lk(ctx, orc, a, b, pts)


def test_lk_empty_input(ctx):
    a, b = synth.textured_pair(100, 80, 3)
    pa, pb = ctx.pyramid(100, 80, 3).build(a), ctx.pyramid(100, 80, 3).build(b)
    out, st, err, me = ctx.lk_track(pa, pb, np.zeros((0, 2), np.float32))
    assert out.shape == (0, 2) and st.shape == (0,)


def test_lk_stereo_full_size_bit_exact(ctx, orc):
    """BASELINE config size: 1241x376x3, grid step 10 (4428 points), left -> right."""
    sc = synth.Scene()
    left, right, depth = sc.stereo(np.eye(3), np.zeros(3))
    pts = orc.grid_keypoints(376, 1241, 10)
    out, st = _compare_lk(ctx, orc, left, right, pts)
    assert st.sum() > 0.8 * len(pts)
    z = depth[pts[:, 1].astype(int), pts[:, 0].astype(int)]
    ok = (st == 1) & (z > 0)
    disp = synth.KITTI_K[0] * synth.KITTI_BASELINE / z[ok]
    assert np.median(np.abs(-(out - pts)[ok, 0] - disp)) < 0.3


def test_lk_temporal_full_size_bit_exact(ctx, orc):
    sc = synth.Scene()
    poses = synth.corridor_trajectory(2)
    f0, _ = sc.render(*poses[0])
    f1, _ = sc.render(*poses[1])
    pts = orc.grid_keypoints(376, 1241, 10)
    _compare_lk(ctx, orc, f0, f1, pts)


def test_grid_keypoints(ctx, orc):
    for step in (30, 10, 7):
        assert np.array_equal(ctx.grid_keypoints(376, 1241, step), orc.grid_keypoints(376, 1241, step))
