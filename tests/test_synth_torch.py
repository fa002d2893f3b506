"""The torch form of the synthetic ray-caster (bench.py and the full-size GPU tests render thousands
of frames with it) must produce the images of the numpy form the oracle tests were written on."""
import numpy as np

from ros_stereo_slam_amd import synth


def test_torch_renderer_equals_numpy_renderer():
    K, size = (180.0, 180.0, 160.0, 60.0), (320, 120)
    sc = synth.Scene(wall_x=22, z_min=-38, z_max=38)
    poses = synth.loop_trajectory(40, half_x=14, half_z=30, radius=8)
    sel = [poses[0], poses[17], poses[39]]
    L, R = synth.stereo_torch(sc, sel, K=K, size=size, device="cpu", batch=2)
    for (Rm, t), l, r in zip(sel, L, R):
        ln, rn, _ = sc.stereo(Rm, t, K=K, size=size)
        # same arithmetic; a last-ulp difference of the ray-direction product may flip a rounding
        assert (l.numpy() != ln).mean() < 1e-4 and (r.numpy() != rn).mean() < 1e-4
        assert np.abs(l.numpy().astype(int) - ln).max() <= 1


def test_closed_loop_revisits_exactly_and_closures_pick_the_revisit():
    P = synth.loop_trajectory(1100, **synth.BENCH_LOOP)
    step = [np.linalg.norm(P[i + 1][1] - P[i][1]) for i in range(600)]
    assert 0.85 < min(step) and max(step) < 0.95
    yaw = [np.arccos(np.clip((np.trace(P[i][0].T @ P[i + 1][0]) - 1) / 2, -1, 1)) for i in range(600)]
    assert max(yaw) <= 0.0201          # slow yaw, <= 0.02 rad per frame
    assert np.array_equal(P[492][1], P[0][1]) and np.array_equal(P[1000][0], P[16][0])
    m = synth.loop_closures(P, max_dist=0.3, pick="nearest")
    assert all(x == -1 for x in m[:492]) and m[492] == 0 and m[1000] in (16, 508)
    # the earliest-match form keeps its old behaviour
    assert synth.loop_closures(P[:500], max_dist=2.0)[491] == 0
