"""svo_map: the keyframe map in HBM and visualSLAM::updateOdometry on it
(src/optimizationStuff.cpp:17-47) against the oracle's loop over update3dtransformation
(src/keyFrameManagement.cpp:33-46): every record's camera-frame cloud re-transformed with
[R_old | t_new], mapHistory rebuilt from the records with retrack.  Bit-exact (double FMA-free
arithmetic rounded to float once, as upstream)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from ros_stereo_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def _records(seed, k, nmax):
    rng = np.random.default_rng(seed)
    recs = []
    for j in range(k):
        n = int(rng.integers(0, nmax)) if j != 2 else 0          # one empty record
        xyz = np.c_[rng.uniform(-20, 20, n), rng.uniform(-3, 3, n), rng.uniform(2, 80, n)].astype(np.float32)
        R = Rot.from_rotvec(rng.normal(0, 0.4, 3)).as_matrix()
        t = rng.normal(0, 30, 3)
        recs.append((j * 3 + 1, R, t, xyz, bool(j % 3 != 1)))    # traj_index, R, t, cloud, retrack
    return recs


def _oracle_map(orc, recs, t_new=None):
    out, counts = [], []
    for ti, R, t, xyz, retrack in recs:
        tt = t if t_new is None or ti >= len(t_new) else t_new[ti]
        upd = orc.transform_points(np.c_[R, tt], xyz) if len(xyz) else np.zeros((0, 3), np.float32)
        if retrack:
            out.append(upd)
            counts.append(len(xyz))
    return (np.vstack(out) if out else np.zeros((0, 3), np.float32)), np.array(counts, np.int32)


@pytest.mark.parametrize("k,nmax", [(1, 50), (9, 3000), (40, 4500)])
def test_map_update_matches_oracle_loop(ctx, orc, k, nmax):
    recs = _records(k, k, nmax)
    m = capi.KeyframeMap(ctx)
    for ti, R, t, xyz, retrack in recs:
        m.add_keyframe(ti, R, t, xyz, retrack)
    assert len(m) == k
    got, counts = m.points()                       # before any solve: the clouds as insertKeyFrames placed them
    want, wcounts = _oracle_map(orc, recs)
    assert np.array_equal(counts, wcounts) and np.array_equal(got, want)
    # the trajectory a pose-graph solve returns is shorter than the last record's index: that record keeps its t
    n_poses = recs[-1][0] if k > 1 else 5
    t_new = np.random.default_rng(99).normal(0, 30, (n_poses, 3))
    m.update(t_new)
    got, counts = m.points()
    want, _ = _oracle_map(orc, recs, t_new)
    assert np.array_equal(got, want)
    m.update(t_new[:0])                             # empty trajectory: nothing moves
    assert np.array_equal(m.points()[0], want)
    m.close()


def test_map_from_the_front_end_keyframes(ctx, orc):
    """The front-end's keyframe cloud goes into the map device to device; it is the camera-frame cloud
    of the oracle's keyframe (`untransformed`), and the map's world cloud is the reference set."""
    sc = synth.Scene()
    poses = synth.corridor_trajectory(4)
    frames = [sc.stereo(R, t)[:2] for R, t in poses]
    g = capi.VisualOdometry(ctx, 1241, 376, 3, grid_step=30, seed=5)
    n0 = g.init(*frames[0])
    cloud0 = g.keyframe_cloud()
    r2, r3 = g.reference()
    assert cloud0.shape == (n0, 3) and np.array_equal(cloud0, r3)       # identity pose at frame 0
    m = capi.KeyframeMap(ctx)
    assert m.add_from_vo(g, 0, np.eye(3), np.zeros(3)) == n0
    rc, R, t, ninl, kf, ntrk = g.track(*frames[1], force_keyframe=True)
    assert rc == 0 and kf
    cloud1 = g.keyframe_cloud()
    _, r3 = g.reference()
    assert np.array_equal(orc.transform_points(np.c_[R, t], cloud1), r3)
    m.add_from_vo(g, 1, R, t)
    pts, counts = m.points()
    assert list(counts) == [n0, len(cloud1)] and np.array_equal(pts[n0:], r3)
    t_new = np.array([[0.5, 0, 0], t + [0, 0.25, 0]])
    m.update(t_new)
    pts, _ = m.points()
    assert np.array_equal(pts[:n0], orc.transform_points(np.c_[np.eye(3), t_new[0]], cloud0))
    assert np.array_equal(pts[n0:], orc.transform_points(np.c_[R, t_new[1]], cloud1))
    m.close()
    g.close()
