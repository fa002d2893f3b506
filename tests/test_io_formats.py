"""The data formats either side of the hot path: KITTI pose files,
trajectory.csv, ATE / RPE, rosPublish's map and pose conversions, the PLY dump.  Host code inside
libsvo_hip.so -- no GPU is touched, so these run everywhere."""
import struct

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from ros_stereo_slam_amd import capi


def _traj(n, seed=0):
    rng = np.random.default_rng(seed)
    R, t = [np.eye(3)], [np.zeros(3)]
    for _ in range(n - 1):
        dR = Rot.from_rotvec(rng.normal(0, 0.03, 3)).as_matrix()
        t.append(t[-1] + R[-1] @ (np.array([0, 0, 0.8]) + rng.normal(0, 0.02, 3)))
        R.append(R[-1] @ dR)
    return np.array(R), np.array(t)


def test_kitti_pose_round_trip_and_layout(tmp_path):
    R, t = _traj(17)
    p = tmp_path / "00.txt"
    capi.write_kitti_poses(p, R, t)
    rows = [l.split() for l in p.read_text().splitlines()]
    assert len(rows) == 17 and all(len(r) == 12 for r in rows)
    first = np.array(rows[3], float).reshape(3, 4)             # [R | t], row-major
    assert np.allclose(first[:, :3], R[3], atol=1e-9) and np.allclose(first[:, 3], t[3], atol=1e-9)
    R2, t2 = capi.read_kitti_poses(p)
    assert np.allclose(R2, R, atol=1e-9) and np.allclose(t2, t, atol=1e-9)
    (tmp_path / "bad.txt").write_text("1 0 0 0 0 1 0 0 0 0 1\n")  # 11 numbers
    with pytest.raises(capi.SvoError):
        capi.read_kitti_poses(tmp_path / "bad.txt")
    with pytest.raises(capi.SvoError):
        capi.read_kitti_poses(tmp_path / "missing.txt")


def test_trajectory_csv_is_the_reference_layout(tmp_path):
    """createData / appendData (include/monoUtils.h:23-49): header, 8 values per row, each followed by ','."""
    p = tmp_path / "trajectory.csv"
    capi.trajectory_csv(p, [[0, 1.5, -2, 3, 1.25, -2, 3, 0]], create=True)
    capi.trajectory_csv(p, [[1, 2.5, 0, 4, 2.5, 0, 4, 0], [2, 1e-7, 0, 5, 0, 0, 5, 0]], create=False)
    lines = p.read_text().splitlines()
    assert lines[0] == "Idx,Xm,Ym,Zm,Xgt,Ygt,Zgt,Const"
    assert lines[1] == "0,1.5,-2,3,1.25,-2,3,0,"
    assert lines[2] == "1,2.5,0,4,2.5,0,4,0," and lines[3].startswith("2,1e-07,0,5,")
    assert len(lines) == 4


def test_ate_and_rpe_against_numpy():
    Rg, tg = _traj(40, seed=1)
    rng = np.random.default_rng(2)
    Re = np.array([R @ Rot.from_rotvec(rng.normal(0, 0.002, 3)).as_matrix() for R in Rg])
    te = tg + rng.normal(0, 0.05, tg.shape)
    assert capi.ate_rmse(te, tg) == pytest.approx(np.sqrt(np.mean(np.sum((te - tg) ** 2, axis=1))), rel=1e-12)
    for delta in (1, 5):
        et, er = [], []
        for i in range(len(Rg) - delta):
            j = i + delta
            Pr, pt = Re[i].T @ Re[j], Re[i].T @ (te[j] - te[i])
            Qr, qt = Rg[i].T @ Rg[j], Rg[i].T @ (tg[j] - tg[i])
            Er, Et = Qr.T @ Pr, Qr.T @ (pt - qt)
            et.append(np.sum(Et ** 2))
            er.append(np.linalg.norm(Rot.from_matrix(Er).as_rotvec()) ** 2)
        a, b = capi.rpe(Re, te, Rg, tg, delta)
        assert a == pytest.approx(np.sqrt(np.mean(et)), rel=1e-9)
        assert b == pytest.approx(np.sqrt(np.mean(er)), rel=1e-6)
    a, b = capi.rpe(Rg, tg, Rg, tg, 1)
    assert a < 1e-12 and b < 1e-7
    with pytest.raises(capi.SvoError):
        capi.rpe(Rg[:3], tg[:3], Rg[:3], tg[:3], 5)


def test_ros_map_points_and_ply(tmp_path):
    xyz = np.array([[1, 2, -3], [4, 5, -600], [-1, 0.5, -499.9]], np.float32)
    bgr = np.array([[10, 20, 30], [1, 2, 3], [255, 128, 0]], np.float32)
    out, rgb = capi.ros_map_points(xyz, bgr)
    # src/rosFuncs.cpp:54-60: -z > 500 skipped; (x, z, -y) * 0.1; r = c.z, g = c.y, b = c.x
    assert np.allclose(out, [[0.1, -0.3, -0.2], [-0.1, -49.99, -0.05]], atol=1e-6)
    assert rgb.tolist() == [[30, 20, 10], [0, 128, 255]]
    p = tmp_path / "map.ply"
    capi.write_ply(p, out, rgb)
    raw = p.read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\n") and b"element vertex 2\n" in head
    assert b"property float x" in head and b"property uchar blue" in head
    assert len(body) == 2 * 15
    x, y, z, r, g, b = struct.unpack("<fffBBB", body[:15])
    assert (x, y, z) == pytest.approx((0.1, -0.3, -0.2), abs=1e-6) and (r, g, b) == (30, 20, 10)
    out2, none = capi.ros_map_points(xyz)
    assert none is None and np.array_equal(out2, out)


def test_ros_pose_reproduces_the_reference_convention():
    """src/rosFuncs.cpp:69-91 with Rmat2Quat (include/monoUtils.h:215-227): the Rodrigues vector's
    components are used as X, Y, Z angles -- reproduced as it is, quirk included."""
    rng = np.random.default_rng(4)
    for _ in range(20):
        rv = rng.normal(0, 0.8, 3)
        R = Rot.from_rotvec(rv).as_matrix()
        t = rng.normal(0, 10, 3)
        pos, q = capi.ros_pose(R, t)
        assert np.allclose(pos, [0.1 * t[0], 0.1 * t[2], -0.1 * t[1]])
        qq = (Rot.from_rotvec([rv[0], 0, 0]) * Rot.from_rotvec([0, rv[1], 0]) * Rot.from_rotvec([0, 0, rv[2]])).as_quat()
        if qq[3] < 0 and q[3] > 0 or qq[3] > 0 and q[3] < 0:
            qq = -qq
        assert np.allclose(q, [qq[0], qq[2], -qq[1], qq[3]], atol=1e-9)
    pos, q = capi.ros_pose(np.eye(3), np.zeros(3))
    assert np.allclose(q, [0, 0, 0, 1]) and np.allclose(pos, 0)
