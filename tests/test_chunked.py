"""The multi-GPU path on CPU: chunk partitioning, and the all-gather of
chunk-boundary poses + rebasing under torch.distributed with the gloo backend, world size 2
and 3 (one process per rank, as one process per GPU on the node)."""
import os
import socket

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from ros_stereo_slam_amd import chunked


def test_chunk_bounds_cover_with_one_frame_overlap():
    for n, g in ((4541, 8), (200, 2), (10, 3), (5, 8), (2, 1)):
        b = chunked.chunk_bounds(n, g)
        assert b[0][0] == 0 and b[-1][1] == n - 1
        for (s0, e0), (s1, e1) in zip(b, b[1:]):
            assert e0 == s1 and e0 > s0
        sizes = [e - s for s, e in b]
        assert max(sizes) - min(sizes) <= 1
    assert chunked.chunk_bounds(4541, 8)[0] == (0, 568)
    with pytest.raises(ValueError):
        chunked.chunk_bounds(1, 2)


def _trajectory(n, seed=0):
    rng = np.random.default_rng(seed)
    poses = [(np.eye(3), np.zeros(3))]
    for _ in range(n - 1):
        dR = Rot.from_rotvec(rng.normal(0, 0.02, 3)).as_matrix()
        dt = np.array([0.0, 0.0, 0.9]) + rng.normal(0, 0.02, 3)
        poses.append(chunked.compose(*poses[-1], dR, dt))
    return poses


def test_prefix_and_rebase_reproduce_the_global_trajectory():
    gt = _trajectory(41)
    bounds = chunked.chunk_bounds(41, 4)
    local, boundaries = [], []
    for s, e in bounds:
        R0, t0 = gt[s]
        loc = [(R0.T @ R, R0.T @ (t - t0)) for R, t in gt[s:e + 1]]  # what a chunk's VO would output
        local.append(loc)
        boundaries.append(loc[-1])
    starts = chunked.prefix_transforms(boundaries)
    for (s, e), loc, st in zip(bounds, local, starts):
        for (R, t), (Rg, tg) in zip(chunked.rebase(loc, *st), gt[s:e + 1]):
            assert np.abs(R - Rg).max() < 1e-12 and np.abs(t - tg).max() < 1e-10


def test_pose7_and_ate():
    R = Rot.from_rotvec([0.3, -0.4, 2.9]).as_matrix()
    p = chunked.pose7(R, [1, 2, 3])
    assert np.abs(Rot.from_quat(p[3:]).as_matrix() - R).max() < 1e-14 and p[6] >= 0
    assert chunked.ate_rmse([[0, 0, 1], [0, 0, 2]], [[0, 0, 1], [0, 3, 2]]) == pytest.approx(np.sqrt(4.5))


def _worker(rank, world, port, n, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gt = _trajectory(n)
        s, e = chunked.chunk_bounds(n, world)[rank]
        R0, t0 = gt[s]
        local = [(R0.T @ R, R0.T @ (t - t0)) for R, t in gt[s:e + 1]]
        mine, traj = chunked.stitch(dist, local)
        err_mine = max(np.abs(R - Rg).max() + np.abs(t - tg).max() for (R, t), (Rg, tg) in zip(mine, gt[s:e + 1]))
        err_traj = max(np.abs(R - Rg).max() + np.abs(t - tg).max() for (R, t), (Rg, tg) in zip(traj, gt))
        q.put((rank, len(traj), err_mine, err_traj))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_stitch_over_gloo(world):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n = 25
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, ntraj, e_mine, e_traj in res:
        assert ntraj == n                      # overlap frames are not duplicated
        assert e_mine < 1e-10 and e_traj < 1e-10


def _worker_multi(rank, world, port, n, m, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gt = _trajectory(n)
        bounds = chunked.chunk_bounds(n, world * m)[rank * m:(rank + 1) * m]  # this rank's m chunks
        local = []
        for s, e in bounds:
            R0, t0 = gt[s]
            local.append([(R0.T @ R, R0.T @ (t - t0)) for R, t in gt[s:e + 1]])
        boundaries = chunked.all_gather_chunk_boundaries(dist, [loc[-1] for loc in local])
        starts = chunked.prefix_transforms(boundaries)
        err = 0.0
        for k, ((s, e), loc) in enumerate(zip(bounds, local)):
            for (R, t), (Rg, tg) in zip(chunked.rebase(loc, *starts[rank * m + k]), gt[s:e + 1]):
                err = max(err, np.abs(R - Rg).max() + np.abs(t - tg).max())
        q.put((rank, len(boundaries), err))
    finally:
        dist.destroy_process_group()


def test_several_chunks_per_rank_over_gloo():
    """svo_vo_run_chunks layout: every rank owns m chunks; one all-gather carries all m
    boundary poses of every rank and the global chunk order is rank-major."""
    import torch.multiprocessing as mp

    world, m, n = 2, 3, 37
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_multi, args=(r, world, port, n, m, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nb, err in res:
        assert nb == world * m and err < 1e-10
