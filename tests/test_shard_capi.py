"""The chunk-sharded batch's exchange step behind the C ABI (svo_shard_*).  Here, without a GPU: the
host arithmetic (prefix composition, rebasing) against chunked.py's numpy form.  tests/test_gpu_sharded.py runs the
RCCL all-gather itself on a one-rank communicator."""
import numpy as np
from scipy.spatial.transform import Rotation as Rot

from ros_stereo_slam_amd import capi, chunked


def _poses(n, seed):
    rng = np.random.default_rng(seed)
    return [(Rot.from_rotvec(rng.normal(0, 0.3, 3)).as_matrix(), rng.normal(0, 5, 3)) for _ in range(n)]


def test_prefix_starts_and_rebase_match_the_numpy_form():
    b = _poses(9, 1)
    s_c, s_np = capi.shard_prefix_starts(b), chunked.prefix_transforms(b)
    assert len(s_c) == len(s_np) == 9
    assert np.array_equal(s_c[0][0], np.eye(3)) and not s_c[0][1].any()
    for (Ra, ta), (Rb, tb) in zip(s_c, s_np):
        assert np.abs(Ra - Rb).max() < 1e-13 and np.abs(ta - tb).max() < 1e-12
    local = _poses(6, 2)
    for (Ra, ta), (Rb, tb) in zip(capi.shard_rebase(s_c[4], local), chunked.rebase(local, *s_np[4])):
        assert np.abs(Ra - Rb).max() < 1e-13 and np.abs(ta - tb).max() < 1e-12
    assert capi.shard_rebase(s_c[2], []) == []


def test_stitching_through_the_c_abi_equals_the_sequential_trajectory():
    """Cut a trajectory into chunks (one frame of overlap), express every chunk relative to its first frame, then
    boundaries -> prefix starts -> rebase must give the trajectory back."""
    traj = [(np.eye(3), np.zeros(3))]
    for R, t in _poses(20, 3):
        R = Rot.from_rotvec(0.05 * Rot.from_matrix(R).as_rotvec()).as_matrix()
        traj.append(chunked.compose(*traj[-1], R, 0.1 * t))
    bounds = chunked.chunk_bounds(len(traj), 4)
    local = []
    for s, e in bounds:
        R0, t0 = traj[s]
        local.append([(R0.T @ R, R0.T @ (t - t0)) for R, t in traj[s:e + 1]])
    starts = capi.shard_prefix_starts([ch[-1] for ch in local])
    out = []
    for g, ch in enumerate(local):
        reb = capi.shard_rebase(starts[g], ch)
        out.extend(reb if g == 0 else reb[1:])
    assert len(out) == len(traj)
    for (Ra, ta), (Rb, tb) in zip(out, traj):
        assert np.abs(Ra - Rb).max() < 1e-12 and np.abs(ta - tb).max() < 1e-11
