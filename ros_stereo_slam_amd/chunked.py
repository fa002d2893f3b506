"""Chunk-sharded stereo VO across GPUs (BASELINE.json configs[3]).

The front-end is sequential in time (frame n needs frame n-1's surviving points and pose,
``src/VisualSLAM.cpp:54-169``), so a sequence shards across GPUs only in time: rank g owns
the contiguous frames ``[s_g, s_{g+1}]`` -- INCLUDING the first frame of the next chunk, so
that its last pose is the boundary transform ``T(s_g -> s_{g+1})`` -- re-initialises at
``s_g`` (stereo keyframe, identity pose) and runs the ordinary front-end.  The path's one
exchange step is an all-gather of the boundary transforms (12 doubles per rank, over RCCL
when the process group is ``nccl``); every rank then composes the prefix
``T(0 -> s_g) = prod_{h<g} T(s_h -> s_{h+1})`` and rebases its local poses.  The rebased
trajectories feed one global pose graph (``PoseGraph``) whose loop closures use global
frame ids.

Results differ from the single-GPU run by construction (extra keyframes at chunk starts);
the stated tolerance is on the trajectory (ATE), not per frame.
"""
from __future__ import annotations

import numpy as np


def chunk_bounds(n_frames: int, n_chunks: int) -> list[tuple[int, int]]:
    """Inclusive frame ranges ``[(s_g, e_g)]`` with ``e_g = s_{g+1}`` (one frame of overlap).

    Frames are split as evenly as possible; the last chunk ends at ``n_frames - 1``."""
    if n_chunks < 1 or n_frames < 2:
        raise ValueError("need at least one chunk and two frames")
    n_chunks = min(n_chunks, n_frames - 1)
    steps = n_frames - 1  # number of frame-to-frame transitions
    base, extra = divmod(steps, n_chunks)
    bounds, s = [], 0
    for g in range(n_chunks):
        e = s + base + (1 if g < extra else 0)
        bounds.append((s, e))
        s = e
    return bounds


def compose(Ra, ta, Rb, tb):
    """Pose composition ``T_a * T_b`` for camera-in-world poses (X_w = R X_c + t)."""
    return Ra @ Rb, Ra @ tb + ta


def rebase(local_poses, R0, t0):
    """Re-express chunk-local poses (relative to the chunk's first frame) in the global frame."""
    return [compose(R0, t0, R, t) for (R, t) in local_poses]


def prefix_transforms(boundaries):
    """``boundaries[g] = (R, t)`` of chunk g's last frame in chunk g's own frame.
    Returns the global pose of each chunk's FIRST frame: identity, B0, B0*B1, ..."""
    out = [(np.eye(3), np.zeros(3))]
    for (R, t) in boundaries[:-1]:
        out.append(compose(*out[-1], R, t))
    return out


def all_gather_boundaries(dist, R, t, device=None):
    """The path's one collective: every rank contributes its chunk-boundary pose (12 doubles)
    and receives all of them.  ``dist`` is ``torch.distributed`` (backend nccl == RCCL on the
    GPU node, gloo in the CPU tests)."""
    import torch

    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    mine = torch.tensor(np.r_[np.asarray(R, np.float64).ravel(), np.asarray(t, np.float64).ravel()],
                        dtype=torch.float64, device=dev)
    got = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(got, mine)
    res = []
    for g in got:
        a = g.cpu().numpy()
        res.append((a[:9].reshape(3, 3).copy(), a[9:].copy()))
    return res


def all_gather_chunk_boundaries(dist, pairs, device=None):
    """As :func:`all_gather_boundaries` when every rank runs several chunks side by side
    (``svo_vo_run_chunks``): ``pairs`` = this rank's chunk-boundary poses in chunk order, the
    same number on every rank.  Still ONE all-gather (12 doubles per chunk); returns the
    boundaries of all chunks in global chunk order (rank-major)."""
    import torch

    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    flat = np.concatenate([np.r_[np.asarray(R, np.float64).ravel(), np.asarray(t, np.float64).ravel()]
                           for R, t in pairs])
    mine = torch.tensor(flat, dtype=torch.float64, device=dev)
    got = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(got, mine)
    res = []
    for g in got:
        a = g.cpu().numpy().reshape(-1, 12)
        res.extend((row[:9].reshape(3, 3).copy(), row[9:].copy()) for row in a)
    return res


def gather_trajectories(dist, poses, device=None):
    """All ranks receive every rank's (rebased) pose list; used to build the global pose graph.
    Chunks may differ in length by one frame, so lists are padded to the longest."""
    import torch

    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    n = torch.tensor([len(poses)], dtype=torch.int64, device=dev)
    counts = [torch.empty_like(n) for _ in range(dist.get_world_size())]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    buf = torch.zeros((m, 12), dtype=torch.float64, device=dev)
    for i, (R, t) in enumerate(poses):
        buf[i] = torch.tensor(np.r_[np.asarray(R).ravel(), np.asarray(t).ravel()], dtype=torch.float64)
    got = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(got, buf)
    out = []
    for c, g in zip(counts, got):
        a = g.cpu().numpy()[:c]
        out.append([(r[:9].reshape(3, 3).copy(), r[9:].copy()) for r in a])
    return out


def stitch(dist, local_poses, device=None):
    """Rank-local poses (relative to the chunk's first frame, one per frame of the chunk
    including the overlap frame) -> the rank's poses in the global frame, plus the global
    trajectory without the duplicated overlap frames (identical on every rank)."""
    rank = dist.get_rank()
    boundaries = all_gather_boundaries(dist, *local_poses[-1], device=device)
    starts = prefix_transforms(boundaries)
    mine = rebase(local_poses, *starts[rank])
    chunks = gather_trajectories(dist, mine, device=device)
    traj = list(chunks[0])
    for ch in chunks[1:]:
        traj.extend(ch[1:])  # the first frame of a chunk is the last frame of the previous one
    return mine, traj


def run_chunk(vo, frames):
    """Runs the front-end over one chunk.  ``frames``: iterable of (left, right) images (host
    arrays or device tensors); the first one seeds the chunk (identity pose).  Returns the list
    of chunk-local (R, t), one per frame, and per-frame stats."""
    it = iter(frames)
    left, right = next(it)
    vo.init(left, right)
    poses = [(np.eye(3), np.zeros(3))]
    stats = []
    for left, right in it:
        rc, R, t, ninl, kf, ntrk = vo.track(left, right)
        if rc:
            raise RuntimeError(f"tracking lost in chunk at local frame {len(poses)}")
        poses.append((R.copy(), t.copy()))
        stats.append((ninl, kf, ntrk))
    return poses, stats


def pose7(R, t):
    """(R, t) -> tx ty tz qx qy qz qw (the pose-graph layout)."""
    R = np.asarray(R, np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    q = np.array(q)
    if q[3] < 0:
        q = -q
    return np.r_[np.asarray(t, np.float64), q / np.linalg.norm(q)]


def ate_rmse(est_t, gt_t):
    """Absolute trajectory error (RMSE of the translation difference, no alignment: both
    trajectories start at the identity pose of frame 0)."""
    d = np.asarray(est_t, np.float64) - np.asarray(gt_t, np.float64)
    return float(np.sqrt(np.mean(np.sum(d * d, axis=1))))
