#!/usr/bin/env python3
"""bench.py -- stereo frames/s of the MI355X front-end on a synthetic 1241x376 stream.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on
rank 0.  A step is one pass of the hot path over one frame (inputs already resident in
HBM): pyramid of the new left image + pyramidal LK of 4096 keypoints from the previous
frame (+ the stages the front-end has grown, see ``config.stages``).  For N > 1 the driver
launches one rank per GPU (torch.distributed, backend nccl == RCCL); every rank processes
its own contiguous chunk of the stream (weak scaling, no data-path collective; the only
exchange is the all-gather of chunk-boundary poses).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, C = 1241, 376, 3
N_KPTS = 4096
PYR_BYTES = 619930 * C  # sum of the 4 level sizes


def lk_algorithmic_bytes(n_pts: int) -> int:
    """HBM bytes the LK kernel must move per launch: both pyramids once + 8 B in / 13 B out
    per point (its share of SURVEY 8d's B_track = 3*pyr + 71*N)."""
    return 2 * PYR_BYTES + 21 * n_pts


def frame_algorithmic_bytes(n_pts: int) -> int:
    return 3 * PYR_BYTES + 71 * n_pts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=6, help="distinct synthetic frames kept in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=12)
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from ros_stereo_slam_amd import capi, synth

    ctx = capi.Context(local_rank)
    scene = synth.Scene()
    # every rank renders its own chunk of the stream (contiguous frames, offset by rank)
    poses = synth.corridor_trajectory(args.frames * world)[rank * args.frames:(rank + 1) * args.frames]
    host_frames = [scene.render(R, t)[0] for (R, t) in poses]
    dev_frames = [torch.from_numpy(f).cuda() for f in host_frames]

    grid = ctx.grid_keypoints(H, W, 10)  # 4428 lattice points
    pts_host = np.ascontiguousarray(grid[:N_KPTS])
    n = pts_host.shape[0]
    d_pts = torch.from_numpy(pts_host).cuda()
    d_out = torch.empty_like(d_pts)
    d_status = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_err = torch.empty(n, dtype=torch.float32, device="cuda")
    pyr = [ctx.pyramid(W, H, C), ctx.pyramid(W, H, C)]
    pyr[0].build(dev_frames[0], capi.MEM_DEVICE)
    torch.cuda.synchronize()
    ctx.sync()

    def step(i: int):
        cur, prev = pyr[(i + 1) & 1], pyr[i & 1]
        cur.build(dev_frames[(i + 1) % args.frames], capi.MEM_DEVICE)
        ctx.lk_track_device(prev, cur, d_pts, n, d_out, d_status, d_err)

    for i in range(args.warmup):
        step(i)
    ctx.sync()
    ctx.enable_kernel_timing(True)
    ctx.reset_kernel_time()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    ctx.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    lk_ms, lk_launches = ctx.kernel_time(capi.K_LK)
    ctx.enable_kernel_timing(False)

    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    tracked = int(d_status.sum().item())
    result = None
    if rank == 0:
        fps = world * args.steps / elapsed
        lk_avg_s = lk_ms / max(lk_launches, 1) * 1e-3
        achieved = lk_algorithmic_bytes(n) / lk_avg_s / 1e9 if lk_avg_s > 0 else 0.0
        result = {
            "metric": "stereo frames/sec @1241x376, 4096 kpts",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/i32 fixed-point + f32",
            "data": "synthetic",
            "config": {
                "workload": "synthetic corridor 1241x376x3 stream, 4096 grid keypoints/frame, front-end "
                            "(BASELINE configs[1] shape)",
                "stages": ["pyramid", "lk_t-1_to_t"],
                "parallelism": f"chunk-per-gpu x{world}",
                "tracked_last_frame": tracked,
            },
            "roofline": {
                "kernel": "lk_track_kernel<3>",
                "bound": "hbm",
                "achieved": achieved,
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": achieved / 8000.0,
                "traffic": None,
                "avg_launch_us": lk_avg_s * 1e6,
                "algorithmic_bytes_per_launch": lk_algorithmic_bytes(n),
                "frame_hbm_frac": frame_algorithmic_bytes(n) / (elapsed / args.steps) / 8e12,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import orc  # the checker, timed as the CPU baseline ("port")

            t0 = time.perf_counter()
            for i in range(args.cpu_frames):
                orc.lk_track(host_frames[i % args.frames], host_frames[(i + 1) % args.frames], pts_host)
            dt = time.perf_counter() - t0
            result["cpu_baseline"] = {
                "value": args.cpu_frames / dt,
                "unit": "frames/s",
                "cores": 1,
                "kind": "port",
                "sample": f"{args.cpu_frames} frames of the same stream, same stages, oracle C -O2, 1 thread",
            }
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
