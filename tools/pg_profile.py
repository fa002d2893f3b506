#!/usr/bin/env python3
"""The pose graph of BASELINE configs[2]/[3] alone: V vertices on the benchmark loop with drift, C identity closures,
10 Gauss-Newton iterations, timed (HIP events inside the library) -- under rocprofv3 --kernel-trace this is the
per-kernel breakdown of a solve.    python tools/pg_profile.py [V] [C] [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.is_available()
from ros_stereo_slam_amd import capi, chunked, synth

V = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
Cn = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 3
poses = synth.loop_trajectory(V, **synth.BENCH_LOOP)
R0, t0 = poses[0]
rng = np.random.default_rng(1)
traj, drift = [], np.zeros(3)
for R, t in poses:
    drift = drift + rng.normal(0, 0.002, 3)
    traj.append((R0.T @ R, R0.T @ (t - t0) + drift))
matches = synth.loop_closures(poses, max_dist=0.3, max_angle_deg=10.0, min_gap=100, pick="nearest")
closures = chunked.gate_closures([m if m >= 1 else -1 for m in matches])
closures = dict(list(closures.items())[:Cn])
ctx = capi.Context(0)
for r in range(rep):
    pg = capi.PoseGraph(ctx)
    ctx.enable_kernel_timing(True)
    ctx.reset_kernel_time()
    t0w = time.perf_counter()
    est, chi2 = chunked.global_solve(pg, traj, closures, iters=10)
    wall = time.perf_counter() - t0w
    ms, _ = ctx.kernel_time(capi.K_POSEGRAPH)
    print(f"{V} vertices, {len(closures)} closures: {ms / 10:.3f} ms per GN iteration (events), wall incl. graph build "
          f"{wall * 1e3:.1f} ms, chi2 {chi2[0]:.4g} -> {chi2[-1]:.4g}", flush=True)
    pg.close()
