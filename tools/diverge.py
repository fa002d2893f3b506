#!/usr/bin/env python3
"""GPU front-end against the oracle's frame loop on the benchmark stream, frame by frame: prints every frame whose
counts, decision or pose differ (diagnostic for tests/test_gpu_frontend.py).   python tools/diverge.py [frames] [kpts]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import orc
from ros_stereo_slam_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 101
kp = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
gs, keep, kf = (10, 4096, 2000) if kp == 4096 else (7, 8192, 4000)
poses = synth.loop_trajectory(n, **synth.BENCH_LOOP)
lefts, rights = synth.stereo_torch(synth.bench_scene(), poses, device="cuda", batch=8)
torch.cuda.synchronize()
frames = [(l.cpu().numpy(), r.cpu().numpy()) for l, r in zip(lefts, rights)]
orc.set_num_threads(min(32, os.cpu_count() or 8))
ctx = capi.Context(0)
g = capi.VisualOdometry(ctx, 1241, 376, 3, grid_step=gs, anms_keep=keep, keyframe_min_inliers=kf, seed=20261003)
o = orc.VO(1241, 376, 3, grid_step=gs, anms_keep=keep, keyframe_min_inliers=kf, seed=20261003)
print("init", g.init(*frames[0]), o.init(*frames[0]), flush=True)
worst = 0.0
for i in range(1, n):
    rg, Rg, tg, ig, kg, ng = g.track(*frames[i])
    ro, Ro, to, io, ko, no = o.track(*frames[i])
    dt = float(np.linalg.norm(tg - to))
    dR = float(np.abs(Rg - Ro).max())
    worst = max(worst, dt)
    if (ng, ig, kg) != (no, io, ko) or dt > 0 or dR > 0 or i % 20 == 0:
        print(f"frame {i}: tracked {ng}/{no} inliers {ig}/{io} kf {kg}/{ko} dt {dt:.3e} dR {dR:.3e}", flush=True)
print("worst dt", worst)
