#!/usr/bin/env python3
"""svo_orb_extract on the benchmark stream's left images, timed; under rocprofv3 --kernel-trace --stats: its kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.is_available()
from ros_stereo_slam_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = capi.Context(0)
poses = synth.loop_trajectory(n, **synth.BENCH_LOOP)
lefts, _ = synth.stereo_torch(synth.bench_scene(), poses, device="cuda", batch=8)
torch.cuda.synchronize()
ctx.orb_extract(lefts[0], 500, 20)
t0 = time.perf_counter()
for im in lefts:
    ctx.orb_extract(im, 500, 20)
print("svo_orb_extract: %.3f ms per frame" % ((time.perf_counter() - t0) / len(lefts) * 1e3))
