#!/usr/bin/env python3
"""rocprofv3 target: the vocabulary-mode detector over N DISTINCT frames of the benchmark stream given by their features
(svo_lc_submit_features), so that the kernel statistics show what a frame costs on the device as the database grows.
    rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/loopdet_profile.py 1400"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ros_stereo_slam_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1400
ctx = capi.Context(0)
poses = synth.loop_trajectory(n, **synth.BENCH_LOOP)
lefts, _ = synth.stereo_torch(synth.bench_scene(), poses, device="cuda", batch=8)
torch.cuda.synchronize()
feats = [ctx.orb_extract(im, 500, 20) for im in lefts]
voc = capi.Vocabulary.train(ctx, [f[4] for f in feats[0:492:4]], k=9, L=6, seed=1)
det = capi.LoopDetector(ctx, 1241, 376, 3, max_entries=n + 8)
det.set_vocabulary(voc, 2)
for lo in range(0, n, 200):
    ctx.sync()
    t0 = time.perf_counter()
    for f in feats[lo:lo + 200]:
        det.submit_features(f[0], f[4])
    k = len(feats[lo:lo + 200])
    v = [det.collect() for _ in range(k)]
    ctx.sync()
    print(f"entries {lo:5d} .. {lo + k:5d}: {(time.perf_counter() - t0) / k * 1e3:.3f} ms per frame, "
          f"{sum(x['status'] == 0 for x in v)} detections", flush=True)
