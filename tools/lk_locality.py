#!/usr/bin/env python3
"""Does the order of the keypoints (L2 locality) cost the tracking kernel anything?  (VERDICT r3 #7.)

The tracking launch deals keypoints to the eight XCDs as contiguous eighths of the point list ("bands"), which is a spatial
band only for the LATTICE order of the stereo pass; the tracked sets of the temporal passes are in ANMS order (response-
sorted: spatially random).  This tool runs svo_lk_track on ONE image pair of the benchmark stream with the same 4428 lattice
points in three orders -- raster (bands = image bands), column-major strips (each XCD a 155-pixel-wide vertical strip: the
smallest halo), shuffled (every XCD reads the whole image) -- REPS launches each, and prints the mean launch time
(HIP events).  Under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` the per-launch fetch counter of the three groups of
launches can be read from the trace (launch i belongs to order i // REPS):

    python tools/lk_locality.py [REPS]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ros_stereo_slam_amd import capi, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = capi.Context(0)
scene = synth.bench_scene()
poses = synth.loop_trajectory(2, **synth.BENCH_LOOP)
lefts, rights = synth.stereo_torch(scene, poses, device="cuda", batch=2)
torch.cuda.synchronize()
a, b = lefts[0].cpu().numpy(), lefts[1].cpu().numpy()
pa, pb = ctx.pyramid(1241, 376, 3).build(a), ctx.pyramid(1241, 376, 3).build(b)
pts = ctx.grid_keypoints(376, 1241, 10)
n = len(pts)
rng = np.random.default_rng(1)
orders = {
    "raster (XCD = image band)": np.arange(n),
    "column-major (XCD = vertical strip)": np.lexsort((pts[:, 1], pts[:, 0])),
    "shuffled (as a tracked set in ANMS order)": rng.permutation(n),
}
ref = None
for name, perm in orders.items():
    d_in = torch.from_numpy(np.ascontiguousarray(pts[perm])).cuda()
    d_out = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
    d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.lk_track_device(pa, pb, d_in, n, d_out, d_st)          # warm
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.lk_track_device(pa, pb, d_in, n, d_out, d_st)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    out = d_out.cpu().numpy()
    inv = np.empty(n, int)
    inv[perm] = np.arange(n)
    if ref is None:
        ref = out[inv]
    assert np.array_equal(out[inv], ref), "the order of the points must not change a result"
    print(f"{name:45s} {dt * 1e6:8.1f} us per launch of {n} keypoints ({reps} launches back to back)", flush=True)
