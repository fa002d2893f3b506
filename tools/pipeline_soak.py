#!/usr/bin/env python3
"""Soak of the four-stream one-chunk pipeline: the same stream through svo_vo_run_chunk(pipeline = 1) and (pipeline = 0),
in pieces of random lengths, many frames -- every pose, count and keyframe decision must be equal bit for bit.  The
pipeline's hazards (buffers and events shared by streams that run two frames ahead) are timing-dependent: a race would
show up here as a rare mismatch.

    python tools/pipeline_soak.py [frames] [rounds] [seed]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ros_stereo_slam_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng(seed)
    poses = synth.loop_trajectory(n + 1, **synth.BENCH_LOOP)
    lefts, rights = synth.stereo_torch(synth.bench_scene(), poses, device="cuda", batch=8)
    torch.cuda.synchronize()
    kw = dict(grid_step=10, anms_keep=4096, keyframe_min_inliers=2000, seed=20261003)
    ctx_a, ctx_b = capi.Context(0), capi.Context(0)
    bad = 0
    for rnd in range(rounds):
        a = capi.VisualOdometry(ctx_a, 1241, 376, 3, **kw)
        b = capi.VisualOdometry(ctx_b, 1241, 376, 3, **kw)
        assert a.init(lefts[0], rights[0]) == b.init(lefts[0], rights[0])
        at, kfs = 1, 0
        while at <= n:
            m = int(min(n + 1 - at, rng.integers(1, 90)))
            ra = a.run_chunk(list(lefts[at:at + m]), list(rights[at:at + m]), pipeline=True)
            rb = b.run_chunk(list(lefts[at:at + m]), list(rights[at:at + m]), pipeline=False)
            assert ra[0] == rb[0] == 0 and ra[1] == rb[1] == m, (ra[0], rb[0], ra[1], rb[1], at)
            for k in range(2, 7):
                if not np.array_equal(ra[k], rb[k]):
                    bad += 1
                    i = int(np.argwhere(np.asarray(ra[k]).reshape(m, -1) != np.asarray(rb[k]).reshape(m, -1))[0][0])
                    print(f"round {rnd}: output {k} differs first at frame {at + i} (piece of {m} from {at})", flush=True)
                    break
            kfs += int(ra[6].sum())
            at += m
        a2, a3 = a.reference()
        b2, b3 = b.reference()
        if not (np.array_equal(a2, b2) and np.array_equal(a3, b3)):
            bad += 1
            print(f"round {rnd}: the final reference sets differ", flush=True)
        print(f"round {rnd}: {n} frames, {kfs} keyframes, mismatches so far {bad}", flush=True)
        a.close()
        b.close()
    print("SOAK", "FAILED" if bad else "OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
