#!/usr/bin/env python3
"""Per-frame time of the GPU loop detector (svo_lc_detect) at 1241x376x3 as its database grows.
Not part of bench.py (the loop detector is
outside BASELINE's metric); numbers quoted in DESIGN.md come from this script:

    python tools/loopdet_timing.py [--frames 2000]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2000)
    args = ap.parse_args()
    import torch
    from ros_stereo_slam_amd import capi, synth

    ctx = capi.Context(0)
    sc = synth.Scene()
    poses = synth.corridor_trajectory(8)
    imgs = [sc.stereo(R, t)[0] for R, t in poses]
    dev = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    det = capi.LoopDetector(ctx, 1241, 376, 3, max_entries=args.frames + 8)
    marks = {}
    t_last, n_last = time.perf_counter(), 0
    for i in range(args.frames):
        det.detect(dev[i % len(dev)])
        if (i + 1) in (50, 250, 500, 1000, 2000, 4000, args.frames):
            now = time.perf_counter()
            marks[i + 1] = (now - t_last) / (i + 1 - n_last) * 1e3
            t_last, n_last = now, i + 1
    for k, v in marks.items():
        print(f"database up to {k:5d} entries: {v:7.3f} ms per frame (features + scoring + bookkeeping)")


if __name__ == "__main__":
    main()
