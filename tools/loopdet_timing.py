#!/usr/bin/env python3
"""Per-frame time of the GPU loop detector at 1241x376x3 as its database grows -- the vocabulary-free similarity of
rounds 2-3 (every database entry is compared: O(N_db)) against the vocabulary mode of round 4 (DBoW2's inverted file:
the Hamming work per frame is the tree descent of 500 features, whatever the database holds).  Frames are QUEUED in
windows (svo_lc_submit, then svo_lc_collect), so the figure is device throughput, not a host round trip per frame.
Not part of bench.py (the loop detector is outside BASELINE's metric); numbers quoted in DESIGN.md come from here:

    python tools/loopdet_timing.py [--frames 4400]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4400)
    args = ap.parse_args()
    import torch
    from ros_stereo_slam_amd import capi, synth

    ctx = capi.Context(0)
    scene = synth.bench_scene()
    poses = synth.loop_trajectory(128, **synth.BENCH_LOOP)
    lefts, _ = synth.stereo_torch(scene, poses, device="cuda", batch=8)
    torch.cuda.synchronize()
    feats = [ctx.orb_extract(im, 500, 20) for im in lefts]
    voc = capi.Vocabulary.train(ctx, [f[4] for f in feats[::2]], k=9, L=6, seed=1)
    print(f"vocabulary: k 9, L 6, {voc.n_nodes} nodes, {voc.n_words} words")
    marks_at = [m for m in (100, 500, 1000, 2000, 4000, args.frames) if m <= args.frames]
    for mode in ("vocabulary-free similarity (rounds 2-3)", "vocabulary: inverted file + L1 score (round 4)"):
        det = capi.LoopDetector(ctx, 1241, 376, 3, max_entries=args.frames + 8)
        if mode.startswith("vocabulary:"):
            det.set_vocabulary(voc, 2)
        done, out = 0, []
        for m in marks_at:
            window = min(200, m - done)
            # fill up to the window's start without timing, then time `window` frames queued back to back
            while done < m - window:
                det.submit(lefts[done % len(lefts)])
                det.collect()
                done += 1
            ctx.sync()
            t0 = time.perf_counter()
            for i in range(window):
                det.submit(lefts[(done + i) % len(lefts)])
            for i in range(window):
                det.collect()
            ctx.sync()
            out.append((m, (time.perf_counter() - t0) / window * 1e3))
            done += window
        print(mode)
        for m, ms in out:
            print(f"   database at {m:5d} entries: {ms:7.3f} ms per frame (features + scoring + bookkeeping, queued)")
        det.close()


if __name__ == "__main__":
    main()
