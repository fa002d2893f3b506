#!/usr/bin/env python3
"""bench.py -- stereo frames/s of the MI355X front-end on a synthetic 1241x376 stream.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on
rank 0.  A step is one pass of the hot path over one stereo frame whose images are already
resident in HBM: pyramid of the new left image, pyramidal LK of the reference keypoints,
F-matrix RANSAC, PnP-RANSAC + refinement, pose composition, keyframe rule and -- when the
rule fires -- the stereo keyframe path (right pyramid, grid + ANMS to 4096 keypoints, LK
left->right, F-RANSAC, DLT triangulation, rigid transform).  That is BASELINE.json
configs[1] ("front-end only, pose-graph off") on the synthetic stream of SURVEY.md 8d.

For N > 1 the driver launches one rank per GPU (torch.distributed, backend nccl == RCCL).
Every rank runs the front-end on its own contiguous chunk of the stream (weak scaling, no
data-path collective); the one exchange step of the path is the all-gather of the
chunk-boundary poses (12 doubles per rank) at the end of the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# One hardware queue per chunk stream, plus room for the default stream (torch's copies) and for
# the streams RCCL creates in a multi-GPU run: with the runtime's default pool of 4 the fourth chunk
# stream shares a queue with another chunk (which queue a stream gets is the runtime's
# least-referenced pick) and the two serialise.  The pool may be larger than needed (5, 6, 8 and 12
# measure the same); what hurts is more than four queues BUSY at once, so the chunk count stays at 4.
# Must be set before the HIP runtime initialises, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, C = 1241, 376, 3
N_KPTS = 4096
GRID_STEP = 10            # 4428 lattice points -> ANMS keeps 4096
KF_MIN_INLIERS = 2000     # the reference's 200-of-440 rule scaled to 4096 keypoints
PYR_BYTES = 619930 * C    # sum of the 4 level sizes
# HBM bytes per lk_track_kernel<3> launch from the PMC counters of this very workload
# (profiles/r01_pmc_hbm_traffic_v8.csv: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes,
# KB units, gfx950 correction for 16-B-per-lane reads: (2*FETCH_SIZE + WRITE_SIZE) * 1024).
# PMC counters cannot be read from inside this process, so the figure is carried from the
# committed profile; None would be the honest value for any other workload.
LK_PMC_TRAFFIC_BYTES = 13144960   # per LK pass: 160.8 MB per launch, 12.23 passes per launch in the timed region of the PMC run
# VALU wave-instructions per lk_track_kernel<3> launch from the SQ counters of the same workload
# (profiles/r01_pmc_sq_v10.csv, SQ_INSTS_VALU: 354.6 M per launch of 12.23 passes).  The kernel's own bound is VALU
# issue, not HBM: 1024 SIMDs x one wave64 VALU instruction per 4 cycles, at 2.4 GHz nominal; the same profile's
# SQ_BUSY_CYCLES against the launch durations give 1.74 GHz under this load.
LK_PMC_VALU_INSTS = 28989000
LK_OBSERVED_SCLK_HZ = 1.74e9


def lk_algorithmic_bytes(n_pts: int) -> int:
    """HBM bytes one LK launch must move: both pyramids once + 8 B in / 13 B out per point
    (the LK share of SURVEY 8d's B_track = 3*pyr + 71*N)."""
    return 2 * PYR_BYTES + 21 * n_pts


def frame_algorithmic_bytes(n_pts: int, keyframe_rate: float) -> float:
    return 3 * PYR_BYTES + 71 * n_pts + keyframe_rate * (3 * PYR_BYTES + 102 * n_pts)


def pingpong(i: int, n: int) -> int:
    """0,1,..,n-1,n-2,..,1,0,1,.. : a temporally continuous walk over n resident frames."""
    p = 2 * (n - 1)
    k = i % p
    return k if k < n else p - k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50,
                    help="timed steps; one step = one frame of every chunk of the GPU (64 frames per GPU with the defaults)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=6, help="distinct synthetic stereo frames kept in HBM per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", choices=("auto", "on", "off"), default="auto",
                    help="two-stream overlap of PnP(t) with pyramid + LK(t+1) inside a chunk; auto = on for one "
                         "chunk per GPU, off when several chunks already fill the hardware queues")
    ap.add_argument("--chunks-per-gpu", type=int, default=int(os.environ.get("SVO_CHUNKS_PER_GPU", "64")),
                    help="independent chunks of the stream run side by side on each GPU (svo_vo_run_chunks); "
                         "with --chunks-per-context 16 that is 4 contexts = 4 busy hardware queues")
    ap.add_argument("--chunks-per-context", type=int, default=int(os.environ.get("SVO_CHUNKS_PER_CONTEXT", "16")),
                    help="chunks that share one context (= one stream): advanced in lock step, every stage of the "
                         "tracking path ONE set of launches for all of them (1..16)")
    ap.add_argument("--kpts", type=int, default=4096, choices=(4096, 8192),
                    help="keypoints per frame: 4096 = BASELINE's metric (grid step 10), 8192 = configs[4] shape "
                         "(grid step 7 -> 9152 lattice points -> ANMS 8192, keyframe rule 4000)")
    ap.add_argument("--cpu-frames", type=int, default=24)
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="diagnostic: no HIP events around the kernels (the roofline object is then empty)")
    args = ap.parse_args()

    import torch

    global N_KPTS, GRID_STEP, KF_MIN_INLIERS
    if args.kpts == 8192:
        N_KPTS, GRID_STEP, KF_MIN_INLIERS = 8192, 7, 4000

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 path on a ONE-GPU box (never used by the driver): all ranks share
    # device 0 and the collectives run over gloo on host tensors.
    rehearsal = os.environ.get("SVO_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from ros_stereo_slam_amd import capi, synth

    M = max(1, args.chunks_per_gpu)
    G = max(1, min(16, args.chunks_per_context))
    pipeline = args.pipeline == "on" or (args.pipeline == "auto" and M == 1)
    scene = synth.Scene()
    # every rank renders its own contiguous chunks of the stream (M per GPU, one context each)
    all_poses = synth.corridor_trajectory(args.frames * world * M)
    ctxs, vos, host_frames, dev_frames = [], [], [], []
    for m in range(M):
        c0 = (rank * M + m) * args.frames
        hf = [scene.stereo(R, t)[:2] for (R, t) in all_poses[c0:c0 + args.frames]]
        host_frames.append(hf)
        dev_frames.append([(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()) for l, r in hf])
        if m % G == 0:
            ctxs.append(capi.Context(local_rank))
        vos.append(capi.VisualOdometry(ctxs[m // G], W, H, C, grid_step=GRID_STEP, anms_keep=N_KPTS,
                                       keyframe_min_inliers=KF_MIN_INLIERS, seed=20261003 + m))
    torch.cuda.synchronize()
    n0 = [vos[m].init(*dev_frames[m][0]) for m in range(M)][0]

    stats = {"keyframes": 0, "inliers": 0, "tracked": 0, "lost": 0}

    def run(first: int, counts, record: bool):
        """Frames first+1 .. first+counts[m] of chunk m's ping-pong walk through the chunk runner
        (one C call, no Python between frames; PnP of frame t overlaps pyramid + LK of frame t+1;
        the M chunks of this GPU run side by side on their own contexts)."""
        last = [None] * M
        done_total = [0] * M
        while any(done_total[m] < counts[m] for m in range(M)):
            active = [m for m in range(M) if done_total[m] < counts[m]]
            idx = {m: [pingpong(first + done_total[m] + k + 1, args.frames) for k in range(counts[m] - done_total[m])]
                   for m in active}
            jobs = [(vos[m], [dev_frames[m][i][0] for i in idx[m]], [dev_frames[m][i][1] for i in idx[m]])
                    for m in active]
            if M == 1:
                res = [vos[0].run_chunk(jobs[0][1], jobs[0][2], pipeline=pipeline)]
            else:
                res = capi.run_chunks(jobs, pipeline=pipeline)
            for m, (rc, done, Rs, ts, inl, trk, kf) in zip(active, res):
                if record:
                    stats["keyframes"] += int(kf[:done].sum())
                    stats["inliers"] += int(inl[:done].sum())
                    stats["tracked"] += int(trk[:done].sum())
                if done:
                    last[m] = (Rs[done - 1], ts[done - 1])
                done_total[m] += done
                if rc:  # tracking lost: re-seed on that frame (the reference would shut down)
                    stats["lost"] += 1
                    vos[m].init(*dev_frames[m][idx[m][done]])
                    done_total[m] += 1
        return last

    def split(total):
        return [total // M + (1 if m < total % M else 0) for m in range(M)]

    def sync_all():
        for c in ctxs:
            c.sync()

    run(0, split(args.warmup * M), False)
    sync_all()
    for c in ctxs:
        c.enable_kernel_timing(not args.no_kernel_timing)
        c.reset_kernel_time()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    sync_all()
    t0 = time.perf_counter()
    # EXACTLY --steps steps; one step = one pass of the front-end over the batch this GPU keeps in flight,
    # i.e. one frame of each of its M chunks (a lock-step group advances all its chunks per set of launches)
    n_frames = args.steps * M
    last = run(args.warmup, split(n_frames), True)
    if dist is not None:
        # the path's one exchange: chunk-boundary poses, 12 doubles per rank, over RCCL
        from ros_stereo_slam_amd import chunked

        pairs = [p if p is not None else (np.eye(3), np.zeros(3)) for p in last]
        boundaries = chunked.all_gather_chunk_boundaries(dist, pairs, device="cpu" if rehearsal else "cuda")
        starts = chunked.prefix_transforms(boundaries)  # global pose of every chunk's first frame
        assert len(starts) == world * M
    sync_all()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    times = {}
    for name, kid in (("pyramid", capi.K_PYRAMID), ("lk", capi.K_LK), ("fransac", capi.K_FRANSAC),
                      ("triangulate", capi.K_TRIANGULATE), ("pnp", capi.K_PNP), ("anms", capi.K_ANMS)):
        per = [c.kernel_time(kid) for c in ctxs]
        times[name] = (sum(p[0] for p in per), sum(p[1] for p in per))
    for c in ctxs:
        c.enable_kernel_timing(False)

    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        fps = world * n_frames / elapsed
        lk_ms, lk_launches = times["lk"]
        lk_avg_s = lk_ms / max(lk_launches, 1) * 1e-3
        # a launch carries the tracking passes of the chunks of one context that are in step
        jobs_per_launch = (n_frames * (1.0 + stats["keyframes"] / max(n_frames, 1))) / max(lk_launches, 1)
        lk_bytes = lk_algorithmic_bytes(N_KPTS) * jobs_per_launch
        achieved = lk_bytes / lk_avg_s / 1e9 if lk_avg_s > 0 else 0.0
        kf_rate = stats["keyframes"] / n_frames
        result = {
            "metric": f"stereo frames/sec @1241x376, {N_KPTS} kpts",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/i32 fixed-point (LK) + f32/f64 (geometry)",
            "data": "synthetic",
            "config": {
                "workload": f"synthetic corridor 1241x376x3 stereo stream, grid step {GRID_STEP} -> ANMS {N_KPTS} keypoints, "
                            "front-end only (BASELINE configs[1]): pyramid + LK + F-RANSAC + PnP-RANSAC + "
                            "keyframe path (LK L->R, F-RANSAC, DLT triangulation)",
                "keyframe_min_inliers": KF_MIN_INLIERS,
                "parallelism": f"{M} contiguous chunk(s) per GPU x{world} GPU(s), all-gather of chunk-boundary poses",
                "chunks_per_gpu": M,
                "chunks_per_context": G,
                "single_chunk_reference": "one chunk alone: 1.85 k frames/s with the two-stream pipeline "
                                          "(--chunks-per-gpu 1), 1.32 k serial (DESIGN.md section 6)",
                "pipeline": "two HIP streams per chunk: PnP(t) beside pyramid+LK(t+1)" if pipeline
                            else "one in-order HIP stream per chunk",
                "keyframe_rate": kf_rate,
                "frames_per_step": M * world,
                "mean_tracked": stats["tracked"] / n_frames,
                "mean_pnp_inliers": stats["inliers"] / n_frames,
                "tracking_lost": stats["lost"],
                "init_points": n0,
                "stage_ms_per_frame": {k: v[0] / n_frames for k, v in times.items()},
            },
            "roofline": {
                "kernel": "lk_track_kernel<3>",
                "bound": "hbm",
                "achieved": achieved,
                "peak": 8000.0,
                "unit": "GB/s",
                "frac": achieved / 8000.0,
                "traffic": LK_PMC_TRAFFIC_BYTES * jobs_per_launch if (W, H, C, N_KPTS) == (1241, 376, 3, 4096) else None,
                "traffic_source": "profiles/r01_pmc_hbm_traffic_v8.csv (rocprofv3 --pmc, (2*FETCH_SIZE+WRITE_SIZE)*1024, "
                                  "per tracking pass) x passes per launch",
                "avg_launch_us": lk_avg_s * 1e6,
                "launches_per_step": lk_launches / args.steps,
                "launches_per_frame": lk_launches / n_frames,
                "algorithmic_bytes_per_launch": lk_bytes,
                "lk_passes_per_launch": jobs_per_launch,
                "valu_issue_bound_us": (LK_PMC_VALU_INSTS * jobs_per_launch * 4 / 1024 / 2.4e9 * 1e6)
                                       if N_KPTS == 4096 else None,
                "valu_issue_frac": (LK_PMC_VALU_INSTS * jobs_per_launch * 4 / 1024 / 2.4e9) / lk_avg_s
                                   if (N_KPTS == 4096 and lk_avg_s > 0) else None,
                # launches of the four contexts overlap, so the chip-level figure is the meaningful one: VALU issue
                # cycles LK asks for per second over what 1024 SIMDs offer (LK issues 83 % of all VALU instructions)
                "valu_chip_frac": (LK_PMC_VALU_INSTS * jobs_per_launch * lk_launches * 4 / 1024 / 2.4e9) / elapsed
                                  if N_KPTS == 4096 else None,
                "valu_chip_frac_at_observed_clock": (LK_PMC_VALU_INSTS * jobs_per_launch * lk_launches * 4 / 1024
                                                     / LK_OBSERVED_SCLK_HZ) / elapsed if N_KPTS == 4096 else None,
                "frame_hbm_frac": frame_algorithmic_bytes(N_KPTS, kf_rate) / (elapsed / n_frames) / 8e12,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import orc  # the checker, timed as the CPU baseline ("port")

            def oracle_rate(threads: int, frames: int) -> float:
                orc.set_num_threads(threads)
                o = orc.VO(W, H, C, grid_step=GRID_STEP, anms_keep=N_KPTS, keyframe_min_inliers=KF_MIN_INLIERS,
                           seed=20261003)
                o.init(*host_frames[0][0])
                c0 = time.perf_counter()
                for i in range(frames):
                    l, r = host_frames[0][pingpong(i + 1, args.frames)]
                    if o.track(l, r)[0]:
                        o.init(l, r)
                dt = time.perf_counter() - c0
                o.close()
                return frames / dt

            cores = min(os.cpu_count() or 1, 16)
            multi = oracle_rate(cores, args.cpu_frames)
            single = oracle_rate(1, max(6, args.cpu_frames // 4))
            result["cpu_baseline"] = {
                "value": multi,
                "unit": "frames/s",
                "cores": cores,
                "kind": "port",
                "sample": f"{args.cpu_frames} frames of the same stream and stages, oracle C (-O2, OpenMP over "
                          f"keypoints in LK and ANMS, RANSAC stages scalar), {cores} threads",
                "single_thread_value": single,
            }
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()
    for v in vos:
        v.close()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
