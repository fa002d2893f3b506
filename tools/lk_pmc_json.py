#!/usr/bin/env python3
"""Writes profiles/rNN_lk_pmc_<kpts>.json (r02: profiles/r02_lk_pmc.json): the per-pass counter figures of lk_track_kernel<3> that bench.py's
roofline object quotes, from the committed per-kernel summaries of the rocprofv3 --pmc passes
(tools/pmc_summary.py), the bench line those passes printed (passes per tracking launch in the timed region)
and the VALU-rate microbenchmark (tools/valu_rate.hip).  The file names the sha256 of lk.hip: bench.py
refuses the figures for any other build.

    python tools/lk_pmc_json.py --sq profiles/r02_pmc_sq.csv --hbm profiles/r02_pmc_hbm_traffic.csv \\
        --bench-line gpurun_out/pmc_sq_bench.json --valu-rate profiles/r02_valu_rate.jsonl \\
        --kernel-stats profiles/r02_pmc_sq_kernel_durations.csv --kpts 4096 > profiles/r02_lk_pmc.json
"""
import argparse
import csv
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def row_of(path, kernel):
    for r in csv.DictReader(open(path)):
        if r["kernel"].startswith(kernel):
            return r
    raise SystemExit(f"{path}: no row for {kernel}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sq", required=True)
    ap.add_argument("--hbm", required=True)
    ap.add_argument("--bench-line", required=True, help="the JSON line bench.py printed under the SQ pass")
    ap.add_argument("--valu-rate", required=True)
    ap.add_argument("--lk-alone-us", type=float, required=True,
                    help="mean duration of a tracking launch in the SQ pass (kernels run one at a time there)")
    ap.add_argument("--kpts", type=int, default=4096)
    ap.add_argument("--lk-launches", type=int, default=0,
                    help="tracking launches of the timed region (the dispatches the --pmc summaries average over), when the "
                         "bench line was printed with --no-kernel-timing")
    ap.add_argument("--source", default=None, help="text of the `source` field (the committed files the figures come from)")
    a = ap.parse_args()
    sq = row_of(a.sq, "lk_track_kernel<3")   # <3> (rounds 1-2) or <3, 16> (jobs per launch as a template parameter)
    hbm = row_of(a.hbm, "lk_track_kernel<3")
    bench = json.loads(open(a.bench_line).read().strip().splitlines()[-1])
    passes = bench["roofline"]["lk_passes_per_launch"]
    if a.lk_launches:   # a bench line printed without its instrumented pass carries the passes of the whole region
        passes = bench["roofline"]["lk_passes_per_launch"] * bench["roofline"].get("launches_per_step", 0) * bench["steps"] / a.lk_launches \
            if bench["roofline"].get("launches_per_step") else bench["roofline"]["lk_passes_per_launch"] / a.lk_launches
    rates = [json.loads(l) for l in open(a.valu_rate) if l.startswith("{") and '"mix"' in l]
    mix4 = next(r for r in rates if r["mix"].startswith("lk_mix") and r["waves_per_simd"] == 4)
    fma4 = next(r for r in rates if r["mix"] == "v_fma_f32" and r["waves_per_simd"] == 4)
    valu = float(sq["SQ_INSTS_VALU_per_dispatch"])
    busy = float(sq["SQ_BUSY_CYCLES_per_dispatch"])
    # SQ_BUSY_CYCLES sums over the 32 shader engines' SQs: / 32 = cycles the launch was resident
    sclk = busy / 32.0 / (a.lk_alone_us * 1e-6)
    # the microbenchmark reports wall time per wave-instruction and SIMD; as cycles of the clock held in the
    # tracking launch itself
    ns_per_inst = mix4["kernel_ms"] * 1e6 / (mix4["insts_per_wave"] * mix4["waves_per_simd"])
    out = {
        "kernel": sq["kernel"],
        "kpts": a.kpts,
        "lk_hip_sha256": hashlib.sha256(open(os.path.join(ROOT, "ros_stereo_slam_amd", "csrc", "lk.hip"), "rb").read()).hexdigest(),
        "passes_per_launch_in_the_pmc_runs": passes,
        "valu_insts_per_pass": valu / passes,
        "hbm_bytes_per_pass": float(hbm["hbm_bytes_per_dispatch_2F_plus_W"]) / passes,
        "fetch_kb_per_launch": float(hbm["FETCH_SIZE_KB_per_dispatch"]),
        "write_kb_per_launch": float(hbm["WRITE_SIZE_KB_per_dispatch"]),
        "sclk_hz_under_load": sclk,
        "valu_ns_per_wave_inst_measured": ns_per_inst,
        "valu_cycles_per_wave_inst": ns_per_inst * 1e-9 * sclk,
        "valu_cycles_per_wave_inst_v_fma_f32": fma4["kernel_ms"] * 1e6 / (fma4["insts_per_wave"] * 4) * 1e-9 * sclk,
        "lk_launch_alone_us": a.lk_alone_us,
        "valu_issue_share_of_a_launch_alone": valu * ns_per_inst * 1e-9 / 1024.0 / (a.lk_alone_us * 1e-6),
        "source": a.source or (f"{a.hbm}, {a.sq} (rocprofv3 --pmc, timed region, (2*FETCH_SIZE+WRITE_SIZE)*1024 per "
                               f"tracking pass), {a.valu_rate}"),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
