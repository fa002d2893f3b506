#!/bin/bash
# The round-4 profiling passes (run on the GPU box through gpurun):
#   1  kernel trace + stats of the bench configuration (4096 keypoints)
#   2  SQ counters, FETCH_SIZE, WRITE_SIZE of the same at 4096 and at 8192 keypoints (separate --pmc passes)
#   3  kernel trace of ONE chunk per GPU (the north-star's partitioning): timeline + gap histogram
#   4  the pose graph alone (4541 vertices, 40 closures): per-kernel stats
#   5  microbenchmarks: VALU issue rate of LK's instruction mix, dependent-launch boundary cost
# rocprofv3 output goes to /tmp (the traces of the torch renderer are hundreds of MB); only the rows of this library's
# kernels come back under gpurun_out/r04/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the profiler brings the GPU up before Python runs: bench.py's own setdefault would come too late
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r04
mkdir -p $OUT
echo "{\"GPU_MAX_HW_QUEUES\": \"$GPU_MAX_HW_QUEUES\", \"lk_hip_sha256\": \"$(sha256sum ros_stereo_slam_amd/csrc/lk.hip | cut -d' ' -f1)\"}" > $OUT/environment.json
keep() { head -1 "$1" > "$2"; grep -v "at::\|elementwise\|vectorized\|Memcpy\|rocprim\|hipcub\|fillBuffer" "$1" | tail -n +2 >> "$2"; }
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing --min-timed-s 0"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_trace -o bench -- python3 bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err || exit 1
keep $(find /tmp/r04_trace -name "bench_kernel_stats.csv") $OUT/kernel_stats.csv
keep $(find /tmp/r04_trace -name "bench_kernel_trace.csv") $OUT/kernel_trace.csv
[ "$1" = trace ] && exit 0
for K in ${KPTS:-4096 8192}; do
PARGS="--steps 6 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing --min-timed-s 0 --kpts $K"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/r04_sq_$K -o sq -- python3 bench.py $PARGS > $OUT/sq_bench_$K.json 2> $OUT/sq_$K.err || exit 2
keep $(find /tmp/r04_sq_$K -name "sq_counter_collection.csv") $OUT/sq_counter_collection_$K.csv
keep $(find /tmp/r04_sq_$K -name "sq_kernel_trace.csv") $OUT/sq_kernel_trace_$K.csv
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/r04_fetch_$K -o fetch -- python3 bench.py $PARGS > $OUT/fetch_bench_$K.json 2> $OUT/fetch_$K.err || exit 3
keep $(find /tmp/r04_fetch_$K -name "fetch_counter_collection.csv") $OUT/fetch_counter_collection_$K.csv
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/r04_write_$K -o write -- python3 bench.py $PARGS > $OUT/write_bench_$K.json 2> $OUT/write_$K.err || exit 4
keep $(find /tmp/r04_write_$K -name "write_counter_collection.csv") $OUT/write_counter_collection_$K.csv
done
[ "$1" = all ] || { du -sh $OUT; exit 0; }
timeout -k 10 300 python3 bench.py --chunks-per-gpu 1 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/one_plain.json 2> $OUT/one_plain.err || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/r04_one -o one -- python3 bench.py --chunks-per-gpu 1 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/one_bench.json 2> $OUT/one.err || exit 6
keep $(find /tmp/r04_one -name "one_kernel_trace.csv") $OUT/one_trace.csv
# the one-chunk stages from device time stamps (no profiler in the way: its interception makes the host the bottleneck of this run)
SVO_CHAIN_STAMPS=1 timeout -k 10 300 python3 bench.py --chunks-per-gpu 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/one_stamps.json 2> $OUT/one_stamps.err || exit 11
grep "svo chain" $OUT/one_stamps.err > $OUT/one_stamps.txt
timeout -k 10 300 python3 tools/pg_profile.py 4541 40 3 > $OUT/pg_plain.log 2>&1 || exit 7
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_pg -o pg -- python3 tools/pg_profile.py 4541 40 1 > $OUT/pg_prof.log 2>&1 || exit 8
grep "pg_\|Name" $(find /tmp/r04_pg -name "pg_kernel_stats.csv") > $OUT/pg_kernel_stats.csv
[ -x tools/valu_rate ] || hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
timeout -k 5 120 tools/valu_rate > $OUT/valu_rate.jsonl 2>&1 || exit 9
[ -x tools/launch_gap ] || hipcc --offload-arch=gfx950 -O3 tools/launch_gap.hip -o tools/launch_gap
timeout -k 5 120 tools/launch_gap > $OUT/launch_gap.jsonl 2>&1 || exit 10
du -sh $OUT
