#!/bin/bash
# The round-2 profiling passes of the default bench (run on the GPU box through gpurun):
#   kernel trace + stats, SQ counters, FETCH_SIZE, WRITE_SIZE (the two cannot share a pass), VALU-rate microbenchmark.
# rocprofv3 output goes to /tmp (the traces of the torch renderer are hundreds of MB); only the rows of this
# library's kernels come back under gpurun_out/r02/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the profiler brings the GPU up before Python runs: bench.py's own setdefault would come too late
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r02
mkdir -p $OUT
keep() { head -1 "$1" > "$2"; grep -v "at::\|elementwise\|vectorized\|Memcpy\|rocprim\|hipcub" "$1" | tail -n +2 >> "$2"; }
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r02_trace -o bench -- python3 bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err || exit 1
cp /tmp/r02_trace/bench_kernel_stats.csv $OUT/kernel_stats_all.csv
keep /tmp/r02_trace/bench_kernel_trace.csv $OUT/kernel_trace.csv
[ "$1" = trace ] && exit 0   # `profile_r02.sh trace`: the kernel trace only
PARGS="--steps 6 --warmup 1 --no-cpu-baseline --no-extras"
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/r02_sq -o sq -- python3 bench.py $PARGS > $OUT/sq_bench.json 2> $OUT/sq.err || exit 2
keep /tmp/r02_sq/sq_counter_collection.csv $OUT/sq_counter_collection.csv
keep /tmp/r02_sq/sq_kernel_trace.csv $OUT/sq_kernel_trace.csv
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/r02_fetch -o fetch -- python3 bench.py $PARGS > $OUT/fetch_bench.json 2> $OUT/fetch.err || exit 3
keep /tmp/r02_fetch/fetch_counter_collection.csv $OUT/fetch_counter_collection.csv
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/r02_write -o write -- python3 bench.py $PARGS > $OUT/write_bench.json 2> $OUT/write.err || exit 4
keep /tmp/r02_write/write_counter_collection.csv $OUT/write_counter_collection.csv
timeout -k 5 120 tools/valu_rate > $OUT/valu_rate.jsonl 2>&1 || exit 5
du -sh $OUT
