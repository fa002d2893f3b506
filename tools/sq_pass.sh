#!/bin/bash
# SQ counter pass only (see profile_r02.sh): per-kernel instruction counts of the default bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the profiler brings the GPU up before Python runs: bench.py's own setdefault would come too late
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/sq; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/sqp -o sq -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras > $OUT/sq_bench.json 2> $OUT/sq.err || exit 2
head -1 /tmp/sqp/*/sq_counter_collection.csv > $OUT/sq_counter_collection.csv 2>/dev/null || head -1 $(find /tmp/sqp -name "sq_counter_collection.csv") > $OUT/sq_counter_collection.csv
grep -v "at::\|elementwise\|vectorized\|Memcpy\|rocprim\|hipcub" $(find /tmp/sqp -name "sq_counter_collection.csv") | tail -n +2 >> $OUT/sq_counter_collection.csv
