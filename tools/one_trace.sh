#!/bin/bash
# kernel trace of ONE chunk per GPU -> gpurun_out/one_tl.txt (timeline of a window) and gpurun_out/one_gaps.txt
# usage (on the GPU box): bash tools/one_trace.sh [SVO_LIB path]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
[ -n "$1" ] && export SVO_LIB="$1"
rm -rf /tmp/one
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/one -o one -- python3 bench.py --chunks-per-gpu 1 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/one_bench.json 2> gpurun_out/one.err || exit 1
T=$(find /tmp/one -name one_kernel_trace.csv)
head -1 $T > gpurun_out/one_trace.csv
grep -v "at::\|elementwise\|vectorized\|Memcpy\|rocprim\|hipcub\|fillBuffer" $T | tail -n +2 >> gpurun_out/one_trace.csv
python3 tools/timeline.py gpurun_out/one_trace.csv --gaps > gpurun_out/one_gaps.txt
python3 tools/timeline.py gpurun_out/one_trace.csv 90 ${WINDOW:-70} > gpurun_out/one_tl.txt
python3 -c "
import json;d=json.loads(open('gpurun_out/one_bench.json').read().strip().splitlines()[-1]);print('frames/s under the profiler', round(d['value']))"
