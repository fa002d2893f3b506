#!/bin/bash
# kernel trace of ONE chunk per GPU (the north-star's partitioning): tools/timeline.py reads it
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/${1:-r03e}; mkdir -p $OUT
keep() { head -1 "$1" > "$2"; grep -v "at::\|elementwise\|vectorized\|Memcpy\|rocprim\|hipcub\|fillBuffer" "$1" | tail -n +2 >> "$2"; }
timeout -k 10 300 python3 bench.py --chunks-per-gpu 1 --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/one_plain.json 2> $OUT/one_plain.err || { echo "plain failed"; tail -5 $OUT/one_plain.err; exit 2; }
python3 -c "
import json; d=json.loads(open('$OUT/one_plain.json').read().strip().splitlines()[-1]); print('one chunk, unprofiled:', round(d['value']), 'frames/s')"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/r03_one -o one -- python3 bench.py --chunks-per-gpu 1 --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/one_bench.json 2> $OUT/one.err || { echo "trace failed"; tail -5 $OUT/one.err; exit 2; }
keep $(find /tmp/r03_one -name "one_kernel_trace.csv") $OUT/one_trace.csv
python3 -c "
import json; d=json.loads(open('$OUT/one_bench.json').read().strip().splitlines()[-1]); print('one chunk, under the profiler:', round(d['value']), 'frames/s')"
