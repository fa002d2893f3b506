#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r03; mkdir -p $OUT
keep() { head -1 "$1" > "$2"; grep -v "at::\|elementwise\|vectorized\|Memcpy\|rocprim\|hipcub\|fillBuffer" "$1" | tail -n +2 >> "$2"; }
timeout -k 10 300 python3 bench.py --chunks-per-gpu 1 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/one_plain.json 2> $OUT/one_plain.err || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/r03_one -o one -- python3 bench.py --chunks-per-gpu 1 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/one_bench.json 2> $OUT/one.err || exit 6
keep $(find /tmp/r03_one -name "one_kernel_trace.csv") $OUT/one_trace.csv
timeout -k 10 400 python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 7
python3 -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'single', d.get('single_chunk_frames_per_s'), 'collective', d['config']['collective']); print(json.dumps(d['roofline'])[:1500])"
