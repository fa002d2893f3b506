#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/${1:-r03h}; mkdir -p $OUT
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q --durations=6 > $OUT/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 $OUT/gpu_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench failed"; tail -5 $OUT/bench_default.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'single', d.get('single_chunk_frames_per_s'), 'collective', d['config']['collective'])
print(d['posegraph'])"
