#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/${1:-r03d}; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_frontend.py -x -q > $OUT/frontend.log 2>&1; rc=$?; echo "frontend rc=$rc"; tail -25 $OUT/frontend.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_frontend.py > $OUT/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $OUT/gpu_tests.log
[ $rc -eq 0 ] || exit 1
bash tools/r03_trace1.sh ${1:-r03d} || exit 1
timeout -k 10 400 python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench failed"; tail -5 $OUT/bench_default.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'single', d.get('single_chunk_frames_per_s'), 'inits', d.get('value_including_chunk_inits'), 'events', d['event_records']['value_with_event_records'], 'lk_us', d['roofline']['avg_launch_us'], 'ate', d['ate_rmse_vs_truth'])"
