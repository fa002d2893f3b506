#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r03; mkdir -p $OUT
for r in 1 2; do
timeout -k 10 400 python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 7
python3 -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'events', d['event_records'], 'inits', round(d['value_including_chunk_inits']), 'single', round(d.get('single_chunk_frames_per_s')))"
done
