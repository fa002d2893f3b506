#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r03g; mkdir -p $OUT
for r in 1 2 3; do for il in 0 1; do
SVO_LK_INTERLEAVE=$il timeout -k 10 300 python3 bench.py --chunks-per-gpu 1 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/one_il${il}_$r.json 2> $OUT/one_il${il}_$r.err || { echo failed; exit 1; }
python3 -c "
import json; d=json.loads(open('$OUT/one_il${il}_$r.json').read().strip().splitlines()[-1]); print('interleave $il round $r:', round(d['value']), 'frames/s')"
done; done
