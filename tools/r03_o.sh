#!/bin/bash
# A/B on ONE box: the round-2 tree (ab/r02, a git worktree of f66778b with its own library) against this tree
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r03o; mkdir -p $OUT
for r in 1 2 3; do
( cd ab/r02 && timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > ../../$OUT/old_$r.json 2> ../../$OUT/old_$r.err ) || { echo "old failed"; tail -3 $OUT/old_$r.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/new_$r.json 2> $OUT/new_$r.err || { echo "new failed"; tail -3 $OUT/new_$r.err; exit 1; }
python3 -c "
import json
o=json.loads(open('$OUT/old_$r.json').read().strip().splitlines()[-1]); n=json.loads(open('$OUT/new_$r.json').read().strip().splitlines()[-1])
print('round $r: round-2 tree', round(o['value']), ' this tree', round(n['value']))"
done
