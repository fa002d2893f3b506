#!/bin/bash
# A/B of bench.py FLAG sets on one GPU box with the current library:  tools/ab_flags.sh ROUNDS "flags A" "flags B" ...
cd "$GRAFT_REPO_ROOT" || exit 1
ROUNDS=$1; shift
mkdir -p gpurun_out/abf
for r in $(seq 1 $ROUNDS); do
  i=0
  for f in "$@"; do
    i=$((i+1))
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $f > gpurun_out/abf/${i}_$r.json 2> gpurun_out/abf/${i}_$r.err || { echo "[$f] round $r failed"; tail -3 gpurun_out/abf/${i}_$r.err; exit 1; }
    python3 -c "
import json
d=json.loads(open('gpurun_out/abf/${i}_$r.json').read().strip().splitlines()[-1])
print('[$f]', $r, round(d['value']), flush=True)"
  done
done
