#!/bin/bash
# A/B of bench.py FLAG SETS on ONE GPU box: tools/ab_flags.sh ROUNDS "flags a" "flags b" ...
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
ROUNDS=$1; shift
mkdir -p gpurun_out/abflags
for r in $(seq 1 "$ROUNDS"); do
  i=0
  for f in "$@"; do
    i=$((i+1))
    timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing $f > gpurun_out/abflags/${i}_$r.json 2> gpurun_out/abflags/${i}_$r.err || { echo "[$f] round $r failed"; tail -5 gpurun_out/abflags/${i}_$r.err; exit 1; }
    python3 -c "
import json
d=json.loads(open('gpurun_out/abflags/${i}_$r.json').read().strip().splitlines()[-1])
print('[$f]', $r, round(d['value']), flush=True)" || exit 1
  done
done
