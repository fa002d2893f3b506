#!/bin/bash
# A/B of environment settings on ONE GPU box: tools/ab_env.sh ROUNDS "VAR=a" "VAR=b" ...  (BENCH_FLAGS as in ab.sh)
# BENCH_KEY: the key of the JSON line to print (default value; e.g. single_chunk_frames_per_s)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
ROUNDS=$1; shift
mkdir -p gpurun_out/abenv
for r in $(seq 1 "$ROUNDS"); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    env $v timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras ${BENCH_FLAGS:-} > gpurun_out/abenv/${i}_$r.json 2> gpurun_out/abenv/${i}_$r.err || { echo "$v round $r failed"; tail -5 gpurun_out/abenv/${i}_$r.err; exit 1; }
    python3 -c "
import json
d=json.loads(open('gpurun_out/abenv/${i}_$r.json').read().strip().splitlines()[-1])
print('$v', $r, round(d['${BENCH_KEY:-value}']), flush=True)" || exit 1
  done
done
