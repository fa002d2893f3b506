#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/${1:-r03m}; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_lk.py tests/test_gpu_frontend.py -x -q > $OUT/lk_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $OUT/lk_tests.log
[ $rc -eq 0 ] || exit 1
for r in 1 2; do
for wg in 512 1024 2048 4432; do
SVO_LK_WAVES_GROUP=$wg timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/b_g${wg}_$r.json 2> $OUT/b_g${wg}_$r.err || { echo failed $wg; tail -3 $OUT/b_g${wg}_$r.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$OUT/b_g${wg}_$r.json').read().strip().splitlines()[-1]); print('group waves $wg round $r:', round(d['value']), 'frames/s')"
done
for wl in 2048 4096 8192; do
SVO_LK_WAVES_LONE=$wl timeout -k 10 300 python3 bench.py --chunks-per-gpu 1 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/b_l${wl}_$r.json 2> $OUT/b_l${wl}_$r.err || { echo failed; exit 1; }
python3 -c "
import json; d=json.loads(open('$OUT/b_l${wl}_$r.json').read().strip().splitlines()[-1]); print('lone waves $wl round $r:', round(d['value']), 'frames/s')"
done
done
