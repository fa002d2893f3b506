#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
for v in base sw3 noscore nogn; do
[ $v = base ] && unset SVO_LIB || export SVO_LIB=$PWD/ab/lib_pnp_$v.so
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$v -o p -- python3 bench.py --chunks-per-gpu 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > /dev/null 2>&1
python3 - $(find /tmp/pp_$v -name "p_kernel_stats.csv") $v <<'PY'
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if 'pnp_solve' in r[0]: print(sys.argv[2], 'pnp_solve calls', r[1], 'avg us %.1f'%(float(r[3])/1e3), 'max us %.1f'%(float(r[6])/1e3))
PY
done
