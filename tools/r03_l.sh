#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in noinv nofac; do
echo "== $v"; SVO_LIB=$PWD/ab/lib_pg_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pgp_$v -o pg -- python3 tools/pg_profile.py 4541 40 1 > /dev/null 2>&1
python3 - $(find /tmp/pgp_$v -name "pg_kernel_stats.csv") <<'PY'
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if 'potf2' in r[0]: print('potf2 calls', r[1], 'avg ns', r[3])
PY
done
