#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r03k}; mkdir -p $OUT
timeout -k 10 300 python3 tools/pg_profile.py 4541 40 3 > $OUT/pg_plain.log 2>&1; cat $OUT/pg_plain.log | grep vertices
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pgp -o pg -- python3 tools/pg_profile.py 4541 40 1 > $OUT/pg_prof.log 2>&1 || { echo "prof failed"; tail -5 $OUT/pg_prof.log; exit 1; }
grep "pg_" $(find /tmp/pgp -name "pg_kernel_stats.csv") | cut -c1-400 > $OUT/pg_kernel_stats.csv; cat $OUT/pg_kernel_stats.csv
python3 - <<'PY'
import csv,re,sys
rows=list(csv.reader(open('gpurun_out/'+(sys.argv[1] if len(sys.argv)>1 else 'r03k')+'/pg_kernel_stats.csv')))
tot=0
for r in rows:
    name=re.sub(r'\(anonymous namespace\)::','',r[0]).split('(')[0]
    tot+=int(r[2]); print(f"{name:26s} calls {r[1]:>5s} total_us {int(r[2])/1e3:9.1f} avg_us {float(r[3])/1e3:8.1f}")
print("sum of kernel time per GN iteration (10 iterations): %.1f us"%(tot/1e4))
PY
timeout -k 10 300 python3 -m pytest tests/test_gpu_posegraph.py tests/test_gpu_properties.py -x -q 2>&1 | tail -3
