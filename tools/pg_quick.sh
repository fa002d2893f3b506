#!/bin/bash
# pose-graph parity tests, the timed solve at BASELINE configs[2]'s size, and the per-kernel table of one solve
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_posegraph.py -m gpu -q -x > gpurun_out/pg_quick_tests.log 2>&1; rc=$?
tail -3 gpurun_out/pg_quick_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/pg_profile.py ${1:-4541} ${2:-40} 3 || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pgp -o pg -- python3 tools/pg_profile.py ${1:-4541} ${2:-40} 1 > /dev/null 2>&1 || exit 2
python tools/kstats.py $(find /tmp/pgp -name "pg_kernel_stats.csv") pg_
cp $(find /tmp/pgp -name "pg_kernel_stats.csv") gpurun_out/pg_kernel_stats.csv
