#!/bin/bash
# kernel trace of the bench configuration for a library variant (ab/lib_<name>.so), summarised over the timed region:
#   tools/trace_variant.sh name
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=8
v=$1
rm -rf /tmp/tv_$v
SVO_LIB=$PWD/ab/lib_$v.so timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tv_$v -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing --min-timed-s 0 > gpurun_out/tv_$v.json 2> gpurun_out/tv_$v.err || exit 1
python3 -c "
import json
d=json.loads(open('gpurun_out/tv_$v.json').read().strip().splitlines()[-1]); print('$v value under the profiler', round(d['value']))"
head -1 $(find /tmp/tv_$v -name "t_kernel_trace.csv") > gpurun_out/tv_$v.csv
grep -v "at::\|elementwise\|vectorized\|Memcpy\|rocprim\|hipcub\|fillBuffer" $(find /tmp/tv_$v -name "t_kernel_trace.csv") | tail -n +2 >> gpurun_out/tv_$v.csv
python3 tools/trace_summary.py gpurun_out/tv_$v.csv | head -22
