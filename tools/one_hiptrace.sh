#!/bin/bash
# HIP API trace of ONE chunk per GPU: which host calls block (gpurun_out/one_hip_api.txt)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
rm -rf /tmp/oneh
timeout -k 10 300 rocprofv3 --hip-runtime-trace --output-format csv -d /tmp/oneh -o one -- python3 bench.py --chunks-per-gpu 1 --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/oneh_bench.json 2> gpurun_out/oneh.err || exit 1
T=$(find /tmp/oneh -name "one_hip_api_trace.csv")
grep "hipLaunchKernel\|hipEventRecord\|hipStreamWaitEvent" "$T" | tail -n 4300 | cut -d, -f2,4,6,7 > gpurun_out/one_hip_tail.csv
python3 - "$T" > gpurun_out/one_hip_api.txt <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
print(len(rows),'calls; columns',list(rows[0].keys()))
d=collections.defaultdict(list)
for r in rows[len(rows)//2:]:
    d[r['Function']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    v.sort()
    print('%-40s n %6d total %10.1f us  med %7.1f  p90 %7.1f  max %8.1f'%(k,len(v),sum(v),v[len(v)//2],v[int(len(v)*.9)],v[-1]))
# the sequence of one frame in the steady state: the last 60 calls
for r in rows[-120:-60]:
    print(r['Function'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
