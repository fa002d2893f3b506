#!/bin/bash
# round 3, first GPU call: tests, default bench, the two-rank rehearsal of the launcher, a kernel trace of ONE chunk
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r03a; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/gpu_tests.log
tail -3 $OUT/gpu_tests.log
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench failed"; tail -5 $OUT/bench_default.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'single', d.get('single_chunk_frames_per_s'), 'events', d['event_records'], 'frac', d['roofline']['frac'], 'lk_us', d['roofline']['avg_launch_us'])"
SVO_BENCH_REHEARSE_ON_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --chunks-per-gpu 16 > $OUT/bench_rehearse2.json 2> $OUT/bench_rehearse2.err; echo "rehearse rc=$?"; cut -c1-300 $OUT/bench_rehearse2.json
keep() { head -1 "$1" > "$2"; grep -v "at::\|elementwise\|vectorized\|Memcpy\|rocprim\|hipcub\|fillBuffer" "$1" | tail -n +2 >> "$2"; }
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/r03_one -o one -- python3 bench.py --chunks-per-gpu 1 --steps 60 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/one_bench.json 2> $OUT/one.err || { echo "trace failed"; tail -5 $OUT/one.err; exit 2; }
keep $(find /tmp/r03_one -name "one_kernel_trace.csv") $OUT/one_trace.csv
cut -c1-200 $OUT/one_bench.json
du -sh $OUT
