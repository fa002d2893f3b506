#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03; mkdir -p $OUT
S0=$(date +%s); timeout -k 10 600 python3 bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err || { tail -5 $OUT/bench_full.err; exit 7; }
echo "default bench wall: $(( $(date +%s) - S0 )) s"
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err || exit 8
timeout -k 10 600 python3 bench.py --kpts 8192 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_8192.json 2> $OUT/bench_8192.err || exit 9
python3 - <<'PY'
import json
for f in ('bench_full','bench_20_5','bench_8192'):
    d=json.loads(open('gpurun_out/r03/'+f+'.json').read().strip().splitlines()[-1])
    print(f, round(d['value']), 'single', round(d['single_chunk_frames_per_s']), 'inits', round(d['value_including_chunk_inits']), 'ate', round(d['ate_rmse_vs_truth'],3), 'seq', round(d['ate_rmse_sequential_vs_truth'],3), 'sh-vs-seq', round(d['ate_rmse_sharded_vs_sequential'],3), 'frac', d['roofline']['frac'], 'lk_us', round(d['roofline']['avg_launch_us']), 'traffic', d['roofline']['traffic'], 'pg', d['posegraph']['posegraph_ms_per_iter'], d['posegraph']['ate_rmse_vs_truth_after'], 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('single_thread_value'), d.get('max_frame_delta_vs_oracle'))
PY
