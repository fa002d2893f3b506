run() { # env, chunks, per-ctx
  env $1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --chunks-per-gpu $2 --chunks-per-context $3 2>gpurun_out/sweep.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 $3', round(d['value']), round(d['roofline']['avg_launch_us']), {k: round(v,3) for k,v in d['config']['stage_ms_per_frame'].items()})" || tail -3 gpurun_out/sweep.err
}
run SVO_LK_WAVES_PER_CU=0 64 16
run SVO_LK_WAVES_PER_CU=15 64 16
run SVO_LK_WAVES_PER_CU=14 64 16
run SVO_LK_WAVES_PER_CU=13 64 16
run SVO_LK_WAVES_PER_CU=12 64 16
run SVO_LK_WAVES_PER_CU=12 96 16
run SVO_LK_WAVES_PER_CU=14 96 16
run SVO_LK_WAVES_PER_CU=8 96 16
