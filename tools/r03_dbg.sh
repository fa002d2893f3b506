for cfg in "SVO_LANE_CUS=0 SVO_T_CUS=0" "SVO_LANE_CUS=224 SVO_T_CUS=0" "SVO_LANE_CUS=224 SVO_T_CUS=224" "SVO_LANE_CUS=128 SVO_T_CUS=224"; do
  echo "== $cfg"
  env $cfg SVO_CHAIN_DEBUG=1 timeout -k 10 200 python3 bench.py --chunks-per-gpu 1 --steps 200 --warmup 10 --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/dbg.log 2>&1 || exit 1
  grep "svo chain" gpurun_out/dbg.log | tail -2
  python3 -c "
import json;d=json.loads([l for l in open('gpurun_out/dbg.log') if l.startswith('{')][-1]);print('frames/s', round(d['value']))"
done
