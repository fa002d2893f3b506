#!/bin/bash
# the default bench once per library variant (ab/lib_<name>.so), with the single-chunk stage table:  tools/ab_stages.sh name1 name2 ...
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  SVO_LIB=$PWD/ab/lib_$v.so timeout -k 10 400 python3 bench.py --no-cpu-baseline > gpurun_out/ab_$v.json 2>/dev/null || exit 1
  python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab_{v}.json").read().strip().splitlines()[-1])
print(v, round(d["value"]), round(d["single_chunk_frames_per_s"]),
      {k: round(x, 1) for k, x in d["single_chunk_stage_us"].items() if isinstance(x, (int, float))}, flush=True)
PY
done
