#!/bin/bash
# A/B of library variants on ONE GPU box (boxes differ by a few percent, runs on one box by about one):
#   [BENCH_FLAGS="..."] tools/ab.sh ROUNDS name1 name2 ...   with the variants built as ab/lib_<name>.so (git-ignored scratch)
# Every round runs the default bench once per variant, alternating; prints frames/s per run and the medians.
# A variant is loaded through SVO_LIB (ros_stereo_slam_amd/capi.py): the installed libsvo_hip.so is never touched,
# so an interrupted run cannot leave a variant build in its place.
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
ROUNDS=$1; shift
mkdir -p gpurun_out/ab
for v in "$@"; do [ -f "ab/lib_$v.so" ] || { echo "ab/lib_$v.so missing"; exit 1; }; done
for r in $(seq 1 "$ROUNDS"); do
  for v in "$@"; do
    SVO_LIB="$PWD/ab/lib_$v.so" timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras ${BENCH_FLAGS:-} > gpurun_out/ab/${v}_$r.json 2> gpurun_out/ab/${v}_$r.err || { echo "$v round $r failed"; exit 1; }
    python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab/${v}_$r.json').read().strip().splitlines()[-1])
print('$v', $r, round(d['value']), round(d.get('single_chunk_frames_per_s') or 0), flush=True)" || exit 1
  done
done
python3 - "$@" <<'PY'
import json,sys,glob,statistics
for v in sys.argv[1:]:
    vals=[json.loads(open(f).read().strip().splitlines()[-1])['value'] for f in sorted(glob.glob(f'gpurun_out/ab/{v}_*.json'))]
    print(v, 'median', round(statistics.median(vals)), 'all', [round(x) for x in vals])
PY
