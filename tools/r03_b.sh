#!/bin/bash
# round 3, second GPU call: shared math on the device, divergence GPU vs oracle over 200 frames, launch-gap microbenchmark
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/r03b; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_math.py -x -q > $OUT/math.log 2>&1; echo "math rc=$?"; tail -3 $OUT/math.log
timeout -k 10 120 tools/launch_gap > $OUT/launch_gap.jsonl 2>&1; echo "gap rc=$?"; cat $OUT/launch_gap.jsonl
timeout -k 10 600 python3 tools/diverge.py 201 4096 > $OUT/diverge_4096.log 2>&1; echo "diverge rc=$?"; tail -40 $OUT/diverge_4096.log
