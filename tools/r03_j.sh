#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/${1:-r03j}; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_slam_loop.py tests/test_gpu_sharded.py -m gpu -q -s > $OUT/slam.log 2>&1; rc=$?; echo "rc=$rc"; grep -v "^$" $OUT/slam.log | tail -22
