#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03c; mkdir -p $OUT
timeout -k 10 1000 python3 -m pytest tests -m gpu -q --durations=8 > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/gpu_tests.log
