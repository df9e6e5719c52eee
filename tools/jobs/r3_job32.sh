#!/bin/bash
# BASELINE config 4 on the reference's kernels
set -e
OUT=${F3D_OUT:-gpurun_out}/r3/job32
mkdir -p $OUT
( while true; do sleep 60; echo "still running $(date +%T)"; done ) &
PING=$!
timeout -k 10 1000 python -m pytest tests/test_gpu_reference_kernels.py -q -m gpu -s -k "config_4" > $OUT/tests.log 2>&1 || { kill $PING; tail -40 $OUT/tests.log; exit 1; }
kill $PING
tail -3 $OUT/tests.log
