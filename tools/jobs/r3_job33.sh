#!/bin/bash
# BASELINE config 5 (1024^3) on the reference's kernels: once, for the record
set -e
OUT=${F3D_OUT:-gpurun_out}/r3/job33
mkdir -p $OUT
( while true; do sleep 60; echo "still running $(date +%T)"; done ) &
PING=$!
F3D_REF_C5=1 timeout -k 10 1100 python -m pytest tests/test_gpu_reference_kernels.py -q -m gpu -s -k "config_5" > $OUT/tests.log 2>&1 || { kill $PING; tail -40 $OUT/tests.log; exit 1; }
kill $PING
tail -4 $OUT/tests.log
