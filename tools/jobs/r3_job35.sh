#!/bin/bash
set -e
OUT=${F3D_OUT:-gpurun_out}/r3/job35
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_reference_kernels.py -q -m gpu -k "random_shapes" > $OUT/tests.log 2>&1 || { tail -60 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
