#!/bin/bash
# the reference's own kernels (oracle/_ref/*.hsaco, built from /root/reference by oracle/Makefile) against the oracle and the product
set -e
OUT=${F3D_OUT:-gpurun_out}/r3/job31
mkdir -p $OUT
ls -la oracle/_ref/ > $OUT/ref_files.txt
timeout -k 10 900 python -m pytest tests/test_gpu_reference_kernels.py -q -m gpu -s > $OUT/tests.log 2>&1 || { tail -60 $OUT/tests.log; exit 1; }
tail -5 $OUT/tests.log
