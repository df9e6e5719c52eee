#!/bin/bash
# round 4, GPU job 26: launch-order experiment -- every other fused launch deals its tiles from the last to the first (F3D_PAIR8_FLIP=1), so
# that a launch starts on the planes the launch before wrote last (memory-side cache, 256 MB of the 1.6 GB a launch writes), with
# non-temporal stores (the shipped library) and with plain stores (ab_flip_plain/lib); whole 512^3 solves, alternating in one call
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job26
mkdir -p $O
F3D_PAIR8_FLIP=1 timeout -k 10 300 python3 -X faulthandler -m pytest tests/test_gpu_configs.py -q -m gpu -x -k "c4" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for round in 1 2; do
  for lib in cuda-flow3d_amd/lib ab_flip_plain/lib; do
    for flip in 0 1; do
      echo "== $lib F3D_PAIR8_FLIP=$flip (round $round)" >> $O/flip.txt
      F3D_PAIR8_FLIP=$flip F3D_LIBDIR=$R/$lib timeout -k 10 300 python3 tools/trace_size.py --size 512 --reps 4 2>&1 | tail -1 >> $O/flip.txt
    done
  done
done
cat $O/flip.txt
