#!/bin/bash
# equal shares per workgroup (k_pair8 segments): parity, then level by level against the z-chunk plan and against the previous build
set -e
OUT=${F3D_OUT:-gpurun_out}/r3/job17
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
for n in 512 463 439 397 340 307 263 226; do
  echo "== $n base" | tee -a $OUT/levels.log
  F3D_LIBDIR=$PWD/ab_base/cuda-flow3d_amd/lib timeout -k 10 200 python3 tools/kbench.py --size $n --reps 10 --kernel bothfd 2>&1 | grep -i "sweep" | tee -a $OUT/levels.log
  for P in 0 1; do
    echo "== $n persist $P" | tee -a $OUT/levels.log
    F3D_PAIR8_PERSIST=$P timeout -k 10 200 python3 tools/kbench.py --size $n --reps 10 --kernel bothfd 2>&1 | grep -i "sweep" | tee -a $OUT/levels.log
  done
done
for L in base new; do
  if [ $L = base ]; then export F3D_LIBDIR=$PWD/ab_base/cuda-flow3d_amd/lib; else unset F3D_LIBDIR; fi
  echo "== $L" | tee -a $OUT/solve.log
  timeout -k 10 300 python3 tools/trace_size.py --size 512 --reps 2 2>&1 | grep "per solve" | tee -a $OUT/solve.log
  timeout -k 10 300 python3 tools/trace_size.py --config c2 --reps 5 2>&1 | grep "per solve" | tee -a $OUT/solve.log
done
