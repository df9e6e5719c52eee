#!/bin/bash
# round 3, GPU job 3: exhaustive check of cheaper routes to 1/(2 sqrt(a)); kernel parity after the instruction diet; A/B timing of
# the fused kernels against the library of the last commit (ab_base/)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job3
mkdir -p $O
tools/lab/bin/phi_exact_lab > $O/phi_exact.log 2>&1
tools/lab/bin/phi_exact_lab 00000000 7f800000 >> $O/phi_exact.log 2>&1
cat $O/phi_exact.log
python3 -X faulthandler -m pytest tests/test_gpu_kernels.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for s in 512 384 256 128; do
  for lib in ab_base/cuda-flow3d_amd/lib cuda-flow3d_amd/lib; do
    echo "== $lib size $s" >> $O/kb.log
    F3D_LIBDIR=$R/$lib python3 tools/kbench.py --size $s --reps 20 --kernel sweep2 2>&1 | grep -v "^\[" >> $O/kb.log
    F3D_LIBDIR=$R/$lib python3 tools/kbench.py --size $s --reps 20 --kernel sweeppk 2>&1 | grep -v "^\[" >> $O/kb.log
  done
done
for ty in 4 8 12; do
  echo "== C3 dims, z march, F3D_PAIR8_TY=$ty" >> $O/kb.log
  F3D_PAIR8_YMARCH=0 F3D_PAIR8_TY=$ty python3 tools/kbench.py --dims 584 388 5 --reps 50 --kernel sweep2 2>&1 | grep -v "^\[" >> $O/kb.log
done
cat $O/kb.log
