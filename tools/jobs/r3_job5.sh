#!/bin/bash
# round 3, GPU job 5: gathered-frame warp on the GPU (in-process and across processes), A/B of the fast weights
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job5
mkdir -p $O
python3 -X faulthandler -m pytest tests/test_gpu_slab.py tests/test_gpu_slab_procs.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for s in 512 256 128; do
  for rep in 1 2; do
  for lib in ab_base/cuda-flow3d_amd/lib cuda-flow3d_amd/lib; do
    echo "== $lib size $s" >> $O/kb.log
    F3D_LIBDIR=$R/$lib python3 tools/kbench.py --size $s --reps 20 --kernel sweeppk 2>&1 | grep -v "^\[" >> $O/kb.log
  done
  done
done
cat $O/kb.log
