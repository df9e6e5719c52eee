#!/bin/bash
# round 4, GPU job 3b: the rest of job 3 (the suite stopped at a test that encoded the old swap count): remaining test files, then the
# k_tri timings and whole solves
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job3
mkdir -p $O
F3D_REF_C5=0 timeout -k 10 900 python3 -X faulthandler -m pytest tests/test_gpu_pipeline.py tests/test_gpu_piecemeal.py tests/test_gpu_slab.py tests/test_gpu_slab_procs.py tests/test_gpu_reference_kernels.py -q -m gpu -x --durations=8 > $O/tests_b.log 2>&1 || { tail -60 $O/tests_b.log; exit 1; }
tail -12 $O/tests_b.log
rm -f $O/tri_kbench.txt $O/tri_solves.txt
for s in 18 24 32 48 64 70 96 128 160 192 256; do
  echo "== size $s: two-stage launches (auto tile)" >> $O/tri_kbench.txt
  timeout -k 10 120 python3 tools/kbench.py --size $s --reps 40 --kernel both 2>&1 | grep -E "sweep2|sweeppk" >> $O/tri_kbench.txt
  for ty in 4 7; do
    echo "== size $s: three-stage launches, TY=$ty" >> $O/tri_kbench.txt
    F3D_TRI_TY=$ty timeout -k 10 120 python3 tools/kbench.py --size $s --reps 40 --kernel tri 2>&1 | grep -E "sweep3|sweep2pk" >> $O/tri_kbench.txt
  done
done
echo "== 584x388x5" >> $O/tri_kbench.txt
timeout -k 10 120 python3 tools/kbench.py --dims 584 388 5 --reps 40 --kernel both 2>&1 | grep -E "sweep2|sweeppk" >> $O/tri_kbench.txt
for ty in 4 7; do F3D_TRI_TY=$ty timeout -k 10 120 python3 tools/kbench.py --dims 584 388 5 --reps 40 --kernel tri 2>&1 | grep -E "sweep3|sweep2pk" >> $O/tri_kbench.txt; done
cat $O/tri_kbench.txt
for cfg in "--config c2" "--config c3" "--size 128" "--size 256" "--size 512"; do
  for env in "F3D_TRI=0" "F3D_TRI=1" "F3D_TRI_MAX_VOXELS=1e6" "F3D_TRI_MAX_VOXELS=8e6" "F3D_TRI_MAX_VOXELS=2e7"; do
    echo "== $cfg  $env" >> $O/tri_solves.txt
    env $env timeout -k 10 300 python3 tools/trace_size.py $cfg --reps 5 2>&1 | tail -1 >> $O/tri_solves.txt
  done
done
cat $O/tri_solves.txt
