#!/bin/bash
# round 3, GPU job 2: the y-marching fused kernels for thin volumes (parity, then timing against the z march), grid barrier lab
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job2
mkdir -p $O
python3 -X faulthandler -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "thin or fused or random_shapes or minimum" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python3 -X faulthandler -m pytest tests/test_gpu_pipeline.py -q -m gpu -x > $O/tests_pipe.log 2>&1 || { tail -30 $O/tests_pipe.log; exit 1; }
tail -3 $O/tests_pipe.log
for ym in 0 1; do
  for dims in "584 388 5" "555 369 5" "501 333 4" "369 245 4"; do
    echo "YMARCH=$ym dims $dims" >> $O/kb.log
    F3D_PAIR8_YMARCH=$ym python3 tools/kbench.py --dims $dims --reps 50 --kernel sweep2 2>&1 | grep -v "^\[" >> $O/kb.log
    F3D_PAIR8_YMARCH=$ym python3 tools/kbench.py --dims $dims --reps 50 --kernel sweeppk 2>&1 | grep -v "^\[" >> $O/kb.log
  done
done
cat $O/kb.log
F3D_PAIR8_YMARCH=0 python3 tools/trace_size.py --config c3 --reps 5 2>&1 | tail -1 | tee $O/c3_zmarch.log
python3 tools/trace_size.py --config c3 --reps 5 2>&1 | tail -1 | tee $O/c3_ymarch.log
tools/lab/bin/grid_sync_lab > $O/grid_sync.log 2>&1
cat $O/grid_sync.log
