#!/bin/bash
# round 4, GPU job 2: (1) the whole GPU suite on the tree with the rewritten host drivers, the multi-GPU bench line and config 5 on the
# reference's kernels in the default run; (2) `bench.py --gpus 2` started bare on the box's one GPU (shared-memory transport: the
# orchestration of the new line -- both exchange orders, microseconds per exchange, single-GPU same size, speedup, config5 leg --
# not a number); (3) the three-stage probe: what a (sweep, sweep, sweep) launch would cost against the two-sweep launch
# (lab library, F3D_ABLATE8=64: WRONG results, right instruction stream; csrc/f3d_solve_pair8.h)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job2
mkdir -p $O
timeout -k 10 1100 python3 -X faulthandler -m pytest tests -q -m gpu -x --durations=15 > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -22 $O/tests.log
export F3D_COMM_BACKEND=shm F3D_SHM_CAP_MB=512
timeout -k 10 1500 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu > $O/bench_gpus2_bare_shm.json 2> $O/bench_gpus2.err || { tail -40 $O/bench_gpus2.err; exit 1; }
unset F3D_COMM_BACKEND F3D_SHM_CAP_MB
tail -1 $O/bench_gpus2_bare_shm.json | cut -c1-600
for ty in 4 8; do
  for s in 24 48 70 96 128 256 512; do
    [ $ty = 4 ] && [ $s -gt 128 ] && continue
    for abl in 0 64; do
      echo "== TY=$ty size=$s ablate=$abl" >> $O/three_stage_probe.txt
      if [ $abl = 0 ]; then
        F3D_LIBDIR=$R/cuda-flow3d_amd/lib/lab F3D_PAIR8_TY=$ty timeout -k 10 120 python3 tools/kbench.py --size $s --reps 40 --kernel sweep2 2>&1 | grep sweep2 >> $O/three_stage_probe.txt
      else
        F3D_PAIR8_TY=$ty timeout -k 10 120 python3 tools/kbench.py --size $s --reps 40 --kernel sweep2 --ablate 64 2>&1 | grep sweep2 >> $O/three_stage_probe.txt
      fi
    done
  done
done
cat $O/three_stage_probe.txt
