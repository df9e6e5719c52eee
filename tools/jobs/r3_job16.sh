#!/bin/bash
# how well does the round model (f3d_solve_pair8.h: pair8_plan_dims) describe the fused kernels level by level?
set -e
OUT=${F3D_OUT:-gpurun_out}/r3/job16
mkdir -p $OUT
for n in 512 487 463 439 418 397 377 358 340 323 307 292 277 263 250 238 226 204 184 166; do
  timeout -k 10 200 python3 tools/kbench.py --size $n --reps 10 --kernel bothfd 2>&1 | grep -i "sweep" | tee -a $OUT/levels.log
done
