#!/bin/bash
# round 4, GPU job 31: frame derivatives read (default) against formed from the frames inside the fused launches, on the final kernels: whole
# 512^3 and 256^3 solves, alternating in one call
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job31
mkdir -p $O
for round in 1 2; do
  for fd in 1 0; do
    for size in 512 256; do
      echo "== F3D_FRAME_DERIVATIVES=$fd size $size (round $round)" >> $O/fd_ab.txt
      F3D_FRAME_DERIVATIVES=$fd timeout -k 10 300 python3 tools/trace_size.py --size $size --reps 4 2>&1 | tail -1 >> $O/fd_ab.txt
    done
  done
done
cat $O/fd_ab.txt
