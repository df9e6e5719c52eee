#!/bin/bash
# round 4, GPU job 10: which cache policy for the hand-issued stores of the solver kernels?  plain (shipped), nt, sc1, sc0 sc1, nt sc1,
# nt sc0 sc1 -- kernel timings at 512^3 and whole 512^3 solves, two rounds alternating in one call
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job10
mkdir -p $O
for round in 1 2; do
  for lib in cuda-flow3d_amd/lib ab_nt/lib ab_s1/lib ab_s01/lib ab_nts1/lib ab_nts01/lib; do
    echo "== $lib (round $round)" >> $O/store_hints.txt
    F3D_LIBDIR=$R/$lib timeout -k 10 200 python3 tools/kbench.py --size 512 --reps 20 --kernel bothfd 2>&1 | grep -E "sweep2|sweeppk" >> $O/store_hints.txt
    F3D_LIBDIR=$R/$lib timeout -k 10 300 python3 tools/trace_size.py --size 512 --reps 3 2>&1 | tail -1 >> $O/store_hints.txt
  done
done
cat $O/store_hints.txt | paste - - - - | cut -c1-330
