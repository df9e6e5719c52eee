#!/bin/bash
# round 4, GPU job 9: cache hints.  The fused kernels' outputs are consumed by the NEXT launch and five of the twelve inputs of the
# frame-derivative builds are read by exactly one tile: A/B of (a) the shipped library, (b) non-temporal stores (`nt` on the hand-issued
# global_store_dword), (c) that plus non-temporal DMA loads of the centre-only inputs -- kernel timings at 512^3 and whole 512^3 solves,
# alternating in one call (ab_nt/lib, ab_nt2/lib: builds of the same tree with one / two lines changed)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job9
mkdir -p $O
for round in 1 2; do
  for lib in cuda-flow3d_amd/lib ab_nt/lib ab_nt2/lib; do
    echo "== $lib (round $round)" >> $O/nt_ab.txt
    F3D_LIBDIR=$R/$lib timeout -k 10 200 python3 tools/kbench.py --size 512 --reps 20 --kernel bothfd 2>&1 | grep -E "sweep2|sweeppk" >> $O/nt_ab.txt
    F3D_LIBDIR=$R/$lib timeout -k 10 300 python3 tools/trace_size.py --size 512 --reps 4 2>&1 | tail -1 >> $O/nt_ab.txt
  done
done
cat $O/nt_ab.txt
