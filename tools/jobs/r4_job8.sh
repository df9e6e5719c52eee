#!/bin/bash
# round 4, GPU job 8: soak of the hand-scheduled fused launches (tools/soak_fused.py, 6 minutes): the tile without halo rows of the thin
# volumes and the three-stage launches are in it since this round
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job8
mkdir -p $O
timeout -k 10 500 python3 -X faulthandler tools/soak_fused.py 360 20261005 > $O/soak_fused.txt 2>&1 || { tail -30 $O/soak_fused.txt; exit 1; }
tail -4 $O/soak_fused.txt
