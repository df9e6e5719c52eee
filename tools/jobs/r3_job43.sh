#!/bin/bash
# round 3, GPU job 43: ten minutes of tools/soak_fused.py on the final tree (random boxes through every fused launch against the separate ones)
set -e
O=${F3D_OUT:-gpurun_out}/r3/job43
mkdir -p $O
timeout -k 10 780 python3 -X faulthandler tools/soak_fused.py 600 7 2>&1 | tee $O/soak.log | tail -30
