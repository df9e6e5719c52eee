#!/bin/bash
# NOTE: bit 1 of F3D_XCD_REMAP existed only in the timing build of this job (tx / ty exchanged in the tile decomposition of k_pair8); not kept -- see DESIGN.md section 7
# round 3, GPU job 45: tile numbering of the fused launches along y first (F3D_XCD_REMAP=3) against x first (=1, shipped), alternating
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job45
mkdir -p $O
for m in 1 3 1 3; do
  F3D_XCD_REMAP=$m python3 bench.py --steps 4 --warmup 1 --no-extra 2>/dev/null > $O/b.json
  python3 -c "
import json
b=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); r=b['roofline']
print('REMAP=$m C4 ms', b['ms_per_step'], 'pair us', r['avg_launch_us'], 'finest', r['finest_level']['avg_launch_us'], 'parity', b['parity']['match'])" | tee -a $O/order.log
done
for m in 1 3 1 3; do
  for n in 512 418 340; do
    F3D_XCD_REMAP=$m python3 tools/kbench.py --size $n --reps 10 --kernel bothfd 2>&1 | grep -E "sweep2|sweeppk" | sed "s/^/REMAP=$m /" | tee -a $O/order.log
  done
done
