#!/bin/bash
# NOTE: F3D_PITCH_PAD existed only in the timing build of this job (f3d_alloc_pitched adding n x 256 B per row); not kept -- see DESIGN.md section 7
# round 3, GPU job 46: row pitch of the containers padded by n x 256 B (F3D_PITCH_PAD) on the 512^3 solve: do 2 KiB rows alias in the memory system?
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job46
mkdir -p $O
for pad in ${PADS:-0 1 2 3 5 0 1}; do
  F3D_PITCH_PAD=$pad python3 bench.py --steps 4 --warmup 1 --no-extra 2>/dev/null > $O/b.json
  python3 -c "
import json
b=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); r=b['roofline']
print('PAD=$pad C4 ms', b['ms_per_step'], 'pair us', r['avg_launch_us'], 'finest', r['finest_level']['avg_launch_us'], 'parity', b['parity']['match'])" | tee -a $O/pad.log
done
