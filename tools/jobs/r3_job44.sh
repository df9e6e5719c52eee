#!/bin/bash
# round 3, GPU job 44: the two constants of the tile-height choice (cost of a 16-wave / an 8-wave step in % of a 12-wave one) swept on
# BASELINE configs 2, 3 (tools/trace_size.py) and 4 (bench.py --no-extra) with the frame-derivative builds that are the default now
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job44
mkdir -p $O
for s12 in 128 112 120 136 128; do
  F3D_PAIR8_STEP12=$s12 python3 bench.py --steps 3 --warmup 1 --no-extra 2>/dev/null > $O/b.json
  python3 -c "
import json
b=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); r=b['roofline']
print('STEP12=$s12 C4 ms', b['ms_per_step'], 'pair us', r['avg_launch_us'], 'sp GB/s', r['sweep_phi_ksi']['achieved'], 'parity', b['parity']['match'])" | tee -a $O/sweep.log
  for c in c2 c3; do F3D_PAIR8_STEP12=$s12 python3 tools/trace_size.py --config $c --reps 5 2>&1 | tail -1 | sed "s/^/STEP12=$s12 /" | tee -a $O/sweep.log; done
done
for s4 in 80 65 95 110 80; do
  for c in c2 c3; do F3D_PAIR8_STEP4=$s4 python3 tools/trace_size.py --config $c --reps 5 2>&1 | tail -1 | sed "s/^/STEP4=$s4 /" | tee -a $O/sweep.log; done
done
