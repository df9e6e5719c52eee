#!/bin/bash
# round 3, GPU job 11: frame derivatives on / off on BASELINE configs 2, 3 and 4
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job11
mkdir -p $O
python3 -X faulthandler -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "frame_derivatives" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for fd in 0 1 0 1; do
  for c in c2 c3; do
    F3D_FRAME_DERIVATIVES=$fd python3 tools/trace_size.py --config $c --reps 5 2>&1 | tail -1 | sed "s/^/FD=$fd /" >> $O/cfg.log
  done
done
cat $O/cfg.log
for fd in 1 0 1 0; do
  F3D_FRAME_DERIVATIVES=$fd python3 bench.py --steps 4 --warmup 1 --no-extra 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('FD=$fd value', b['value'], 'ms', b['ms_per_step'], 'pair frac', r['frac'], 'finest', r['finest_level']['frac'], 'sp', r['sweep_phi_ksi']['achieved'], 'parity', b['parity']['match'])" >> $O/bench.log
done
cat $O/bench.log
