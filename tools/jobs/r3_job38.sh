#!/bin/bash
# round 3, GPU job 38: the batched launches (f3d_*_n) against the previous build (ab_old/) on BASELINE configs 2, 3, 4 in one call;
# slab drivers on the batched entries (N slabs == one GPU), C4 / C5 digests
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job38
mkdir -p $O
timeout -k 10 900 python3 -X faulthandler -m pytest tests/test_gpu_slab.py tests/test_gpu_slab_procs.py tests/test_gpu_configs.py tests/test_gpu_piecemeal.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do
  for lib in ab_old new; do
    for c in c2 c3; do
      if [ $lib = new ]; then python3 tools/trace_size.py --config $c --reps 5 2>&1 | tail -1 | sed "s/^/$lib /" >> $O/cfg.log
      else F3D_LIBDIR=$R/ab_old python3 tools/trace_size.py --config $c --reps 5 2>&1 | tail -1 | sed "s/^/$lib /" >> $O/cfg.log; fi
    done
  done
done
cat $O/cfg.log
for lib in ab_old new ab_old new; do
  if [ $lib = new ]; then python3 bench.py --steps 4 --warmup 1 --no-extra 2>/dev/null > $O/b.json
  else F3D_LIBDIR=$R/ab_old python3 bench.py --steps 4 --warmup 1 --no-extra 2>/dev/null > $O/b.json; fi
  python3 -c "
import json,sys
b=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); r=b['roofline']
print('$lib value', b['value'], 'ms', b['ms_per_step'], 'pair frac', r['frac'], 'parity', b['parity']['match'])" >> $O/bench.log
done
cat $O/bench.log
for lib in ab_old new; do
  for order in outer stage; do
    if [ $lib = new ]; then F3D_SLAB_EXCHANGE=$order python3 tools/slab8_profile.py --size 512 --only slabs 2>&1 | tail -1 | sed "s/^/$lib $order /" >> $O/slab8.log
    else F3D_LIBDIR=$R/ab_old F3D_SLAB_EXCHANGE=$order python3 tools/slab8_profile.py --size 512 --only slabs 2>&1 | tail -1 | sed "s/^/$lib $order /" >> $O/slab8.log; fi
  done
done
cat $O/slab8.log
