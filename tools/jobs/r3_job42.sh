#!/bin/bash
# NOTE: the "new" library of this job was a timing build (idle role in k_pair8) that was not kept -- see DESIGN.md section 7
# round 3, GPU job 42: waves whose stage 1 nobody reads (rows outside the volume, the column wave of a one-column level) only keep
# the barriers: kernel / pipeline / config / slab tests, then BASELINE configs 2, 3, 4 against the build before (ab_old/), alternating
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job42
mkdir -p $O
timeout -k 10 900 python3 -X faulthandler -m pytest tests -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
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
  if [ $lib = new ]; then python3 tools/slab8_profile.py --size 512 --only slabs 2>&1 | tail -1 | sed "s/^/$lib /" >> $O/slab8.log
  else F3D_LIBDIR=$R/ab_old python3 tools/slab8_profile.py --size 512 --only slabs 2>&1 | tail -1 | sed "s/^/$lib /" >> $O/slab8.log; fi
done
cat $O/slab8.log
