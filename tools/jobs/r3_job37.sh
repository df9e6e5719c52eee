#!/bin/bash
# round 3, GPU job 37: u, v, w (and the two frames) through "+=", median, resampling and the clearing of the increments in one
# launch each (f3d_*_n): the new entries against the single-volume ones and the oracle, the pipeline / config digests, and
# BASELINE configs 2, 3, 4 timed against the previous build (ab_old/) in the same call
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job37
mkdir -p $O
timeout -k 10 600 python3 -X faulthandler -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
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
timeout -k 10 600 python3 -X faulthandler -m pytest tests/test_gpu_configs.py -q -m gpu -x > $O/tests_configs.log 2>&1 || { tail -60 $O/tests_configs.log; exit 1; }
tail -2 $O/tests_configs.log
for lib in ab_old new ab_old new; do
  if [ $lib = new ]; then python3 bench.py --steps 4 --warmup 1 --no-extra 2>/dev/null > $O/b.json
  else F3D_LIBDIR=$R/ab_old python3 bench.py --steps 4 --warmup 1 --no-extra 2>/dev/null > $O/b.json; fi
  python3 -c "
import json,sys
b=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); r=b['roofline']
print('$lib value', b['value'], 'ms', b['ms_per_step'], 'pair frac', r['frac'], 'parity', b['parity']['match'])" >> $O/bench.log
done
cat $O/bench.log
