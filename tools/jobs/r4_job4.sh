#!/bin/bash
# round 4, GPU job 4: the y-marching tile without halo rows (k_pair8t) for thin volumes.  (1) whole GPU suite (1024^3 reference run left
# out); (2) kernel timings on the level shapes of BASELINE config 3, halo rows against none, y march against z march; (3) config 3 whole
# solves under the same switches; (4) a short 512^3 bench (the body of k_pair8 moved into a device function: same speed?)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job4
mkdir -p $O
F3D_REF_C5=0 timeout -k 10 1000 python3 -X faulthandler -m pytest tests -q -m gpu -x --durations=6 > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -10 $O/tests.log
for dims in "584 388 5" "555 369 5" "528 351 5" "501 333 4" "476 316 4" "369 245 4"; do
  for env in "F3D_PAIR8_TIGHT=0 F3D_PAIR8_YMARCH=1" "F3D_PAIR8_TIGHT=1 F3D_PAIR8_YMARCH=1" "F3D_PAIR8_YMARCH=0"; do
    echo "== $dims  $env" >> $O/tight_kbench.txt
    env $env timeout -k 10 120 python3 tools/kbench.py --dims $dims --reps 60 --kernel both 2>&1 | grep -E "sweep2|sweeppk" >> $O/tight_kbench.txt
  done
done
cat $O/tight_kbench.txt
for env in "F3D_PAIR8_TIGHT=0" "F3D_PAIR8_TIGHT=1" "F3D_PAIR8_TIGHT=1 F3D_PAIR8_YMARCH=1" "F3D_PAIR8_TIGHT=0 F3D_PAIR8_YMARCH=1" "F3D_PAIR8_YMARCH=0"; do
  echo "== c3  $env" >> $O/tight_solves.txt
  env $env timeout -k 10 300 python3 tools/trace_size.py --config c3 --reps 8 2>&1 | tail -1 >> $O/tight_solves.txt
done
cat $O/tight_solves.txt
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-extra > $O/bench_short.json 2> $O/bench_short.err || { tail -20 $O/bench_short.err; exit 1; }
tail -1 $O/bench_short.json | cut -c1-200
