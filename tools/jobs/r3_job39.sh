#!/bin/bash
# round 3, GPU job 39 (final tree with the batched launches): the whole GPU suite, smoke, the bench exactly as the driver runs
# it, the round's rocprofv3 evidence (tools/profile_round.sh: default bench line, kernel stats of the same command, PMC passes) and
# the kernel traces of BASELINE configs 2 and 3 with per-level tables
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job39
mkdir -p $O
timeout -k 10 900 python3 -X faulthandler -m pytest tests -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
tail -1 $O/bench_steps20_warmup5.json | cut -c1-300
F3D_OUT=$O bash tools/profile_round.sh > $O/profile_round.log 2>&1 || { tail -30 $O/profile_round.log; exit 1; }
tail -3 $O/profile_round.log
cd /tmp && export TMPDIR=/tmp
for c in c2 c3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$c -- python3 $R/tools/trace_size.py --config $c --reps 3 > $O/$c.log 2>&1
  tail -1 $O/$c.log
  t=$(ls $O/$c/*/*_kernel_trace.csv | head -1)
  lv=40; [ $c = c3 ] && lv=10
  python3 $R/tools/level_table.py $t --levels $lv --out $O/${c}_levels.json > $O/${c}_levels.md
  cp $(ls $O/$c/*/*_kernel_stats.csv | head -1) $O/${c}_kernel_stats.csv
  rm -f $O/$c/*/*_kernel_trace.csv
done
ls $O
