#!/bin/bash
# round 4, GPU job 6b: the round's rocprofv3 evidence on the final kernels -- tools/profile_round.sh (default bench line, kernel stats of
# the same command, PMC passes), kernel traces of BASELINE configs 2 and 3 with per-level tables, and the 8-slab decomposition of the
# 512^3 and 1024^3 runs on one GPU against the unsplit runs, per level (frame derivatives per slab since this round)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job6b
mkdir -p $O
# (first: the tests added after job 6a was staged -- stage exchanges hidden behind the interior, thin volumes without skipped cases)
timeout -k 10 600 python3 -X faulthandler -m pytest tests/test_gpu_slab_procs.py tests/test_gpu_slab.py tests/test_gpu_kernels.py -q -m gpu -x -k "slab or thin or stage or overlapped" > $O/tests_new.log 2>&1 || { tail -40 $O/tests_new.log; exit 1; }
tail -3 $O/tests_new.log
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
for size in 512 1024; do
  for side in slabs unsplit; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/s${size}_$side -- python3 $R/tools/slab8_profile.py --size $size --only $side > $O/s${size}_$side.log 2>&1
    tail -1 $O/s${size}_$side.log
    t=$(ls $O/s${size}_$side/*/*_kernel_trace.csv | head -1)
    w=1; [ $side = slabs ] && w=8
    python3 $R/tools/level_table.py $t --warps-per-level $w --out $O/s${size}_${side}_levels.json > $O/s${size}_${side}_levels.md
    cp $(ls $O/s${size}_$side/*/*_kernel_stats.csv | head -1) $O/s${size}_${side}_kernel_stats.csv
    rm -f $O/s${size}_$side/*/*_kernel_trace.csv
  done
done
ls $O
