#!/bin/bash
# round 3, GPU job 1: baseline kernel numbers, kernel traces of BASELINE configs 2 and 3 with per-level tables, and the
# 8-slab decomposition of the 1024^3 run on one GPU against the unsplit run (per level).
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job1
mkdir -p $O
cd $R
python3 tools/kbench.py --size 512 --reps 10 --kernel sweep2 > $O/kb512.log 2>&1
python3 tools/kbench.py --size 512 --reps 10 --kernel sweeppk >> $O/kb512.log 2>&1
cat $O/kb512.log
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
for side in slabs unsplit; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1024_$side -- python3 $R/tools/slab8_profile.py --size 1024 --only $side > $O/s1024_$side.log 2>&1
  tail -1 $O/s1024_$side.log
  t=$(ls $O/s1024_$side/*/*_kernel_trace.csv | head -1)
  w=1; [ $side = slabs ] && w=8
  python3 $R/tools/level_table.py $t --warps-per-level $w --out $O/s1024_${side}_levels.json > $O/s1024_${side}_levels.md
  cp $(ls $O/s1024_$side/*/*_kernel_stats.csv | head -1) $O/s1024_${side}_kernel_stats.csv
  rm -f $O/s1024_$side/*/*_kernel_trace.csv
done
ls -la $O
cd $R
python3 -X faulthandler -m pytest tests/test_bench_launcher.py tests/test_gpu_pipeline.py -q -m gpu -k "bare or bag" > $O/tests.log 2>&1 || true
tail -5 $O/tests.log
