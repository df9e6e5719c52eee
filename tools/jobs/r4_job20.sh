#!/bin/bash
# round 4, GPU job 20: out-of-core with the compute-only fields shared between the two chunk sets: the out-of-core tests and the C4 / C5
# schedules of test_gpu_configs that go through this path; 1024^3 on 16 GB, per-level solver seconds, with every field in the chunk sets
# and with the layout chosen per level, result checked against the resident driver
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job20
mkdir -p $O
timeout -k 10 600 python3 -X faulthandler -m pytest tests/test_gpu_piecemeal.py tests/test_gpu_configs.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for mode in 0 auto; do
  unset F3D_P_CONSTANTS
  chk="--check"
  [ $mode != auto ] && export F3D_P_CONSTANTS=$mode && chk="--no-resident"
  timeout -k 10 600 python3 tools/pbench.py --size 1024 --budget-mb 16384 $chk --verbose > $O/verbose_$mode.txt 2>&1 || { tail -20 $O/verbose_$mode.txt; exit 1; }
  echo "== F3D_P_CONSTANTS=$mode" >> $O/per_level.txt
  grep -E "solver of level|piecemeal:|frames |identical|DIFFER" $O/verbose_$mode.txt >> $O/per_level.txt
done
cut -c1-230 $O/per_level.txt
