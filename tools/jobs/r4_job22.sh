#!/bin/bash
# round 4, GPU job 21 (rerun as job 22 with the increments handed on as well): out-of-core with the constants' shared planes handed from chunk set to chunk set on the device: tests, then
# 1024^3 on 16 GB per level, with whole windows uploaded (F3D_P_HANDOVER=0) and with the hand-over, result checked against the resident driver
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job22
mkdir -p $O
timeout -k 10 600 python3 -X faulthandler -m pytest tests/test_gpu_piecemeal.py tests/test_gpu_configs.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for mode in 0 1; do
  chk="--check"; [ $mode = 0 ] && chk="--no-resident"
  F3D_P_HANDOVER=$mode timeout -k 10 600 python3 tools/pbench.py --size 1024 --budget-mb 16384 $chk --verbose > $O/verbose_$mode.txt 2>&1 || { tail -20 $O/verbose_$mode.txt; exit 1; }
  echo "== F3D_P_HANDOVER=$mode" >> $O/per_level.txt
  grep -E "solver of level|piecemeal:|frames |identical|DIFFER" $O/verbose_$mode.txt >> $O/per_level.txt
done
cut -c1-230 $O/per_level.txt
