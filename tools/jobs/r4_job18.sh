#!/bin/bash
# round 4, GPU job 18: out-of-core 1024^3 on 16 GB, per-level solver seconds with every field in the chunk sets, with the cost model's
# choice, and with the constant fields on the device wherever they fit -- where does the layout pay, where does the model misprice it?
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job18
mkdir -p $O
for mode in 0 auto 1; do
  unset F3D_P_CONSTANTS
  [ $mode != auto ] && export F3D_P_CONSTANTS=$mode
  timeout -k 10 600 python3 tools/pbench.py --size 1024 --budget-mb 16384 --no-resident --verbose > $O/verbose_$mode.txt 2>&1 || { tail -20 $O/verbose_$mode.txt; exit 1; }
  echo "== F3D_P_CONSTANTS=$mode" >> $O/per_level.txt
  grep -E "solver of level|piecemeal:|frames " $O/verbose_$mode.txt >> $O/per_level.txt
done
cat $O/per_level.txt | cut -c1-200
