#!/bin/bash
# round 4, GPU job 23: out-of-core 1024^3 on 16 GB now that link and device are in balance: the last sweep + next weights as one launch
# inside chunked residencies (F3D_P_FUSED=1: 23 buffers instead of 21), and 7 / 8 / 10 / 12 outer iterations per residency against the planner's choice
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job23
mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout -k 10 600 python3 tools/pbench.py --size 1024 --budget-mb 16384 --no-resident --verbose > $O/verbose.txt 2>&1 || { tail -20 $O/verbose.txt; exit 1; }
  echo "== $tag: $*" >> $O/variants.txt
  grep -E "solver of level [0-3]:|piecemeal:|frames " $O/verbose.txt >> $O/variants.txt
}
run "planner's choice" F3D_DUMMY=1
run "last sweep + next weights in one launch" F3D_P_FUSED=1
run "7 outer iterations per residency" F3D_P_OUTER_PER_PASS=7
run "8" F3D_P_OUTER_PER_PASS=8
run "10" F3D_P_OUTER_PER_PASS=10
run "12" F3D_P_OUTER_PER_PASS=12
cut -c1-200 $O/variants.txt
