#!/bin/bash
# round 4, GPU job 19: out-of-core, final: the out-of-core tests; 1024^3 on 16 GB with every field in the chunk sets and with the layout
# chosen per level by the summed link traffic (twice each, alternating: runs differ by ~1 s), result checked against the resident driver
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job19
mkdir -p $O
timeout -k 10 400 python3 -X faulthandler -m pytest tests/test_gpu_piecemeal.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() {  # tag, env...
  tag=$1; shift
  echo "== $tag: $*" >> $O/pbench_1024_16gb.txt
  env "$@" timeout -k 10 600 python3 tools/pbench.py --size 1024 --budget-mb 16384 $CHK >> $O/pbench_1024_16gb.txt 2>&1 || { tail -20 $O/pbench_1024_16gb.txt; exit 1; }
}
CHK="--no-resident"
run "every field in the chunk sets (1)" F3D_P_CONSTANTS=0
run "layout per level (1)" F3D_DUMMY=1
run "every field in the chunk sets (2)" F3D_P_CONSTANTS=0
CHK="--check"
run "layout per level (2)" F3D_DUMMY=1
grep -E "^==|piecemeal:|frames|identical|DIFFER" $O/pbench_1024_16gb.txt
