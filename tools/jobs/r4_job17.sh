#!/bin/bash
# round 4, GPU job 17: out-of-core -- (1) the out-of-core tests (constant fields held on the device, resample operator with two buffer
# sets); (2) 1024^3 on 16 GB: every field in the chunk sets (F3D_P_CONSTANTS=0) against the cost model's choice, result checked against
# the resident driver; the resample operator in order on one stream (F3D_P_OVERLAP=0 would also serialise the solver, so this run only
# shows in the frames / flow_resample clocks against job 16's 1.95 / 2.31 s)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job17
mkdir -p $O
timeout -k 10 400 python3 -X faulthandler -m pytest tests/test_gpu_piecemeal.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() {  # tag, env...
  tag=$1; shift
  echo "== $tag: $*" >> $O/pbench_1024_16gb.txt
  env "$@" timeout -k 10 600 python3 tools/pbench.py --size 1024 --budget-mb 16384 $CHK >> $O/pbench_1024_16gb.txt 2>&1 || { tail -20 $O/pbench_1024_16gb.txt; exit 1; }
}
CHK="--no-resident"
run "every field in the chunk sets" F3D_P_CONSTANTS=0
CHK="--check"
run "default: the cost model decides per level whether the frames and u, v, w stay on the device" F3D_DUMMY=1
CHK="--no-resident"
run "constants on the device wherever they fit" F3D_P_CONSTANTS=1
grep -E "^==|piecemeal:|frames|identical|DIFFER" $O/pbench_1024_16gb.txt
