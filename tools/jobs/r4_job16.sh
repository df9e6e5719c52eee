#!/bin/bash
# round 4, GPU job 16: out-of-core 1024^3 on 16 GB with the resample operator's two-set schedule (upload of the next chunk beside the
# kernels and the download of this one), volumes page-locked four at a time, host scratch on a helper thread -- each against its switch,
# result checked against the resident driver once; then the out-of-core tests (incl. the resample operator on page-locked volumes)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job16
mkdir -p $O
timeout -k 10 400 python3 -X faulthandler -m pytest tests/test_gpu_piecemeal.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() {  # tag, env...
  tag=$1; shift
  echo "== $tag: $*" >> $O/pbench_1024_16gb.txt
  env "$@" timeout -k 10 600 python3 tools/pbench.py --size 1024 --budget-mb 16384 $CHK >> $O/pbench_1024_16gb.txt 2>&1 || { tail -20 $O/pbench_1024_16gb.txt; exit 1; }
}
CHK="--no-resident"
run "volumes page-locked one after the other, scratch in line" F3D_P_PIN_THREADS=1 F3D_P_SCRATCH_THREAD=0
run "four at a time, scratch in line" F3D_P_PIN_THREADS=4 F3D_P_SCRATCH_THREAD=0
run "eight at a time, scratch in line" F3D_P_PIN_THREADS=8 F3D_P_SCRATCH_THREAD=0
CHK="--check"
run "default: four at a time, scratch beside the resident levels" F3D_DUMMY=1
grep -E "^==|piecemeal:|frames|identical|DIFFER" $O/pbench_1024_16gb.txt
