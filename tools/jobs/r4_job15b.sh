#!/bin/bash
# round 4, GPU job 15b: (1) out-of-core 1024^3 on 16 GB with the host scratch prepared on a helper thread beside the resident levels
# against preparing it in line (F3D_P_SCRATCH_THREAD=0), result checked against the resident driver; (2) the out-of-core tests; (3) the
# round's rocprofv3 evidence on the final kernels (tools/profile_round.sh: default bench line, kernel stats of the same command, counters)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job15b
mkdir -p $O
for th in 0 1; do
  echo "== F3D_P_SCRATCH_THREAD=$th" >> $O/pbench_1024_16gb.txt
  chk="--no-resident"; [ $th = 1 ] && chk="--check"
  F3D_P_SCRATCH_THREAD=$th timeout -k 10 600 python3 tools/pbench.py --size 1024 --budget-mb 16384 $chk >> $O/pbench_1024_16gb.txt 2>&1 || { tail -20 $O/pbench_1024_16gb.txt; exit 1; }
done
grep -E "^==|piecemeal:|frames|identical|DIFFER" $O/pbench_1024_16gb.txt
timeout -k 10 400 python3 -X faulthandler -m pytest tests/test_gpu_piecemeal.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
F3D_OUT=$O timeout -k 10 700 bash tools/profile_round.sh > $O/profile_round.log 2>&1 || { tail -30 $O/profile_round.log; exit 1; }
tail -3 $O/profile_round.log
