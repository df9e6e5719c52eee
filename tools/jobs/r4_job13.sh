#!/bin/bash
# round 4, GPU job 13: frame 1 registered inside the out-of-core solver's first residency.  (1) the out-of-core tests; (2) 1024^3 on a
# 16 GB budget with the separate registration operator and with the registration inside (result checked against the resident driver);
# (3) the bench as the driver runs it, now with the counter record of the shipped kernels in place (traffic non-null)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job13
mkdir -p $O
timeout -k 10 600 python3 -X faulthandler -m pytest tests/test_gpu_piecemeal.py tests/test_gpu_configs.py -q -m gpu -x --durations=5 > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -8 $O/tests.log
for fw in 0 1; do
  echo "== F3D_P_FUSED_WARP=$fw" >> $O/pbench_1024_16gb.txt
  chk="--no-resident"; [ $fw = 1 ] && chk="--check"
  F3D_P_FUSED_WARP=$fw timeout -k 10 600 python3 tools/pbench.py --size 1024 --budget-mb 16384 $chk >> $O/pbench_1024_16gb.txt 2>&1 || { tail -20 $O/pbench_1024_16gb.txt; exit 1; }
done
grep -v "^\[" $O/pbench_1024_16gb.txt | tail -12
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
tail -1 $O/bench_steps20_warmup5.json | cut -c1-300
