#!/bin/bash
# round 4, GPU job 6a: the suite exactly as the driver runs it (1024^3 on the reference's kernels included), the out-of-core 1024^3 run
# on a 16 GB budget with and without the flow update inside the solver (result checked against the resident driver once), BASELINE
# config 5 on one GPU
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job6a
mkdir -p $O
timeout -k 10 1000 python3 -X faulthandler -m pytest tests -q -m gpu -x --durations=6 > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -10 $O/tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
for fa in 0 1; do
  echo "== F3D_P_FUSED_ADD=$fa" >> $O/pbench_1024_16gb.txt
  chk="--no-resident"; [ $fa = 1 ] && chk="--check"
  F3D_P_FUSED_ADD=$fa timeout -k 10 600 python3 tools/pbench.py --size 1024 --budget-mb 16384 $chk >> $O/pbench_1024_16gb.txt 2>&1 || { tail -20 $O/pbench_1024_16gb.txt; exit 1; }
done
grep -v "^\[" $O/pbench_1024_16gb.txt | tail -24
timeout -k 10 600 python3 bench.py --size 1024 --steps 2 --warmup 1 --no-extra > $O/c5_one_gpu.json 2> $O/c5_one_gpu.err || { tail -20 $O/c5_one_gpu.err; exit 1; }
tail -1 $O/c5_one_gpu.json | cut -c1-200
