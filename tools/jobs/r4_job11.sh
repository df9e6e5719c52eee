#!/bin/bash
# round 4, GPU job 11: non-temporal stores in the solver kernels (the hand-issued global_store_dword of k_pair8 / k_tri / k_sweep6 carry `nt`):
# (0) lab: does a vector instruction cost less when lanes are masked off (tools/lab/exec_mask_lab)?  (1) whole GPU suite (1024^3
# reference run left out); (2) the counter record on the new kernels; (3) the bench as the driver runs it; (4) configs 2 and 3
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job11
mkdir -p $O
timeout -k 10 120 tools/lab/bin/exec_mask_lab > $O/exec_mask_lab.txt 2>&1 || { cat $O/exec_mask_lab.txt; exit 1; }
cat $O/exec_mask_lab.txt
F3D_REF_C5=0 timeout -k 10 1000 python3 -X faulthandler -m pytest tests -q -m gpu -x --durations=6 > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -10 $O/tests.log
F3D_OUT=$O timeout -k 10 900 bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 || { tail -30 $O/pmc_traffic.log; exit 1; }
grep -A1 '"_solver_kernels_sha16"' $O/traffic/pmc_traffic.json | head -2
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
tail -1 $O/bench_steps20_warmup5.json | cut -c1-300
