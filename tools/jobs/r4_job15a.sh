#!/bin/bash
# round 4, GPU job 15a: the final tree -- the suite exactly as the driver runs it (1024^3 on the reference's kernels included), the smoke
# entry, five minutes of the fused-launch soak on the shipped kernels (non-temporal stores)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job15a
mkdir -p $O
timeout -k 10 900 python3 -X faulthandler -m pytest tests -q -m gpu -x --durations=6 > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -10 $O/tests.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 400 python3 tools/soak_fused.py 270 > $O/soak.txt 2>&1 || { tail -20 $O/soak.txt; exit 1; }
tail -3 $O/soak.txt
