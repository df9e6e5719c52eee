#!/bin/bash
# round 4, GPU job 30: the final tree (after the huge-page request was taken out) -- the suite exactly as the driver runs it, the smoke entry, the bench as the driver runs it
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job30
mkdir -p $O
timeout -k 10 900 python3 -X faulthandler -m pytest tests -q -m gpu -x --durations=6 > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -10 $O/tests.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
tail -1 $O/bench_steps20_warmup5.json | cut -c1-300
