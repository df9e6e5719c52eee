#!/bin/bash
# round 3, GPU job 41: the tree after the grid.z guards of the batched launches: kernel, pipeline and slab tests, smoke, default bench
set -e
O=${F3D_OUT:-gpurun_out}/r3/job41
mkdir -p $O
timeout -k 10 900 python3 -X faulthandler -m pytest tests -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
tail -1 $O/bench_default.json | cut -c1-200
