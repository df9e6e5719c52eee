#!/bin/bash
# re-entry check of the rebuilt tree: the whole GPU suite, smoke, the bench as the driver runs it
set -e
OUT=${F3D_OUT:-gpurun_out}/r3/job36
mkdir -p $OUT
timeout -k 10 900 python -X faulthandler -m pytest tests -q -m gpu -x > $OUT/tests.log 2>&1 || { tail -60 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 || { tail -30 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail -30 $OUT/bench.err; exit 1; }
tail -1 $OUT/bench.json | cut -c1-600
