#!/bin/bash
# component lanes: parity first, then C2 / C3 / C4 with the chains on one lane, on three up to 200^3 (default), on three everywhere
set -e
OUT=${F3D_OUT:-gpurun_out}/r3/job15
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -x -q -m gpu -k "not c5" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
for L in 0 default 2100000 17000000 1000000000; do
  if [ $L = default ]; then unset F3D_COMPONENT_LANES; else export F3D_COMPONENT_LANES=$L; fi
  echo "== F3D_COMPONENT_LANES=$L" | tee -a $OUT/lanes.log
  timeout -k 10 300 python3 tools/trace_size.py --config c2 --reps 5 | tee -a $OUT/lanes.log
  timeout -k 10 300 python3 tools/trace_size.py --config c3 --reps 5 | tee -a $OUT/lanes.log
  timeout -k 10 300 python3 tools/trace_size.py --size 256 --reps 3 | tee -a $OUT/lanes.log
  timeout -k 10 300 python3 tools/trace_size.py --size 512 --reps 2 | tee -a $OUT/lanes.log
done
