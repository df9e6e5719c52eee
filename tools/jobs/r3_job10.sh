#!/bin/bash
# round 3, GPU job 10: frame-derivative builds with the split ring (12-row tiles): parity, then timing against the frame builds
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job10
mkdir -p $O
python3 -X faulthandler -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "frame_derivatives or fused or thin" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for s in 512 384 256 128; do
  for k in sweep2 sweep2fd sweeppk sweeppkfd; do
    python3 tools/kbench.py --size $s --reps 20 --kernel $k 2>&1 | grep -v "^\[" >> $O/kb.log
  done
done
cat $O/kb.log
F3D_FRAME_DERIVATIVES=1 python3 -X faulthandler -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -m gpu -x -k "not 1024" > $O/tests_fd.log 2>&1 || { tail -40 $O/tests_fd.log; exit 1; }
tail -2 $O/tests_fd.log
F3D_FRAME_DERIVATIVES=1 python3 bench.py --steps 3 --warmup 1 --no-extra > $O/bench_fd.json 2> $O/bench_fd.err || { tail -20 $O/bench_fd.err; exit 1; }
python3 bench.py --steps 3 --warmup 1 --no-extra > $O/bench_nofd.json 2> $O/bench_nofd.err
python3 - <<PY
import json
for t in ("fd","nofd"):
    b=json.loads(open("$O/bench_%s.json"%t).read().strip().splitlines()[-1])
    print(t, "value", b["value"], "ms", b["ms_per_step"], "parity", b["parity"]["match"])
PY
