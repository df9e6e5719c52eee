#!/bin/bash
# round 3, GPU job 4: exact short road to the weights -- exhaustive self-test, kernel parity, A/B timing, C4 digest switches
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job4
mkdir -p $O
python3 -X faulthandler -m pytest tests/test_gpu_kernels.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for s in 512 256; do
  for lib in ab_base/cuda-flow3d_amd/lib cuda-flow3d_amd/lib; do
    echo "== $lib size $s" >> $O/kb.log
    F3D_LIBDIR=$R/$lib python3 tools/kbench.py --size $s --reps 20 --kernel sweeppk 2>&1 | grep -v "^\[" >> $O/kb.log
  done
done
cat $O/kb.log
python3 -X faulthandler -m pytest tests/test_gpu_configs.py tests/test_gpu_pipeline.py -q -m gpu -x -k "not 1024" > $O/tests_cfg.log 2>&1 || { tail -30 $O/tests_cfg.log; exit 1; }
tail -2 $O/tests_cfg.log
python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<PY
import json
b=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
r=b["roofline"]
print("value", b["value"], "ms", b["ms_per_step"], "pair frac", r["frac"], "sp", r["sweep_phi_ksi"], "parity", b["parity"]["match"], "configs", [(c["ms_per_step"]) for c in b.get("configs",[])])
PY
