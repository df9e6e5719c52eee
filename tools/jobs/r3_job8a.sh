#!/bin/bash
# round 3, GPU job 8a: the whole GPU suite, smoke, the bench as the driver runs it, and `bench.py --gpus 2` started bare (rehearsal)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job34
mkdir -p $O
python3 -X faulthandler -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err || { tail -20 $O/bench_steps20.err; exit 1; }
python3 - <<PY
import json
b=json.loads(open("$O/bench_steps20.json").read().strip().splitlines()[-1])
r=b["roofline"]
print("value", b["value"], "ms", b["ms_per_step"], "whole", b["whole_run_roofline_frac"], "pair", r["frac"], "finest", r["finest_level"]["frac"], "sp", r["sweep_phi_ksi"]["achieved"], "parity", b["parity"]["match"])
print("host_inclusive", b["host_inclusive"]["value"], b["host_inclusive"]["steps"], "configs", [(c["ms_per_step"], c["value"]) for c in b.get("configs",[])])
print("cpu", b.get("cpu_baseline",{}).get("value"), b.get("fixed_sample"))
PY
F3D_COMM_BACKEND=shm F3D_SHM_CAP_MB=512 python3 bench.py --gpus 2 --size 512 --steps 1 --warmup 1 > $O/bench_gpus2_shm.json 2> $O/bench_gpus2_shm.err || { tail -30 $O/bench_gpus2_shm.err; exit 1; }
python3 - <<PY
import json
b=json.loads(open("$O/bench_gpus2_shm.json").read().strip().splitlines()[-1])
print("gpus2 rehearsal: value", b["value"], "n_gpus", b["n_gpus"], "launched_by", b["launched_by"], "parity", b["parity"]["match"], "rccl_ranks", b["rccl_ranks"], b["comm"]["halo_GB_sent_per_step"], "cpu_baseline" in b, "fixed_sample" in b, b["roofline"].get("hbm_frac"))
PY
