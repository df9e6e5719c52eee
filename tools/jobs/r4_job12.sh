#!/bin/bash
# round 4, GPU job 12: the bench line with plain stores (ab_plain/lib = the library of the commit before) against non-temporal stores,
# alternating inside one call -- job 11's box gave another balance between the two fused launches than job 5's, and boxes differ
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job12
mkdir -p $O
for round in 1 2 3; do
  for lib in ab_plain/lib cuda-flow3d_amd/lib; do
    tag=$(echo $lib | tr '/' '_')
    F3D_LIBDIR=$R/$lib timeout -k 10 300 python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu > $O/bench_${tag}_$round.json 2> $O/bench_${tag}_$round.err || { tail -20 $O/bench_${tag}_$round.err; exit 1; }
    python3 - $O/bench_${tag}_$round.json $lib $round <<'PY' | tee -a $O/store_policy_bench_ab.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:24s} round {sys.argv[3]}: {d['value']:.2f} Mvox/s {d['ms_per_step']:.1f} ms | two sweeps {r['frac']:.4f} over the pyramid, finest {r['finest_level']['avg_launch_us']:.0f} us | "
      f"sweep+phi/ksi {r['sweep_phi_ksi']['achieved']:.0f} GB/s, finest {r['sweep_phi_ksi']['finest_level']['avg_launch_us']:.0f} us | all solver launches {r['all_solver_launches']['frac']:.4f}")
PY
  done
done
