#!/bin/bash
# round 4, GPU job 5: (1) whole GPU suite (1024^3 reference run left out); (2) config 3 with thin levels on the frame launches (y march,
# tile without halo rows) against the switches that take them back; (3) A/B of the 512^3 solve: the library before k_pair8's body
# became a device function (ab_old/lib, commit 63eb758) against the current one, alternating, same call; (4) the counter record on the
# current kernels; (5) the bench as the driver runs it
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job5
mkdir -p $O
F3D_REF_C5=0 timeout -k 10 1000 python3 -X faulthandler -m pytest tests -q -m gpu -x --durations=6 > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -10 $O/tests.log
for env in "F3D_DUMMY=1" "F3D_PAIR8_TIGHT=0" "F3D_PAIR8_YMARCH=0" "F3D_FRAME_DERIVATIVES=0 F3D_PAIR8_YMARCH=0"; do
  echo "== c3  $env" >> $O/thin_solves.txt
  env $env timeout -k 10 300 python3 tools/trace_size.py --config c3 --reps 8 2>&1 | tail -1 >> $O/thin_solves.txt
done
cat $O/thin_solves.txt
for round in 1 2; do
  for lib in ab_old/lib cuda-flow3d_amd/lib; do
    echo "== 512^3  $lib (round $round)" >> $O/ab_body.txt
    F3D_LIBDIR=$R/$lib timeout -k 10 300 python3 tools/trace_size.py --size 512 --reps 4 2>&1 | tail -1 >> $O/ab_body.txt
  done
done
cat $O/ab_body.txt
F3D_OUT=$O timeout -k 10 900 bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 || { tail -30 $O/pmc_traffic.log; exit 1; }
cp $O/traffic/pmc_traffic.json profiles/r04_pmc_traffic.json
grep -A1 '"_solver_kernels_sha16"' $O/traffic/pmc_traffic.json | head -2
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
tail -1 $O/bench_steps20_warmup5.json | cut -c1-300
