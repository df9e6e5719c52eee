#!/bin/bash
# round 4, GPU job 36: the bench as the driver runs it and the round's rocprofv3 evidence (tools/profile_round.sh) on ONE box, final tree
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job36
mkdir -p $O
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
tail -1 $O/bench_steps20_warmup5.json | cut -c1-200
F3D_OUT=$O timeout -k 10 800 bash tools/profile_round.sh > $O/profile_round.log 2>&1 || { tail -30 $O/profile_round.log; exit 1; }
tail -3 $O/profile_round.log
