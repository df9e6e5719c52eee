#!/bin/bash
# round 3, GPU job 47: BASELINE config 5 (1024^3) on ONE GPU on the final tree: the N = 1 point the multi-GPU runs divide, and the
# 8-slab decomposition of the same solve run in one process (tools/slab8_profile.py)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job47
mkdir -p $O
python3 bench.py --size 1024 --steps 2 --warmup 1 --no-extra > $O/bench_1024.json 2> $O/bench_1024.err || { tail -20 $O/bench_1024.err; exit 1; }
python3 -c "
import json
b=json.loads(open('$O/bench_1024.json').read().strip().splitlines()[-1]); r=b['roofline']
print('C5 one GPU: value', b['value'], 'ms', b['ms_per_step'], 'pair frac', r['frac'], 'finest us', r['finest_level']['avg_launch_us'], 'parity', b['parity'])" | tee $O/c5.log
python3 tools/slab8_profile.py --size 1024 --only slabs 2>&1 | tail -1 | tee -a $O/c5.log
