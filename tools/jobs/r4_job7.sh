#!/bin/bash
# round 4, GPU job 7: rehearsals of the multi-GPU bench line the way the driver starts it -- under torch.distributed.run, two and four
# rank processes on the box's one GPU over the shared-memory transport (orchestration, not numbers) -- and the rank-process tests by name
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job7
mkdir -p $O
timeout -k 10 600 python3 -X faulthandler -m pytest tests/test_gpu_slab_procs.py -q -m gpu -x -v > $O/tests_procs.log 2>&1 || { tail -40 $O/tests_procs.log; exit 1; }
grep -E "PASSED|FAILED|passed|failed" $O/tests_procs.log | tail -16
export F3D_COMM_BACKEND=shm F3D_SHM_CAP_MB=512
for n in 2 4; do
  timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --size 256 --steps 2 --warmup 1 --no-extra > $O/bench_torchrun_gpus$n.json 2> $O/bench_torchrun_gpus$n.err || { tail -40 $O/bench_torchrun_gpus$n.err; exit 1; }
  tail -1 $O/bench_torchrun_gpus$n.json | cut -c1-400
done
