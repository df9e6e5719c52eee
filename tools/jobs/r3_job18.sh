#!/bin/bash
# do launches on two lanes fill each other's last rounds?  (tools/two_chain_lab.py)
set -e
OUT=${F3D_OUT:-gpurun_out}/r3/job18
mkdir -p $OUT
for n in 512 463 439 397 340 307 263 226 184; do
  timeout -k 10 200 python3 tools/two_chain_lab.py --size $n 2>&1 | grep "two sweeps" | tee -a $OUT/lab.log
done
