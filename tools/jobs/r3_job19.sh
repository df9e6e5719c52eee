#!/bin/bash
# a level cut into n z windows launched one after the other against one launch (tools/zsplit_lab.py)
set -e
OUT=${F3D_OUT:-gpurun_out}/r3/job19
mkdir -p $OUT
for n in 512 487 463 439 418 397 377 358 340 323 307 292 277 263 250 238 226 204 184 166 150 128; do
  timeout -k 10 200 python3 tools/zsplit_lab.py --size $n --reps 12 2>&1 | grep "\^3" | tee -a $OUT/lab.log
done
