#!/bin/bash
# round 3, GPU job 6: lanes (two drivers side by side), per-stage exchange on the GPU, slab8 one-GPU figure under both exchange orders
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job6
mkdir -p $O
python3 -X faulthandler -m pytest tests/test_gpu_pipeline.py tests/test_gpu_slab.py tests/test_gpu_slab_procs.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for s in 64 128 192; do python3 tools/sequence_bench.py --size $s --frames 9 --concurrent 2 3 4 >> $O/seq.log 2>&1; done
cat $O/seq.log
for mode in outer stage; do
  F3D_SLAB_EXCHANGE=$mode python3 tools/slab8_profile.py --size 512 --only slabs --reps 2 2>&1 | tail -1 | sed "s/^/exchange=$mode /" >> $O/slab8.log
done
python3 tools/slab8_profile.py --size 512 --only unsplit --reps 2 2>&1 | tail -1 >> $O/slab8.log
cat $O/slab8.log
