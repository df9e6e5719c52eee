#!/bin/bash
# round 3, GPU job 7: out-of-core solver with the fused last sweep (parity, then 1024^3 on a 16 GB budget), kernel time of the
# 8-slab decomposition at 512^3 under both exchange orders
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job7
mkdir -p $O
python3 -X faulthandler -m pytest tests/test_gpu_piecemeal.py -q -m gpu > $O/tests.log 2>&1 || { tail -60 $O/tests.log; }
tail -2 $O/tests.log
python3 tools/pbench.py --size 512 --budget-mb 4096 --check > $O/pbench512.log 2>&1 || true
tail -4 $O/pbench512.log
python3 tools/pbench.py --size 1024 --budget-mb 16384 --no-resident > $O/pbench1024.log 2>&1 || true
tail -4 $O/pbench1024.log
F3D_FUSED_PHI_KSI=0 python3 tools/pbench.py --size 1024 --budget-mb 16384 --no-resident > $O/pbench1024_unfused.log 2>&1 || true
tail -2 $O/pbench1024_unfused.log
cd /tmp && export TMPDIR=/tmp
for mode in outer stage; do
  F3D_SLAB_EXCHANGE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $O/slab_$mode -- python3 $R/tools/slab8_profile.py --size 512 --only slabs > $O/slab_$mode.log 2>&1
  cp $(ls $O/slab_$mode/*/*_kernel_stats.csv | head -1) $O/slab8_512_${mode}_kernel_stats.csv
  rm -rf $O/slab_$mode
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/unsplit -- python3 $R/tools/slab8_profile.py --size 512 --only unsplit > $O/unsplit.log 2>&1
cp $(ls $O/unsplit/*/*_kernel_stats.csv | head -1) $O/unsplit_512_kernel_stats.csv
rm -rf $O/unsplit
ls $O
