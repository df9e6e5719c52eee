#!/bin/bash
# round 3, GPU job 8c: kernel time of the 8-slab decomposition at 512^3 under both exchange orders (batched plane copies), out-of-core 1024^3
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job8c
mkdir -p $O
tools/lab/bin/issue_cost_lab > $O/issue_cost.log 2>&1
cat $O/issue_cost.log
for abl in 0 16 0 16; do
  echo "== F3D_ABLATE8=$abl" >> $O/kb_abl.log
  F3D_ABLATE8=$abl python3 tools/kbench.py --size 512 --reps 20 --kernel sweep2 2>&1 | grep -v "^\[" >> $O/kb_abl.log
  F3D_ABLATE8=$abl python3 tools/kbench.py --size 512 --reps 20 --kernel sweeppk 2>&1 | grep -v "^\[" >> $O/kb_abl.log
done
cat $O/kb_abl.log
python3 tools/pbench.py --size 1024 --budget-mb 16384 --no-resident > $O/pbench1024.log 2>&1 || true
tail -2 $O/pbench1024.log
python3 tools/pbench.py --size 512 --budget-mb 8192 --no-resident > $O/pbench512_8g.log 2>&1 || true
tail -2 $O/pbench512_8g.log
cd /tmp && export TMPDIR=/tmp
for mode in outer stage; do
  F3D_SLAB_EXCHANGE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $O/slab_$mode -- python3 $R/tools/slab8_profile.py --size 512 --only slabs > $O/slab_$mode.log 2>&1
  cp $(ls $O/slab_$mode/*/*_kernel_stats.csv | head -1) $O/slab8_512_${mode}_kernel_stats.csv
  rm -rf $O/slab_$mode
  grep "per solve" $O/slab_$mode.log
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/unsplit -- python3 $R/tools/slab8_profile.py --size 512 --only unsplit > $O/unsplit.log 2>&1
cp $(ls $O/unsplit/*/*_kernel_stats.csv | head -1) $O/unsplit_512_kernel_stats.csv
rm -rf $O/unsplit
grep "per solve" $O/unsplit.log
