#!/bin/bash
# round 3, GPU job 40: what the prologue of a fused launch costs on small levels, and whether it is bound by the issue of its ~70 DMA
# instructions (one loader wave) or by their latency: rocprofv3 kernel durations of the two-sweep launch at 24^3, 48^3, 96^3 --
# whole kernel (F3D_ABLATE8=0), compute waves that only keep the barriers (4), prologue only (8), prologue of ONE plane only (40)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job40
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 24 48 96; do
  for abl in 0 4 8 40; do
    export F3D_ABLATE8=$abl
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_${n}_$abl -- python3 $R/tools/kbench.py --size $n --reps 300 --kernel sweep2 > $O/kb_${n}_$abl.log 2>&1
    f=$(ls $O/t_${n}_$abl/*/*_kernel_stats.csv | head -1)
    python3 - "$f" $n $abl >> $O/prologue.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_pair8" in r["Name"]:
        print(f"{sys.argv[2]}^3 ABLATE8={sys.argv[3]:>2}: {r['Name'][28:52]} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.2f} us min {float(r['MinNs'])/1e3:.2f} us")
PY
    rm -rf $O/t_${n}_$abl
  done
done
unset F3D_ABLATE8
cat $O/prologue.txt
