#!/bin/bash
# round 4, GPU job 24: out-of-core 1024^3 on 16 GB: transparent huge pages for the 34 GB of host scratch (madvise before anything touches
# them) against 4 KiB pages, with the scratch prepared on the helper thread and in line
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job24
mkdir -p $O
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag > $O/thp_mode.txt 2>&1 || true
grep -i -E "AnonHugePages|MemFree|MemTotal" /proc/meminfo >> $O/thp_mode.txt
run() {
  tag=$1; shift
  env "$@" timeout -k 10 600 python3 tools/pbench.py --size 1024 --budget-mb 16384 $CHK > $O/out.txt 2>&1 || { tail -20 $O/out.txt; exit 1; }
  echo "== $tag: $*" >> $O/huge_pages.txt
  grep -E "piecemeal:|frames |identical|DIFFER" $O/out.txt >> $O/huge_pages.txt
}
CHK="--no-resident"
run "4 KiB pages, scratch in line" F3D_P_HUGE_PAGES=0 F3D_P_SCRATCH_THREAD=0
run "huge pages, scratch in line" F3D_P_HUGE_PAGES=1 F3D_P_SCRATCH_THREAD=0
run "4 KiB pages, scratch on the helper thread" F3D_P_HUGE_PAGES=0 F3D_P_SCRATCH_THREAD=1
CHK="--check"
run "huge pages, scratch on the helper thread (default)" F3D_DUMMY=1
cat $O/thp_mode.txt; cut -c1-250 $O/huge_pages.txt
