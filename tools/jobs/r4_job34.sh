#!/bin/bash
# round 4, GPU job 34: the final tree (the drivers page-lock only volumes of 32 MiB and more): the suite exactly as the driver runs it, the
# smoke entry, then eight minutes of the out-of-core soak (page-locking forced for every size, every array in a mapping of its own)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job34
mkdir -p $O
timeout -k 10 900 python3 -X faulthandler -m pytest tests -q -m gpu -x --durations=4 > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -7 $O/tests.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
MALLOC_MMAP_THRESHOLD_=131072 timeout -k 10 600 python3 tools/soak_piecemeal.py 480 23 > $O/soak_raw.txt 2>&1 || { grep -a -E "^RUN|soak:|MISMATCH|rror|fault|Low GPU" $O/soak_raw.txt | tail -12; exit 1; }
grep -a -o -E "\[ *[0-9]+ s\] [0-9]+ volumes, [0-9]+ out-of-core runs checked, [0-9]+ mismatches|soak: .*|MISMATCH.*|.*Low GPU.*" $O/soak_raw.txt > $O/soak_piecemeal.txt
tail -3 $O/soak_piecemeal.txt
rm -f $O/soak_raw.txt
