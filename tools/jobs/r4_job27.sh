#!/bin/bash
# round 4, GPU job 27: out-of-core 1024^3 on other budgets (8, 32, 64 GB): other plans -- more residencies, constants held on more
# levels, fewer chunked levels -- each checked against the resident driver's result
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job27
mkdir -p $O
for mb in 8192 32768 65536; do
  echo "== budget $mb MB" >> $O/budgets.txt
  timeout -k 10 600 python3 tools/pbench.py --size 1024 --budget-mb $mb --check --verbose > $O/out_$mb.txt 2>&1 || { tail -20 $O/out_$mb.txt; exit 1; }
  grep -E "solver of level|piecemeal:|frames |identical|DIFFER" $O/out_$mb.txt >> $O/budgets.txt
done
cut -c1-230 $O/budgets.txt
