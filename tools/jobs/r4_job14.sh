#!/bin/bash
# round 4, GPU job 14: is the round model's choice per level the best one INSIDE the 512^3 pyramid?  Per-level times of the fused
# launches (rocprofv3 kernel trace of tools/trace_size.py --size 512, tools/level_table.py) with the default plan, with the tile height
# pinned to 8 and to 12 rows, and with the model's per-chunk overhead set to 4, 10 and 14 plane steps instead of 7 (fewer / more
# z-chunks); then a 6-minute soak of the shipped kernels (non-temporal stores)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job14
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for tag in default ty8 ty12 steps4 steps10 steps14; do
  unset F3D_PAIR8_TY F3D_PAIR8_CHUNK_STEPS
  case $tag in
    ty8) export F3D_PAIR8_TY=8 ;;
    ty12) export F3D_PAIR8_TY=12 ;;
    steps4) export F3D_PAIR8_CHUNK_STEPS=4 ;;
    steps10) export F3D_PAIR8_CHUNK_STEPS=10 ;;
    steps14) export F3D_PAIR8_CHUNK_STEPS=14 ;;
  esac
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -- python3 $R/tools/trace_size.py --size 512 --reps 2 > $O/$tag.log 2>&1 || { tail -20 $O/$tag.log; exit 1; }
  tail -1 $O/$tag.log
  t=$(ls $O/$tag/*/*_kernel_trace.csv | head -1)
  python3 $R/tools/level_table.py $t --levels 40 --out $O/${tag}_levels.json > $O/${tag}_levels.md
  rm -rf $O/$tag
done
unset F3D_PAIR8_TY F3D_PAIR8_CHUNK_STEPS
cd $R
python3 - $O <<'PY' | tee $O/plan_per_level.txt
import json, sys, math
O = sys.argv[1]
tags = ["default", "ty8", "ty12", "steps4", "steps10", "steps14"]
T = {t: json.load(open(f"{O}/{t}_levels.json"))["levels"] for t in tags}
edges = [math.ceil(512 * 0.95 ** l) for l in range(40)][::-1]
print("level edge | fused launches (two sweeps + sweep/phi/ksi) ms: " + "  ".join(tags) + " | best")
tot = {t: 0.0 for t in tags}; best_tot = 0.0
for i, e in enumerate(edges):
    v = {t: (T[t][i]["pair_ss_us"] + T[t][i]["pair_sp_us"]) / 1e3 for t in tags}
    for t in tags: tot[t] += v[t]
    b = min(v, key=v.get); best_tot += v[b]
    print(f"{i:2d} {e:4d} | " + "  ".join(f"{v[t]:8.3f}" for t in tags) + f" | {b} ({100 * (v[b] / v['default'] - 1):+.1f} %)")
print("sum       | " + "  ".join(f"{tot[t]:8.1f}" for t in tags) + f" | best per level {best_tot:.1f} ({100 * (best_tot / tot['default'] - 1):+.2f} %)")
PY
timeout -k 10 500 python3 tools/soak_fused.py 360 > $O/soak.txt 2>&1 || { tail -20 $O/soak.txt; exit 1; }
tail -3 $O/soak.txt
