#!/usr/bin/env python3
"""Condenses a tools/profile_round.sh output directory into profiles/<tag>_*.{csv,json,md} (the committed evidence)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)
bench = json.loads(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, f"{tag}_bench_default.json"), "w"), indent=1)
newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)  # the directory keeps the files of earlier runs
stats = newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, f"{tag}_bench512_rocprofv3_kernel_stats.csv"))
lines = [f"# {tag}: rocprofv3 evidence for `python bench.py` (512^3, full default pyramid, 1x MI355X)", ""]
lines += ["## bench line", "```json", json.dumps(bench), "```", ""]
lines += ["## rocprofv3 --kernel-trace --stats  -- python3 bench.py --steps 1 --warmup 0 --no-cpu", "",
          "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in csv.DictReader(open(stats)):
    lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
lines += ["", "## PMC passes on tools/kbench.py --size 512 (one 512^3 level; separate rocprofv3 --pmc runs), averages per launch", "",
          "| kernel | counter | value |", "|---|---|---|"]
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(max(f, key=os.path.getmtime))):
        kn = r["Kernel_Name"]
        fd = ", true, " in kn and "k_pair8<" in kn   # k_pair8<MODE, TY, ABL, FD, YM>
        for name, key in (("k_pair8<0", "k_pair8 (two sweeps%s)" % (", frame derivatives" if fd else "")),
                          ("k_pair8<1", "k_pair8 (sweep + phi/ksi%s)" % (", frame derivatives" if fd else "")), ("k_sweep7", "k_sweep7"),
                          ("k_sweep6", "k_sweep6"), ("k_phiksi6", "k_phiksi6")):
            if name in kn:
                agg[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
                break
    for (k, c), v in sorted(agg.items()):
        lines.append(f"| {k} | {c} | {sum(v) / len(v):.6g} |")
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-30:]))
