#!/bin/bash
# does cutting a level into z windows change what the two-sweep launch fetches?  FETCH_SIZE / TCC hit rate of one 418^3 level as
# one launch and as three (tools/zsplit_lab.py)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job20
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 1 3; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch$n -- python3 $R/tools/zsplit_lab.py --size 418 --reps 4 --splits $n > $O/fetch$n.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2$n -- python3 $R/tools/zsplit_lab.py --size 418 --reps 4 --splits $n > $O/l2$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in ("fetch1", "fetch3", "l21", "l23"):
    f = glob.glob("$O/" + d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = "SS" if ("k_pair8<0" in n or "k_pair8ILi0" in n) else ("SP" if "k_pair8" in n else None)
        if key:
            agg[(key, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
    for k in sorted(agg):
        print(d, k, "launches", cnt[k], "total %.6g" % agg[k])
PY
