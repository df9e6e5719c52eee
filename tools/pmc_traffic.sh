#!/bin/bash
# HBM traffic of the whole bench run (per-dispatch FETCH_SIZE / WRITE_SIZE, separate passes) + calibration of FETCH_SIZE
# on lab kernels with a known byte count (10 x 512^3 floats read once, 3 x written once).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- $R/tools/lab/bin/stream_lab > $O/cal_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -- $R/tools/lab/bin/stream_lab > $O/cal_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/bench_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $O/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/bench_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > $O/bench_write.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("cal_fetch", "cal_write", "bench_fetch", "bench_write"):
    f = glob.glob("$O/" + d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    with open("$O/" + d + "_summary.csv", "w") as out:
        out.write("kernel,counter,dispatches,sum,avg_per_dispatch\n")
        for (k, c), (s, n) in sorted(agg.items()):
            out.write(f'"{k}",{c},{n},{s:.6g},{s / n:.6g}\n')
    print(open("$O/" + d + "_summary.csv").read())
PY
rm -rf $O/bench_fetch $O/bench_write $O/cal_fetch $O/cal_write
