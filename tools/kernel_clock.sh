#!/bin/bash
# Effective shader clock of the solver kernels: GRBM_GUI_ACTIVE / 8 / kernel duration (MI355X_MICROARCH.md, DVFS give-back)
# over tools/kbench.py launches.   tools/kernel_clock.sh <kernel: sweep|sweep2|sweeppk|phi> <tag> [env assignments...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=${1:-sweep2}; T=${2:-clk}; shift 2
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/run -- python3 $R/tools/kbench.py --size 512 --reps 5 --kernel $K > $O/run.log 2>&1
python3 - <<PY
import csv, glob
cc = glob.glob("$O/run/*/*counter_collection.csv")[0]
kt = glob.glob("$O/run/*/*kernel_trace.csv")[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt))}
for r in csv.DictReader(open(cc)):
    d, n = dur[r["Dispatch_Id"]]
    if d > 300000:
        print("%-60s %8.1f us  %.3f GHz" % (n.split("(")[0][-60:], d / 1e3, float(r["Counter_Value"]) / 8 / d))
PY
rm -rf $O/run
