#!/bin/bash
# SQ-level counters of one solver kernel on a 512^3 level (tools/kbench.py), separate rocprofv3 --pmc passes.
#   tools/pmc_sq.sh <kernel: sweep|sweep2|phi> <tag>
set -e
R=$(pwd)   # the tree the command was started in (a staged copy under tools/gpu_stage.sh)
K=${1:-sweep2}
T=${2:-sq}
O=${F3D_OUT:-$R/gpurun_out}/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python3 $R/tools/kbench.py --size 512 --reps 3 --kernel $K > $O/$n.log 2>&1
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
pass b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
pass c SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM
pass d GRBM_GUI_ACTIVE GRBM_COUNT
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("k_sweep", "k_phiksi", "k_pair8")):
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$O/summary.csv", "w") as out:
    out.write("kernel,counter,launches,avg\n")
    for (k, c), v in sorted(agg.items()):
        out.write(f"{k},{c},{len(v)},{sum(v) / len(v):.6g}\n")
print(open("$O/summary.csv").read())
PY
rm -rf $O/a $O/b $O/c
