#!/bin/bash
# HBM traffic of the solver kernels on one 512^3 level (tools/kbench.py): FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes (MI355X_MICROARCH.md, HBM section), plus the same two counters on lab kernels with a known byte
# count (tools/lab/stream_lab: 10 x 512^3 floats read once, 3 x written once) as the calibration of the gfx950 factor.
set -e
R=$(pwd)   # the tree the command was started in (a staged copy under tools/gpu_stage.sh)
O=${F3D_OUT:-$R/gpurun_out}/traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- $R/tools/lab/bin/stream_lab > $O/cal_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_write -- $R/tools/lab/bin/stream_lab > $O/cal_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/k_fetch -- python3 $R/tools/kbench.py --size 512 --reps 3 > $O/k_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/k_write -- python3 $R/tools/kbench.py --size 512 --reps 3 > $O/k_write.log 2>&1
# the frame-derivative builds of the two fused launches (what the resident operator runs by default since round 3)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fd_fetch -- python3 $R/tools/kbench.py --size 512 --reps 3 --kernel bothfd > $O/fd_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/fd_write -- python3 $R/tools/kbench.py --size 512 --reps 3 --kernel bothfd > $O/fd_write.log 2>&1
python3 - <<PY
import csv, glob, collections, json
res = {}
for d in ("cal_fetch", "cal_write", "k_fetch", "k_write", "fd_fetch", "fd_write"):
    f = glob.glob("$O/" + d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("k_pair8ILi0", "k_pair8ILi1", "k_pair8<0", "k_pair8<1", "k_sweep7", "k_sweep6", "k_phiksi6", "k_flat", "k_march_packed", "k_march"):
            if key in n:
                if key.startswith("k_pair8"):
                    key = "k_pair8" if key.endswith("0") else "k_pair8_sweep_phi_ksi"
                    # k_pair8<MODE, TY, ABL, FD, YM>: the fourth template argument tells the frame-derivative builds apart
                    m = __import__("re").search(r"k_pair8<\d+, \d+, \d+, (true|false)", n) or __import__("re").search(r"k_pair8ILi\d+ELi\d+ELi\d+ELb([01])", n)
                    if m and m.group(1) in ("true", "1"):
                        key += "_fd"
                if key == "k_flat":
                    key = "k_flat<float4>" if "float4" in n or "HIP_vector" in n else "k_flat<float>"
                agg[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
                break
    for (k, c), v in sorted(agg.items()):
        res.setdefault(k, {})[c] = sum(v) / len(v)
        print(f"{d:10s} {k:18s} {c:11s} launches {len(v):3d}  avg {sum(v) / len(v):.6g} KiB")
import hashlib
import sys
sys.path.insert(0, "$R")
sys.argv = ["bench.py"]
import bench  # the stamp bench.py checks the record against: the machine code of the solver kernels in the library that just ran
res["_solver_kernels_sha16"] = bench.solver_kernel_stamp()
json.dump(res, open("$O/traffic_raw.json", "w"), indent=1)
# the record bench.py reads (profiles/rNN_pmc_traffic.json): HBM bytes per 512^3 launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
S = 512
alg = {"k_pair8": 104.0, "k_pair8_sweep_phi_ksi": 92.0, "k_pair8_fd": 104.0, "k_pair8_sweep_phi_ksi_fd": 92.0, "k_sweep7": 104.0,
       "k_sweep6": 52.0, "k_phiksi6": 40.0}
out = {"_note": "HBM bytes per 512^3 launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                "separate passes over tools/kbench.py --size 512 (tools/pmc_traffic.sh); both counters are in KiB; FETCH_SIZE is doubled as "
                "MI355X_MICROARCH.md prescribes for gfx950, and the same run calibrates it on tools/lab/stream_lab (k_flat reads 10 x 512 MiB, "
                "writes 3 x 512 MiB).",
       "_size": S, "_solver_kernels_sha16": res["_solver_kernels_sha16"],
       "_calibration": {k: v for k, v in res.items() if k.startswith("k_flat")}}
for k, b in alg.items():
    if k in res and "FETCH_SIZE" in res[k] and "WRITE_SIZE" in res[k]:
        hbm = (2 * res[k]["FETCH_SIZE"] + res[k]["WRITE_SIZE"]) * 1024
        out[k] = {"fetch_size_kib_reported": res[k]["FETCH_SIZE"], "write_size_kib": res[k]["WRITE_SIZE"], "hbm_bytes_per_launch": int(hbm),
                  "algorithmic_bytes_per_launch": int(b * S ** 3), "ratio": round(hbm / (b * S ** 3), 3)}
json.dump(out, open("$O/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
PY
rm -rf $O/cal_fetch $O/cal_write $O/k_fetch $O/k_write $O/fd_fetch $O/fd_write
