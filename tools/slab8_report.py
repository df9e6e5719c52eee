#!/usr/bin/env python3
"""profiles/<tag>_slab8_onegpu_<S>.{md,json} from the two per-level tables tools/level_table.py wrote for
`tools/slab8_profile.py --size S --only slabs` and `--only unsplit` (tools/r3_job1.sh): kernel time of the 8 slabs run one
after the other on ONE GPU divided by the kernel time of the unsplit solve, per pyramid level.  1.00 = the decomposition
costs nothing (an 8-GPU run would divide the kernel time by eight), above it = redundant planes of the communication-avoiding
windows, the zone launches, and the launch-latency floor every rank pays.  tools/scale_model.py reads the JSON."""
import json, os, sys
slabs, unsplit, size, tag, walls = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5:]
a = json.load(open(slabs))["levels"]
b = json.load(open(unsplit))["levels"]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import math
dims = [math.ceil(size * 0.95 ** l) for l in range(len(a))][::-1]   # level 0 of the tables = coarsest
rows, ts, tu = [], 0.0, 0.0
for x, y, d in zip(a, b, dims):
    solver = lambda r: r["pair_ss_us"] + r["pair_sp_us"] + r["phi_ksi_us"] + r["sweep_us"]
    rows.append({"level_index": x["level_index"], "edge": d, "slabs_ms": round(x["total_us"] / 1e3, 2),
                 "unsplit_ms": round(y["total_us"] / 1e3, 2), "ratio": round(x["total_us"] / y["total_us"], 3),
                 "solver_ratio": round(solver(x) / solver(y), 3),
                 "two_sweep_ratio": round(x["pair_ss_us"] / y["pair_ss_us"], 3) if y["pair_ss_us"] else None})
    ts += x["total_us"]
    tu += y["total_us"]
doc = {"size": size, "ranks": 8, "slabs_kernel_s": round(ts / 1e6, 3), "unsplit_kernel_s": round(tu / 1e6, 3),
       "ratio": round(ts / tu, 4), "wall": walls, "levels": rows}
json.dump(doc, open(os.path.join(root, "profiles", f"{tag}_slab8_onegpu_{size}.json"), "w"), indent=1)
out = [f"# {tag}: the 8-slab decomposition of the {size}^3 run on ONE GPU, per pyramid level", "",
       f"`rocprofv3 --kernel-trace --stats -- python3 tools/slab8_profile.py --size {size} --only slabs|unsplit`, one solve after a",
       "warm-up each, kernel times summed per level by `tools/level_table.py` (levels are told apart by the one warp per level",
       "and rank).  All eight ranks live in one process and run one after the other (plane copies instead of RCCL), so the slab",
       "column is the kernel work an 8-GPU run divides by eight.", "",
       f"Whole solve: 8 slabs {ts / 1e6:.2f} s of kernel time, unsplit {tu / 1e6:.2f} s: **{ts / tu:.3f} x** ({'; '.join(walls)}).", "",
       "| level (coarsest first) | edge | 8 slabs ms | unsplit ms | ratio | solver kernels only | two-sweep launches only |", "|---|---|---|---|---|---|---|"]
for r in rows:
    out.append(f"| {r['level_index']} | {r['edge']} | {r['slabs_ms']} | {r['unsplit_ms']} | {r['ratio']} | {r['solver_ratio']} | {r['two_sweep_ratio']} |")
open(os.path.join(root, "profiles", f"{tag}_slab8_onegpu_{size}.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
