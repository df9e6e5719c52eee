#!/usr/bin/env python3
"""Per-pyramid-level kernel time from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`).

The drivers launch exactly one warp (`k_warp`) per level and rank, so dispatches sorted by start time fall into levels at
every `--warps-per-level`-th `k_warp` (1 for the resident driver, the rank count for the one-process z-slab driver); the solves
of a trace are told apart by the level counter wrapping at --levels.  Prints a table and writes JSON:
  python3 tools/level_table.py trace.csv [--warps-per-level 8] [--levels 40] [--skip-solves 1] [--out file.json]
Columns: microseconds per level of the two-sweep launches (k_pair8<0>), sweep + next phi/ksi (k_pair8<1>), phi/ksi alone,
one sweep, everything else; `n` = solver launches."""
import argparse, collections, csv, json, sys

CLASSES = (("k_pair8<0", "pair_ss"), ("k_pair8<1", "pair_sp"), ("k_pair8t<0", "pair_ss"), ("k_pair8t<1", "pair_sp"), ("k_phiksi6", "phi_ksi"), ("k_sweep6", "sweep"), ("k_sweep7", "pair7"))


def classify(name):
    for pat, key in CLASSES:
        if pat in name:
            return key
    return "other"


def levels_of(path, warps_per_level, n_levels, skip_solves):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    counts = collections.defaultdict(lambda: collections.defaultdict(int))
    warps = 0
    level = -1          # dispatches before the first warp (upload, blur) are not part of a level
    for s, e, name in rows:
        if "k_warp" in name:
            if warps % warps_per_level == 0:
                level += 1
            warps += 1
        if level < 0:
            continue
        solve, lv = divmod(level, n_levels)
        if solve < skip_solves:
            continue
        key = classify(name)
        out[(solve, lv)][key] += (e - s) * 1e-3
        if key != "other":
            counts[(solve, lv)][key] += 1
    # average over the solves kept
    table = []
    solves = sorted({s for s, _ in out})
    for lv in range(n_levels):
        row = {"level_index": lv}
        for _, key in CLASSES + (("", "other"),):
            vals = [out[(s, lv)].get(key, 0.0) for s in solves if (s, lv) in out]
            row[key + "_us"] = round(sum(vals) / len(vals), 1) if vals else 0.0
        row["solver_launches"] = sum(counts[(solves[0], lv)].values()) if solves else 0
        row["total_us"] = round(sum(v for k, v in row.items() if k.endswith("_us")), 1)
        table.append(row)
    return table, len(solves)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--warps-per-level", type=int, default=1)
    ap.add_argument("--levels", type=int, default=40)
    ap.add_argument("--skip-solves", type=int, default=1, help="leading solves to drop (warm-up)")
    ap.add_argument("--out")
    a = ap.parse_args()
    table, n = levels_of(a.trace, a.warps_per_level, a.levels, a.skip_solves)
    print(f"# {a.trace}: {n} solve(s) averaged; level 0 = coarsest")
    print("| level | pair SS us | pair SP us | phi/ksi us | sweep us | other us | total us | solver launches |")
    print("|---|---|---|---|---|---|---|---|")
    for r in table:
        print(f"| {r['level_index']} | {r['pair_ss_us']} | {r['pair_sp_us']} | {r['phi_ksi_us']} | {r['sweep_us']} | "
              f"{round(r['other_us'] + r['pair7_us'], 1)} | {r['total_us']} | {r['solver_launches']} |")
    if a.out:
        json.dump({"trace": a.trace, "solves": n, "levels": table}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
