#!/usr/bin/env python3
"""A MODEL (not a measurement) of the strong-scaling bench at 2 / 4 / 8 GPUs.

Per pyramid level: time(N) = kernel time of the level on N slabs / N  +  what the exchanges cost where they are not hidden.
For 8 slabs the first term is MEASURED where a one-GPU trace of the 8-slab decomposition exists for the size
(profiles/rNN_slab8_onegpu_<S>.json, the newest round's: kernel time of the eight slabs run one after the other on one GPU, per level, against the
unsplit solve -- it contains the redundant planes of the communication-avoiding windows, the zone launches and the latency
floor every rank pays); other rank counts scale the measured excess by the halo depth per owned plane, sizes without a trace
fall back to the formula of round 2 (a ~2 ms latency floor per level that does not divide, widened windows).  The exchange
term stays an assumption until a multi-GPU box has been measured: --exchange-us per exchange (pack + grouped send/recv +
unpack), a fifth of it where the slab is thick enough for the overlapped order.  `bench.py --gpus N` reports the measured figure
(`exchange_us.worst_rank_mean_us_blocking`): put it here once a node has produced one.
   python tools/scale_model.py [--size 1024] [--exchange-us 50]"""
import argparse
import json
import math
import os

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--exchange-us", type=float, default=50.0)
a = ap.parse_args()
S, K, OUTER = a.size, 5, 40
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import glob
traces = sorted(glob.glob(os.path.join(root, "profiles", f"r0*_slab8_onegpu_{S}.json")))   # the newest round's trace of this size
trace = traces[-1] if traces else os.path.join(root, "profiles", f"r04_slab8_onegpu_{S}.json")
edges = [math.ceil(S * 0.95 ** l) for l in range(40)][::-1]          # coarsest first, like the trace tables
measured = json.load(open(trace)) if os.path.exists(trace) else None
if measured:
    unsplit = [r["unsplit_ms"] * 1e-3 for r in measured["levels"]]
    slabs8 = [r["slabs_ms"] * 1e-3 for r in measured["levels"]]
    t1 = sum(unsplit)
    print(f"{S}^3: one GPU {t1:.2f} s of kernel time (trace), 8 slabs on one GPU {sum(slabs8):.2f} s = {sum(slabs8) / t1:.3f} x  [{os.path.basename(trace)}]")
else:
    t1 = 1.97 * (S / 512) ** 3
    floor = 2.0e-3
    c = (t1 - 40 * floor) / sum(d ** 3 for d in edges)
    unsplit = [floor + c * d ** 3 for d in edges]
    slabs8 = None
    print(f"{S}^3: no 8-slab trace for this size; formula with one GPU = {t1:.2f} s, {floor * 1e3:.1f} ms floor per level")


def exchange_cost(d, n):
    p = d / n
    thick = p >= 32
    nex = 1
    if not thick:
        for cand in (4, 3, 2):
            if d * d * (math.ceil(p) + 2 * cand * (K + 1)) <= 1.5e6:
                nex = cand
                break
    return OUTER / nex * a.exchange_us * 1e-6 * (0.2 if thick else 1.0) + 6 * a.exchange_us * 1e-6, nex


for n in (1, 2, 4, 8):
    total = 0.0
    for i, d in enumerate(edges):
        if n == 1:
            total += unsplit[i]
            continue
        ex, nex = exchange_cost(d, n)
        if slabs8:
            # excess of the 8-slab run over the unsplit one, per rank: (slabs8 - unsplit) / 8 is paid by each of 8 ranks; with fewer
            # ranks the redundant planes per rank stay the same (they depend on K, not on N) while the owned part grows
            excess_per_rank = max(0.0, slabs8[i] - unsplit[i]) / 8.0
            total += unsplit[i] / n + excess_per_rank + ex
        else:
            p = d / n
            redundancy = (p + (K + 1) * nex) / p
            total += 2.0e-3 + (unsplit[i] - 2.0e-3) * redundancy / n + ex
    print(f"  {n} GPU(s): {total:.3f} s  speed-up {t1 / total:.2f}")
