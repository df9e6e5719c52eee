#!/usr/bin/env python3
"""A MODEL (not a measurement) of the strong-scaling bench: per level, time = launch-latency floor + bandwidth part / N with
the redundant halo planes of the communication-avoiding windows + what the exchanges cost where they are not hidden.
Constants come from this round's one-GPU measurements: 1.97 s per 512^3 solve (end of round 2), ~2 ms of launch latency
per level (40 outer x ~50 us), ~50 us per exchange (pack + grouped send/recv + unpack; an assumption until a multi-GPU
box has been measured).   python tools/scale_model.py [--size 512]"""
import argparse
import math

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--exchange-us", type=float, default=50.0)
a = ap.parse_args()
S, K, OUTER = a.size, 5, 40
levels = [math.ceil(S * 0.95 ** l) for l in range(40)]
floor = 2.0e-3
vox = [d ** 3 for d in levels]
t1_total = 1.97 * (S / 512) ** 3 if S != 512 else 1.97
c = (t1_total - 40 * floor) / sum(vox)
print(f"{S}^3: one GPU {t1_total:.2f} s (input), {c * 1e9:.2f} ns per voxel-level above a {floor * 1e3:.1f} ms floor per level")
for n in (1, 2, 4, 8):
    total = 0.0
    for d, v in zip(levels, vox):
        p = d / n
        if n == 1:
            total += floor + c * v
            continue
        thick = p >= 32
        nex = 1
        if not thick:
            for cand in (4, 3, 2):
                if d * d * (math.ceil(p) + 2 * cand * (K + 1)) <= 1.5e6:
                    nex = cand
                    break
        redundancy = (p + (K + 1) * nex) / p          # average widening of the windows of a group
        exch = OUTER / nex * a.exchange_us * 1e-6 * (0.2 if thick else 1.0)   # thick slabs hide most of it behind the interior
        total += floor + c * v * redundancy / n + exch + 6 * a.exchange_us * 1e-6
    print(f"  {n} GPU(s): {total:.3f} s  speed-up {t1_total / total:.2f}")
