#!/usr/bin/env python3
"""A few default solves of one configuration (after a warm-up), for
`rocprofv3 --kernel-trace --stats -- python3 tools/trace_size.py --size 128` (S^3 synthetic pair) or `--config c2|c3`
(BASELINE configs 2 and 3: the shipped 128^3 pair / the 584x388x5 thin slab from tests/golden)."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--config", choices=("c2", "c3"))
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
pkg = importlib.import_module("cuda-flow3d_amd")
if a.config:
    import bench
    f0, f1 = bench.golden_pairs()[a.config]
    name = a.config
else:
    n = a.size
    f0, f1 = pkg.synth_pair(n, n, n)
    name = f"{n}^3"
d, h, w = f0.shape
flow = pkg.OpticalFlow(); flow.initialize(w, h, d); flow.upload(f0, f1)
flow.compute_resident(silent=True)
secs = [flow.compute_resident(silent=True) for _ in range(a.reps)]
print(f"{name} ({w}x{h}x{d}): {min(secs) * 1e3:.1f} ms per solve ({a.reps} solves after one warm-up)")
flow.destroy()
