#!/usr/bin/env python3
"""One default solve of an S^3 synthetic pair (after a warm-up), for `rocprofv3 --kernel-trace --stats -- python3 tools/trace_size.py --size 128`."""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
pkg = importlib.import_module("cuda-flow3d_amd")
n = a.size
f0, f1 = pkg.synth_pair(n, n, n)
flow = pkg.OpticalFlow(); flow.initialize(n, n, n); flow.upload(f0, f1)
secs = [flow.compute_resident(silent=True) for _ in range(a.reps)]
print(f"{n}^3: {min(secs) * 1e3:.1f} ms per solve ({a.reps} solves)")
flow.destroy()
