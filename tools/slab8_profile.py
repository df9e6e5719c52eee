#!/usr/bin/env python3
"""The work an 8-GPU run divides by eight, on ONE GPU: the default solve of an S^3 pair on 8 z-slabs run one after the other
in one process (plane copies instead of RCCL), and the unsplit solve beside it.  Under
`rocprofv3 --kernel-trace --stats -- python3 tools/slab8_profile.py --size 1024 --only slabs|unsplit` it gives the kernel
time of either side per kernel (profiles/rNN_slab8_onegpu*.md)."""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--only", choices=("slabs", "unsplit", "both"), default="both")
ap.add_argument("--reps", type=int, default=1)
a = ap.parse_args()
pkg = importlib.import_module("cuda-flow3d_amd")
n, r = a.size, a.ranks
f0, f1 = pkg.synth_pair(n, n, n)
if a.only in ("slabs", "both"):
    flow = pkg.SlabOpticalFlow(r, list(range(r)), halo_capacity=32)
    flow.initialize(n, n, n)
    flow.upload(f0, f1)
    t = [flow.compute_resident() for _ in range(a.reps + 1)]
    print(f"{n}^3 on {r} slabs in one process: {min(t[1:]):.3f} s per solve (first {t[0]:.3f})", flush=True)
    flow.destroy()
if a.only in ("unsplit", "both"):
    flow = pkg.OpticalFlow()
    flow.initialize(n, n, n)
    flow.upload(f0, f1)
    t = [flow.compute_resident(silent=True) for _ in range(a.reps + 1)]
    print(f"{n}^3 unsplit: {min(t[1:]):.3f} s per solve (first {t[0]:.3f})", flush=True)
    flow.destroy()
