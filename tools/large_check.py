#!/usr/bin/env python3
"""Sanity of the resident driver beyond the largest configuration with an oracle pin (1024^3): a full default solve of the
synthetic translated pair at --size (default 1280) must be finite and recover the translation (+2, -1, +0.5) in the
textured interior.   python tools/large_check.py [--size 1280]"""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1280)
a = ap.parse_args()
pkg = importlib.import_module("cuda-flow3d_amd")
n = a.size
f0, f1 = pkg.synth_pair(n, n, n)
flow = pkg.OpticalFlow()
flow.initialize(n, n, n)
flow.upload(f0, f1)
secs = flow.compute_resident(silent=True)
u, v, w = flow.download()
flow.destroy()
inner = (slice(n // 4, -n // 4),) * 3
means = [float(x[inner].mean()) for x in (u, v, w)]
finite = all(bool(np.isfinite(x).all()) for x in (u, v, w))
print(f"{n}^3 default solve: {secs:.2f} s, {n ** 3 / secs / 1e6:.1f} Mvoxels/s, finite={finite}, interior means {means}")
ok = finite and abs(means[0] - 2.0) < 0.1 and abs(means[1] + 1.0) < 0.1 and abs(means[2] - 0.5) < 0.1
sys.exit(0 if ok else 1)
