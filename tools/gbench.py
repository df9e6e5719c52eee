#!/usr/bin/env python3
"""Times the Gaussian pre-blur operator (rows + columns fused, slices) on an S^3 volume.   python tools/gbench.py [--size 512]"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--sigma", type=float, default=2.0)
a = ap.parse_args()
pkg = importlib.import_module("cuda-flow3d_amd")
S = a.size
cont = pkg.Containers(S, S, S)
src, dst, tmp = cont.new(np.random.default_rng(1).uniform(0, 255, (S, S, S)).astype(np.float32)), cont.new(fill=0), cont.new(fill=0)
op = pkg.Operation("convolution")
op.initialize(cont)
kw = dict(dev_input=src, dev_output=dst, dev_temp=tmp, data_size=(S, S, S), gaussian_sigma=a.sigma)
op.execute(**kw)
pkg.sync()
t0 = time.perf_counter()
for _ in range(10):
    op.execute(**kw)
pkg.sync()
dt = (time.perf_counter() - t0) / 10
print(f"gaussian sigma {a.sigma} on {S}^3: {dt * 1e3:.3f} ms per volume (three passes), {3 * 8 * S ** 3 / dt / 1e9:.0f} GB/s algorithmic (8 B per voxel and pass)")
