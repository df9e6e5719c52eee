#!/usr/bin/env python3
"""Back-to-back dependent launches of a trivial kernel (f3d_add on an 8^3 box) on the library stream: the time per launch is
what the runtime + hardware need between two dependent kernels, the floor under every tiny pyramid level.
   python tools/launch_gap.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("cuda-flow3d_amd"); hip = pkg.hip()
box = pkg.Containers(64, 8, 8)
a = box.new(np.zeros((8, 8, 64), np.float32)); b = box.new(np.ones((8, 8, 64), np.float32))
box.set_current()
for n in (200, 2000, 20000):
    pkg.sync(); t0 = time.perf_counter()
    for _ in range(n):
        hip.f3d_add(a, b, 64, 8, 8, None)
    t1 = time.perf_counter(); pkg.sync(); t2 = time.perf_counter()
    print(f"{n} launches: host enqueue {1e6 * (t1 - t0) / n:.2f} us each, end to end {1e6 * (t2 - t0) / n:.2f} us each")
box.free()
