#!/usr/bin/env python3
"""Whole default pyramids on small cubes (64^3 ... 256^3), device-resident: the latency floor of the launch sequence
("Small levels are latency-bound").   python tools/latency_floor.py"""
import importlib, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cuda-flow3d_amd")
for n in (64, 96, 128, 192, 256):
    f0, f1 = pkg.synth_pair(n, n, n)
    flow = pkg.OpticalFlow(); flow.initialize(n, n, n); flow.upload(f0, f1)
    flow.compute_resident(silent=True)
    secs = min(flow.compute_resident(silent=True) for _ in range(3))
    lv = pkg.max_warp_level(n, n, n, 0.95)
    print(f"{n}^3: {secs*1e3:8.1f} ms  levels {min(lv,40)}  {n**3/secs/1e6:6.1f} Mvox/s", flush=True)
    flow.destroy()
