#!/usr/bin/env python3
"""512^3: OpticalFlowE::ComputeFlow including the host <-> device copies (the reference's timed region) beside the
device-resident solve, and what page-locking the frames costs and buys.   python tools/host_inclusive.py"""
import importlib, sys, time, ctypes as C
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("cuda-flow3d_amd")
n = 512
f0, f1 = pkg.synth_pair(n, n, n)
flow = pkg.OpticalFlow(); flow.initialize(n, n, n)
flow.upload(f0, f1); r = flow.compute_resident(silent=True); r = flow.compute_resident(silent=True)
print(f"resident: {r:.3f} s")
for rep in range(2):
    t0 = time.time(); u, v, w = flow.compute(f0, f1, silent=True); t1 = time.time()
    print(f"host-inclusive ComputeFlow (pageable, incl. numpy alloc of outputs): {t1 - t0:.3f} s")
hip = pkg.hip()
for a in (f0, f1):
    t0 = time.time(); pkg.check(hip.f3d_host_register(a.ctypes.data_as(C.c_void_p), a.nbytes)); print(f"register {a.nbytes/1e6:.0f} MB: {time.time()-t0:.3f} s")
t0 = time.time(); flow.upload(f0, f1); pkg.sync(); print(f"upload pinned 2 frames: {time.time()-t0:.3f} s")
f2 = f0.copy()
t0 = time.time(); flow.upload(f2, f2); pkg.sync(); print(f"upload pageable 2 frames: {time.time()-t0:.3f} s")
t0 = time.time(); out = flow.download(); print(f"download 3 flows pageable (incl. alloc): {time.time()-t0:.3f} s")
flow.destroy()
