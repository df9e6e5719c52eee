#!/usr/bin/env python3
"""Probe for the SIGSEGV seen under `rocprofv3 --pmc X -- python3 bench.py` (counter collection WITHOUT --kernel-trace): 3-D copies
between numpy volumes and a pitched container through the C ABI, optionally after kernels of this library and optionally with the
host source page-locked -- no timing events anywhere.
   rocprofv3 --pmc FETCH_SIZE -- python3 tools/pmc_copy_probe.py [--size 256] [--kernels N] [--pin]"""
import argparse, ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--kernels", type=int, default=0)
ap.add_argument("--pin", action="store_true")
a = ap.parse_args()
pkg = importlib.import_module("cuda-flow3d_amd")
hip = pkg.hip()
S = a.size
cont = pkg.Containers(S, S, S)
p, q = cont.alloc(), cont.alloc()
cont.set_current()
vol = np.arange(S * S * S, dtype=np.float32).reshape(S, S, S) * np.float32(1e-3)
if a.pin:
    pkg.check(hip.f3d_host_register(C.c_void_p(vol.ctypes.data), vol.nbytes))
for i in range(a.reps):
    cont.upload(p, vol)
    cont.upload(q, vol)
    for _ in range(a.kernels):
        pkg.check(hip.f3d_add(p, q, S, S, S, None))
    back = cont.download(p, (S, S, S))
    print("round trip", i, float(back[1, 2, 3]), flush=True)
if a.pin:
    hip.f3d_host_unregister(C.c_void_p(vol.ctypes.data))
cont.free()
pkg.shutdown()
print("probe done", flush=True)
