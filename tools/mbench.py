#!/usr/bin/env python3
"""Times f3d_median (and the small streaming kernels) on one level with HIP events."""
import argparse, ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=256); ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--radius", type=int, default=5)
a = ap.parse_args()
pkg = importlib.import_module("cuda-flow3d_amd"); hip = pkg.hip()
S = a.size
cont = pkg.Containers(S, S, S)
rng = np.random.default_rng(0)
vol = rng.normal(size=(S, S, S)).astype(np.float32)
pin = cont.alloc(); cont.upload(pin, vol); pout = cont.alloc(fill=0)
cont.set_current()
e0, e1 = C.c_void_p(), C.c_void_p()
hip.f3d_event_create(C.byref(e0)); hip.f3d_event_create(C.byref(e1))
pkg.check(hip.f3d_median(pin, S, S, S, a.radius, pout, None)); pkg.sync()
hip.f3d_event_record(e0)
for _ in range(a.reps): pkg.check(hip.f3d_median(pin, S, S, S, a.radius, pout, None))
hip.f3d_event_record(e1); hip.f3d_event_sync(e1)
ms = C.c_float(); hip.f3d_event_elapsed_ms(C.byref(ms), e0, e1)
print(f"median r={a.radius} {S}^3: {ms.value / a.reps * 1e3:.1f} us/launch, {S**3 / (ms.value / a.reps * 1e-3) / 1e9:.2f} Gvoxel/s")
