#!/usr/bin/env python3
"""Cold-start check of f3d_solve_sweep2: in a fresh process the fused launch is the FIRST solver kernel to run; it is then
compared with two single sweeps.  Exit code 1 on a mismatch.   tools/cold_fused.py [case index]"""
import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
f3d = importlib.import_module("cuda-flow3d_amd")
cases = [((37, 21, 9), (64, 32, 16), (1.0, 1.0, 1.0)), ((131, 7, 13), (192, 8, 16), (7.1, 1.6, 1.25)),
         ((64, 64, 130), (64, 64, 130), (1.0, 1.0, 1.0))]
dims, cdims, h = cases[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
hip = f3d.hip()
W, H, D = dims
rng = np.random.default_rng(77)
def mk(lo, hi):
    a = np.full(cdims[::-1], np.nan, np.float32)
    a[:D, :H, :W] = rng.uniform(lo, hi, (D, H, W)).astype(np.float32)
    return a
arrs = [mk(0, 255), mk(0, 255), mk(-3, 3), mk(-3, 3), mk(-3, 3), mk(-.5, .5), mk(-.5, .5), mk(-.5, .5), mk(0.01, 2), mk(0.01, 2)]
box = f3d.Containers(*cdims)
box.alloc(fill=0xFF)
box.set_current()
ptr = [box.new(a) for a in arrs]
outs = [box.new() for _ in range(3)]
f3d.check(hip.f3d_solve_sweep2(*ptr, W, H, D, *h, 7.5, *outs, None))
got = [box.download(p, cdims)[:D, :H, :W].copy() for p in outs]
t1 = [box.new() for _ in range(3)]
t2 = [box.new() for _ in range(3)]
f3d.check(hip.f3d_solve_sweep(*ptr, W, H, D, *h, 7.5, *t1, None))
f3d.check(hip.f3d_solve_sweep(*ptr[:5], *t1, *ptr[8:], W, H, D, *h, 7.5, *t2, None))
f3d.sync()
rc = 0
for name, g, p in zip("uvw", got, t2):
    e = box.download(p, cdims)[:D, :H, :W]
    bad = np.argwhere(g.view(np.uint32) != e.view(np.uint32))
    if len(bad):
        rc = 1
        zs = {int(z): int((bad[:, 0] == z).sum()) for z in sorted(set(bad[:, 0]))}
        print(f"MISMATCH {dims} d{name}: {len(bad)} voxels; per plane {zs}; y range {bad[:,1].min()}..{bad[:,1].max()} x range {bad[:,2].min()}..{bad[:,2].max()}", flush=True)
box.free()
sys.exit(rc)
