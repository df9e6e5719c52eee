#!/usr/bin/env python3
"""Like the parity test, over and over: fresh containers, uploads, one f3d_solve_sweep2, compare with two single sweeps."""
import os, sys, importlib, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
f3d = importlib.import_module("cuda-flow3d_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
idle_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0
cases = [((131, 7, 13), (192, 8, 16), (7.1, 1.6, 1.25)), ((64, 64, 130), (64, 64, 130), (1.0, 1.0, 1.0)),
         ((37, 21, 9), (64, 32, 16), (1.0, 1.0, 1.0)), ((70, 70, 70), (128, 72, 70), (7.1, 1.6, 1.25))]
hip = f3d.hip()
total_bad = 0
for it in range(reps):
    for dims, cdims, h in cases:
        W, H, D = dims
        rng = np.random.default_rng(hash((dims, it)) % 2**32)
        def mk(lo, hi):
            a = np.full(cdims[::-1], np.nan, np.float32)
            a[:D, :H, :W] = rng.uniform(lo, hi, (D, H, W)).astype(np.float32)
            return a
        arrs = [mk(0, 255), mk(0, 255), mk(-3, 3), mk(-3, 3), mk(-3, 3), mk(-.5, .5), mk(-.5, .5), mk(-.5, .5)]
        box = f3d.Containers(*cdims)
        box.alloc(fill=0xFF)
        box.set_current()
        ptr = [box.new(a) for a in arrs]
        phi, ksi = box.new(), box.new()
        f3d.check(hip.f3d_phi_ksi(*ptr, W, H, D, *h, 0.001, 0.001, phi, ksi, None))
        outs = [box.new() for _ in range(3)]
        f3d.sync()
        if idle_ms:
            time.sleep(idle_ms * 1e-3)   # let the GPU fall idle, as it does while the oracle runs in the parity tests
        f3d.check(hip.f3d_solve_sweep2(*ptr, phi, ksi, W, H, D, *h, 7.5, *outs, None))
        got = [box.download(p, cdims)[:D, :H, :W].copy() for p in outs]
        t1 = [box.new() for _ in range(3)]
        t2 = [box.new() for _ in range(3)]
        f3d.check(hip.f3d_solve_sweep(*ptr, phi, ksi, W, H, D, *h, 7.5, *t1, None))
        f3d.check(hip.f3d_solve_sweep(*ptr[:5], *t1, phi, ksi, W, H, D, *h, 7.5, *t2, None))
        f3d.sync()
        for name, g, p in zip("uvw", got, t2):
            e = box.download(p, cdims)[:D, :H, :W]
            bad = np.argwhere(g.view(np.uint32) != e.view(np.uint32))
            if len(bad):
                total_bad += 1
                print(f"  it {it} {dims} d{name}: {len(bad)} differ; z {sorted(set(bad[:,0]))[:12]} y {sorted(set(bad[:,1]))[:12]} "
                      f"x {sorted(set(bad[:,2]))[:16]}", flush=True)
                break
        f3d.sync()
        box.free()
print(f"{total_bad} bad launches of {reps * len(cases)}", flush=True)
