#!/usr/bin/env python3
"""Soak of the hand-scheduled fused launches: for a time budget, random boxes (widths around the 64-lane tile edges, tile heights
4 / 8 / 12 / the y march, random container slack and spacings) go through every fused launch -- two sweeps and sweep + next phi/ksi,
on frames and on frame derivatives -- and through the batched median / add, and every result is compared bit for bit with the
SEPARATE launches of the same library (k_sweep6 / k_phiksi6 / the single-volume entries: other kernels, other data movement).  A
counted wait off by one or a slot read too early shows up as a handful of wrong voxels once in hundreds of launches; this is the
tool that looks for that.   python tools/soak_fused.py [seconds] [seed]"""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
f3d = importlib.import_module("cuda-flow3d_amd")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
hip = f3d.hip()
t0 = time.time(); last = t0; launches = bad = it = 0
same = lambda a, b: bool((a.view(np.uint32) == b.view(np.uint32)).all())
while time.time() - t0 < budget:
    it += 1
    kind = rng.integers(0, 4)
    if kind == 0:   W, H, D = int(rng.choice([63, 64, 65, 127, 128, 129, 130, 191, 193])), int(rng.integers(4, 40)), int(rng.integers(4, 24))
    elif kind == 1: W, H, D = int(rng.integers(4, 200)), int(rng.integers(4, 400)), int(rng.integers(4, 9))     # thin: the y march (4 and 5 planes: the tile without halo rows)
    elif kind == 2: W, H, D = int(rng.integers(8, 90)), int(rng.integers(8, 90)), int(rng.integers(8, 90))      # small cubes: 4-row tiles
    else:           W, H, D = int(rng.integers(100, 330)), int(rng.integers(60, 200)), int(rng.integers(20, 70))  # several rounds of 12-row tiles
    cdims = (W + int(rng.integers(0, 9)), H + int(rng.integers(0, 5)), D + int(rng.integers(0, 3)))
    h = tuple(float(x) for x in rng.choice([1.0, 1.25, 1.6, 512 / 487, 7.1], 3))
    def mk(lo, hi):
        a = np.full(cdims[::-1], np.nan, np.float32)
        a[:D, :H, :W] = rng.uniform(lo, hi, (D, H, W)).astype(np.float32)
        return a
    arrs = [mk(0, 255), mk(0, 255), mk(-3, 3), mk(-3, 3), mk(-3, 3), mk(-.5, .5), mk(-.5, .5), mk(-.5, .5)]
    box = f3d.Containers(*cdims); box.alloc(fill=0xFF); box.set_current()
    ptr = [box.new(a) for a in arrs]
    phi, ksi = box.new(), box.new()
    f3d.check(hip.f3d_phi_ksi(*ptr, W, H, D, *h, 0.001, 0.002, phi, ksi, None))
    get = lambda p: box.download(p, cdims)[:D, :H, :W].copy()
    # the separate launches
    s1 = [box.new() for _ in range(3)]; s2 = [box.new() for _ in range(3)]; pn, kn = box.new(), box.new()
    f3d.check(hip.f3d_solve_sweep(*ptr, phi, ksi, W, H, D, *h, 7.5, *s1, None))
    f3d.check(hip.f3d_solve_sweep(*ptr[:5], *s1, phi, ksi, W, H, D, *h, 7.5, *s2, None))
    f3d.check(hip.f3d_phi_ksi(*ptr[:5], *s1, W, H, D, *h, 0.001, 0.002, pn, kn, None))
    f3d.sync()
    exp2 = [get(p) for p in s2]; exp1 = [get(p) for p in s1] + [get(pn), get(kn)]
    # three sweeps / two sweeps + the next phi/ksi by the separate launches (round 4: k_tri)
    s3 = [box.new() for _ in range(3)]; pn2, kn2 = box.new(), box.new()
    f3d.check(hip.f3d_solve_sweep(*ptr[:5], *s2, phi, ksi, W, H, D, *h, 7.5, *s3, None))
    f3d.check(hip.f3d_phi_ksi(*ptr[:5], *s2, W, H, D, *h, 0.001, 0.002, pn2, kn2, None))
    f3d.sync()
    exp3 = [get(p) for p in s3]; exp2p = exp2 + [get(pn2), get(kn2)]
    fd = [box.new() for _ in range(4)]
    f3d.check(hip.f3d_frame_derivatives(ptr[0], ptr[1], W, H, D, *h, *fd, None))
    reps = 6 if W * H * D < 2e5 else 2
    for rep in range(reps):
        o2 = [box.new() for _ in range(3)]; o1 = [box.new() for _ in range(5)]
        q2 = [box.new() for _ in range(3)]; q1 = [box.new() for _ in range(5)]
        f3d.check(hip.f3d_solve_sweep2(*ptr, phi, ksi, W, H, D, *h, 7.5, *o2, None))
        f3d.check(hip.f3d_solve_sweep_phi_ksi(*ptr, phi, ksi, W, H, D, *h, 7.5, 0.001, 0.002, *o1, None))
        f3d.check(hip.f3d_solve_sweep2_fd(*fd, *ptr[2:], phi, ksi, W, H, D, *h, 7.5, *q2, None))
        f3d.check(hip.f3d_solve_sweep_phi_ksi_fd(*fd, *ptr[2:], phi, ksi, W, H, D, *h, 7.5, 0.001, 0.002, *q1, None))
        t3 = [box.new() for _ in range(3)]; t2 = [box.new() for _ in range(5)]
        f3d.check(hip.f3d_solve_sweep3(*ptr, phi, ksi, W, H, D, *h, 7.5, *t3, None))
        f3d.check(hip.f3d_solve_sweep2_phi_ksi(*ptr, phi, ksi, W, H, D, *h, 7.5, 0.001, 0.002, *t2, None))
        f3d.sync(); launches += 6
        for name, got, exp in (("sweep2", o2, exp2), ("sweep+phi/ksi", o1, exp1), ("sweep2_fd", q2, exp2), ("sweep+phi/ksi_fd", q1, exp1),
                               ("sweep3", t3, exp3), ("sweep2+phi/ksi", t2, exp2p)):
            for i, (g, e) in enumerate(zip(got, exp)):
                gg = get(g)
                if not same(gg, e):
                    bad += 1
                    w = np.argwhere(gg.view(np.uint32) != e.view(np.uint32))
                    print(f"  MISMATCH it {it} rep {rep} {name}[{i}] {W}x{H}x{D} in {cdims} h {h}: {len(w)} voxels, first {w[:4].tolist()}", flush=True)
    # batched median / add against the single-volume entries
    if min(W, H, D) > 2:
        m1 = [box.new() for _ in range(3)]; m3 = [box.new() for _ in range(3)]
        arr = lambda ps: (C.c_uint64 * len(ps))(*ps)
        for i in range(3):
            f3d.check(hip.f3d_median(s2[i], W, H, D, 5, m1[i], None))
        f3d.check(hip.f3d_median_n(arr(s2), 3, W, H, D, 5, arr(m3), None))
        f3d.sync(); launches += 1
        for i in range(3):
            if not same(get(m1[i]), get(m3[i])):
                bad += 1; print(f"  MISMATCH it {it} median_n[{i}] {W}x{H}x{D}", flush=True)
    f3d.sync(); box.free()
    if time.time() - last > 30:
        last = time.time()
        print(f"[{last - t0:6.0f} s] {it} boxes, {launches} fused / batched launches checked, {bad} mismatches", flush=True)
print(f"soak: {it} boxes, {launches} fused / batched launches, {bad} mismatches in {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
