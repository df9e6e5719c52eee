#!/usr/bin/env python3
"""One fused launch over a whole level against the same work cut into n z windows launched one after the other (same lane, same
buffers: the windows are independent, every launch ends in a device-wide join).  Two sweeps and sweep + phi/ksi on frame derivatives.
   python3 tools/zsplit_lab.py --size 397 [--reps 20] [--splits 1 2 3 4 6]"""
import argparse, ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=397)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--splits", type=int, nargs="+", default=[1, 2, 3, 4, 6])
    a = ap.parse_args()
    W = H = D = a.size
    pkg = importlib.import_module("cuda-flow3d_amd")
    hip = pkg.hip()
    cont = pkg.Containers(W, H, D)
    rng = np.random.default_rng(1)
    ptr = []
    for lo, hi in [(0, 255), (0, 255), (-3, 3), (-3, 3), (-3, 3), (-.5, .5), (-.5, .5), (-.5, .5)]:
        p = cont.alloc()
        vol = np.repeat(rng.uniform(lo, hi, size=(1, H, W)).astype(np.float32), D, axis=0)
        vol += rng.uniform(-0.01, 0.01, size=(D, 1, 1)).astype(np.float32)
        cont.upload(p, vol)
        ptr.append(p)
    cont.set_current()
    phi, ksi = cont.alloc(fill=0), cont.alloc(fill=0)
    phi2, ksi2 = cont.alloc(fill=0), cont.alloc(fill=0)
    out = [cont.alloc(fill=0) for _ in range(3)]
    h = (1.0, 1.0, 1.0)
    pkg.check(hip.f3d_phi_ksi(*ptr, W, H, D, *h, 0.001, 0.001, phi, ksi, None))
    fd = [cont.alloc(fill=0) for _ in range(4)]
    pkg.check(hip.f3d_frame_derivatives(ptr[0], ptr[1], W, H, D, *h, *fd, None))
    pkg.sync()

    def windows(n):
        cuts = [D * i // n for i in range(n + 1)]
        return [pkg.Slab(0, cuts[i], cuts[i + 1]) for i in range(n)]

    def ss(slab):
        pkg.check(hip.f3d_solve_sweep2_fd(*fd, *ptr[2:], phi, ksi, W, H, D, *h, 7.5, *out, C.byref(slab)))

    def sp(slab):
        pkg.check(hip.f3d_solve_sweep_phi_ksi_fd(*fd, *ptr[2:], phi, ksi, W, H, D, *h, 7.5, 0.001, 0.001, *out, phi2, ksi2, C.byref(slab)))

    def timed(fn, wins):
        for w in wins:
            fn(w)
        pkg.sync()
        t = time.perf_counter()
        for _ in range(a.reps):
            for w in wins:
                fn(w)
        pkg.sync()
        return (time.perf_counter() - t) / a.reps * 1e6

    for name, fn in (("two sweeps", ss), ("sweep+phi/ksi", sp)):
        res = [(n, timed(fn, windows(n))) for n in a.splits]
        base = res[0][1]
        print(f"{a.size}^3 {name:14s} " + "  ".join(f"n={n}: {us:7.1f} us ({us / base:.3f})" for n, us in res))
    cont.free()


if __name__ == "__main__":
    main()
