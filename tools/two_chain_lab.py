#!/usr/bin/env python3
"""Do two fused-sweep launches on two lanes fill each other's last rounds?  One level, random data:
  (a) R two-sweep launches over the whole level on one lane;
  (b) the same level as two z windows, R launches each, one window per lane (no dependencies between them: an upper bound
      for any schedule that lets the head of one launch run under the tail of another);
  (c) both windows one after the other on ONE lane (what cutting the launch costs by itself).
   python3 tools/two_chain_lab.py --size 439 [--reps 20] [--cut 0.5]"""
import argparse, ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=439)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cut", type=float, default=0.5)
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
    out = [cont.alloc(fill=0) for _ in range(3)]
    out2 = [cont.alloc(fill=0) for _ in range(3)]
    h = (1.0, 1.0, 1.0)
    pkg.check(hip.f3d_phi_ksi(*ptr, W, H, D, *h, 0.001, 0.001, phi, ksi, None))
    fd = [cont.alloc(fill=0) for _ in range(4)]
    pkg.check(hip.f3d_frame_derivatives(ptr[0], ptr[1], W, H, D, *h, *fd, None))
    pkg.sync()
    mid = int(D * a.cut)
    lower, upper = pkg.Slab(0, 0, mid), pkg.Slab(0, mid, D)

    def sweep2(o, slab):
        pkg.check(hip.f3d_solve_sweep2_fd(*fd, *ptr[2:], phi, ksi, W, H, D, *h, 7.5, *o, C.byref(slab) if slab else None))

    def timed(fn):
        fn(2)
        pkg.sync()
        t = time.perf_counter()
        fn(a.reps)
        pkg.sync()
        return (time.perf_counter() - t) / a.reps * 1e6

    whole = timed(lambda n: [sweep2(out, None) for _ in range(n)])
    one_lane = timed(lambda n: [(sweep2(out, lower), sweep2(out, upper)) for _ in range(n)])
    lane = pkg.Lane()

    def both(n):
        for _ in range(n):
            sweep2(out, lower)
            lane.make_current()
            cont.set_current()
            sweep2(out2, upper)
            lane.release()
        lane.make_current()
        pkg.sync()          # the lane's stream
        lane.release()

    two_lanes = timed(both)
    lane.destroy()
    print(f"{a.size}^3 two sweeps: whole level {whole:8.1f} us   two windows on one lane {one_lane:8.1f} us   on two lanes {two_lanes:8.1f} us"
          f"   ({two_lanes / whole:.3f} of the whole-level launch)")
    cont.free()


if __name__ == "__main__":
    main()
