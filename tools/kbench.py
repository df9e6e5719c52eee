#!/usr/bin/env python3
"""Kernel micro-benchmark: times f3d_phi_ksi / f3d_solve_sweep on one W x H x D level with HIP events
(f3d_prof_*), on random data.  Used for tuning and for the rocprofv3 / PMC runs whose summaries live in profiles/.
   python tools/kbench.py [--size 512 | --dims W H D] [--reps 20] [--kernel sweep|sweep2|sweeppk|sweep2fd|sweeppkfd|sweep3|sweep2pk|tri|phi|both|bothfd|all]
                          [--ablate N]
--ablate N: timing-only builds of the solver kernels that skip parts of the work (WRONG results; N as described at k_pair8 / k_sweep7 /
k_sweep6).  They exist only in the LAB library (make -C cuda-flow3d_amd lab -> lib/lab/), which this option loads instead of the
product; the shipped library has no such switch.
"""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dims", type=int, nargs=3)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--kernel", default="both")
    ap.add_argument("--ablate", type=int, default=0)
    a = ap.parse_args()
    if a.ablate:
        lab = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-flow3d_amd", "lib", "lab")
        if not os.path.exists(os.path.join(lab, "libf3d_hip.so")):
            sys.exit("kbench --ablate: build the lab library first (make -C cuda-flow3d_amd lab)")
        os.environ["F3D_LIBDIR"] = lab                     # read by the binding when it is imported, below
        for name in ("F3D_ABLATE", "F3D_ABLATE7", "F3D_ABLATE8"):
            os.environ.setdefault(name, str(a.ablate))
    W, H, D = a.dims if a.dims else (a.size,) * 3
    pkg = importlib.import_module("cuda-flow3d_amd")
    hip = pkg.hip()
    cont = pkg.Containers(W, H, D)
    rng = np.random.default_rng(1)
    plane = lambda lo, hi: rng.uniform(lo, hi, size=(1, H, W)).astype(np.float32)
    ptr = []
    for lo, hi in [(0, 255), (0, 255), (-3, 3), (-3, 3), (-3, 3), (-.5, .5), (-.5, .5), (-.5, .5)]:
        p = cont.alloc()
        base = plane(lo, hi)
        vol = np.repeat(base, D, axis=0)
        vol += rng.uniform(-0.01, 0.01, size=(D, 1, 1)).astype(np.float32)
        cont.upload(p, vol)
        ptr.append(p)
    cont.set_current()
    phi, ksi = cont.alloc(fill=0), cont.alloc(fill=0)
    out = [cont.alloc(fill=0) for _ in range(3)]
    phi2, ksi2 = cont.alloc(fill=0), cont.alloc(fill=0)
    h = (1.0, 1.0, 1.0)
    pkg.check(hip.f3d_phi_ksi(*ptr, W, H, D, *h, 0.001, 0.001, phi, ksi, None))
    pkg.check(hip.f3d_solve_sweep(*ptr, phi, ksi, W, H, D, *h, 7.5, *out, None))
    fd = [cont.alloc(fill=0) for _ in range(4)]
    pkg.check(hip.f3d_frame_derivatives(ptr[0], ptr[1], W, H, D, *h, *fd, None))
    pkg.sync()
    hip.f3d_prof_reset()
    hip.f3d_prof_enable(1)
    for _ in range(a.reps):
        if a.kernel in ("phi", "both", "all"):
            pkg.check(hip.f3d_phi_ksi(*ptr, W, H, D, *h, 0.001, 0.001, phi, ksi, None))
        if a.kernel in ("sweep", "both", "all"):
            pkg.check(hip.f3d_solve_sweep(*ptr, phi, ksi, W, H, D, *h, 7.5, *out, None))
        if a.kernel in ("sweep2", "both", "all"):
            pkg.check(hip.f3d_solve_sweep2(*ptr, phi, ksi, W, H, D, *h, 7.5, *out, None))
        if a.kernel in ("sweep2fd", "bothfd", "all"):
            pkg.check(hip.f3d_solve_sweep2_fd(*fd, *ptr[2:], phi, ksi, W, H, D, *h, 7.5, *out, None))
        if a.kernel in ("sweeppkfd", "bothfd", "all"):
            pkg.check(hip.f3d_solve_sweep_phi_ksi_fd(*fd, *ptr[2:], phi, ksi, W, H, D, *h, 7.5, 0.001, 0.001, *out, phi2, ksi2, None))
        if a.kernel in ("sweep3", "tri"):
            pkg.check(hip.f3d_solve_sweep3(*ptr, phi, ksi, W, H, D, *h, 7.5, *out, None))
        if a.kernel in ("sweep2pk", "tri"):
            pkg.check(hip.f3d_solve_sweep2_phi_ksi(*ptr, phi, ksi, W, H, D, *h, 7.5, 0.001, 0.001, *out, phi2, ksi2, None))
        if a.kernel in ("sweeppk", "both", "all"):
            pkg.check(hip.f3d_solve_sweep_phi_ksi(*ptr, phi, ksi, W, H, D, *h, 7.5, 0.001, 0.001, *out, phi2, ksi2, None))
    pkg.sync()
    # sweep2: two sweeps of algorithmic work (2 x 52 B per voxel) per launch
    # sweeppk: one sweep + the next phi/ksi (52 + 40 B per voxel) per launch
    # sweep3: three sweeps (3 x 52 B); sweep2pk: two sweeps + the next phi/ksi (2 x 52 + 40 B)
    for kid, name, bpv in ((0, "phi_ksi", 40.0), (1, "sweep", 52.0), (2, "sweep2", 104.0), (3, "sweeppk", 92.0), (4, "sweep3", 156.0),
                           (5, "sweep2pk", 144.0)):
        ms, n, vox = C.c_double(), C.c_uint64(), C.c_double()
        hip.f3d_prof_read(kid, 0, C.byref(ms), C.byref(n), C.byref(vox))
        if n.value:
            us = ms.value / n.value * 1e3
            gbs = bpv * vox.value / (ms.value * 1e-3) / 1e9
            print(f"{name:8s} {W}x{H}x{D}: {us:9.1f} us/launch  {gbs:8.1f} GB/s algorithmic  ({gbs / 80:.1f} % of 8 TB/s)")
    cont.free()


if __name__ == "__main__":
    main()
