#!/usr/bin/env python3
"""Times the out-of-core driver (OpticalFlowP) on a synthetic pair and, beside it, the resident driver on the same
schedule (no pre-blur, no median, which is what the piecemeal driver computes).
   python tools/pbench.py --size 512 [--budget-mb 4096] [--outer 40] [--levels 40] [--no-resident] [--check]
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dims", type=int, nargs=3, help="W H D instead of a cube")
    ap.add_argument("--budget-mb", type=float, default=0)
    ap.add_argument("--outer", type=int, default=40)
    ap.add_argument("--levels", type=int, default=40)
    ap.add_argument("--per-pass", type=int, default=0)
    ap.add_argument("--no-resident", action="store_true")
    ap.add_argument("--all-through-host", action="store_true", help="no resident coarse levels in the piecemeal driver")
    ap.add_argument("--check", action="store_true", help="compare the two results bit for bit")
    ap.add_argument("--verbose", action="store_true", help="the driver's own log (levels, chunk plans)")
    ap.add_argument("--full", action="store_true", help="full pipeline on both sides: pre-blur and median included")
    a = ap.parse_args()
    pkg = importlib.import_module("cuda-flow3d_amd")
    n = a.size
    W, H, D = a.dims if a.dims else (n, n, n)
    t0 = time.time()
    f0, f1 = pkg.synth_pair(W, H, D)
    print(f"synthetic {W}x{H}x{D} pair in {time.time() - t0:.1f} s", flush=True)
    kw = dict(outer_iterations_count=a.outer, warp_levels_count=a.levels)
    exp = None
    if not a.no_resident:
        flow = pkg.OpticalFlow()
        flow.initialize(W, H, D)
        flow.upload(f0, f1)
        secs = flow.compute_resident(silent=True, **(kw if a.full else dict(kw, gaussian_sigma=0.0, median_radius=1)))
        print(f"resident : {secs:8.3f} s on the device  {W * H * D / secs / 1e6:7.2f} Mvoxels/s", flush=True)
        if a.check:
            exp = flow.download()
        flow.destroy()
    if a.budget_mb > 0:
        os.environ["F3D_P_BUDGET_MB"] = repr(a.budget_mb)
    if a.per_pass > 0:
        os.environ["F3D_P_OUTER_PER_PASS"] = str(a.per_pass)
    flow = pkg.PiecemealOpticalFlow()
    flow.initialize(W, H, D)
    flow.set_resident(not a.all_through_host)
    flow.set_full_pipeline(a.full)
    t0 = time.time()
    got = flow.compute(f0, f1, silent=not a.verbose, **kw)
    wall = time.time() - t0
    passes, streamed, on_device = flow.stats()
    print(f"piecemeal: {flow.device_seconds:8.3f} s ({wall:.3f} s wall with host allocation and page-locking)  "
          f"{W * H * D / flow.device_seconds / 1e6:7.2f} Mvoxels/s  budget {a.budget_mb or 'auto'} MB  "
          f"{passes} solver residencies, {streamed} levels in chunks ({flow.levels_registered_inside()} host levels registered inside the solver, {flow.levels_with_constants_on_device()} with the constant fields held on the device), {on_device} levels on the device"
          f"{' with the originals' if flow.originals_on_device() else ''}", flush=True)
    print("           " + "  ".join(f"{k} {v:.3f}s" for k, v in flow.operator_seconds().items()), flush=True)
    flow.destroy()
    if exp is not None:
        ok = all(bool(np.all(g == e)) for g, e in zip(got, exp))
        print("results identical" if ok else "RESULTS DIFFER")
        if not ok:
            inner = tuple(slice(k // 4, -(k // 4)) for k in (D, H, W))
            for name, g, e in zip("uvw", got, exp):
                bad = g != e
                zs = np.flatnonzero(bad.any(axis=(1, 2)))
                print(f"  {name}: {int(bad.sum())} voxels differ, z planes {zs[:4].tolist()} .. {zs[-4:].tolist()} ({len(zs)} planes), "
                      f"max |diff| {float(np.abs(g - e).max()):.3e}; interior mean resident {float(e[inner].mean()):+.4f} "
                      f"piecemeal {float(g[inner].mean()):+.4f}", flush=True)
            sys.exit(1)


if __name__ == "__main__":
    main()
