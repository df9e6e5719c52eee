#!/usr/bin/env python3
"""Differential check of the z-slab driver at a realistic size on ONE GPU: all ranks in this process (plane copies instead
of RCCL), full default schedule, against the single-GPU driver; then the same with one process per rank over the
shared-memory transport (tests/slab_proc_worker.py).   python tools/slab_check.py [--size 384] [--ranks 8] [--procs 4]"""
import argparse
import importlib
import os
import subprocess
import sys
import tempfile
import time
import uuid

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=384)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--procs", type=int, default=4)
ap.add_argument("--proc-size", type=int, default=256)
a = ap.parse_args()
if a.procs > 5:
    sys.exit("at most 5 rank processes: this process holds the GPU too and a gpurun box allows 6")
pkg = importlib.import_module("cuda-flow3d_amd")


def single(n):
    f0, f1 = pkg.synth_pair(n, n, n)
    flow = pkg.OpticalFlow()
    flow.initialize(n, n, n)
    t0 = time.time()
    out = flow.compute(f0, f1, silent=True)
    print(f"single GPU {n}^3: {time.time() - t0:.2f} s", flush=True)
    flow.destroy()
    return f0, f1, out


n = a.size
f0, f1, exp = single(n)
flow = pkg.SlabOpticalFlow(a.ranks, list(range(a.ranks)), halo_capacity=32)
flow.initialize(n, n, n)
t0 = time.time()
got = flow.compute(f0, f1)
print(f"{a.ranks} slabs in one process: {time.time() - t0:.2f} s, {flow.batched_exchanges()} batched exchange groups", flush=True)
flow.destroy()
ok = all(bool(np.array_equal(g, e)) for g, e in zip(got, exp))
print("virtual ranks:", "identical" if ok else "DIFFER", flush=True)

m = a.proc_size
_, _, exp = single(m)
with tempfile.TemporaryDirectory() as tmp:
    session = uuid.uuid4().hex[:12]
    procs, outs = [], []
    for r in range(a.procs):
        out = os.path.join(tmp, f"rank{r}.npz")
        outs.append(out)
        cmd = [sys.executable, os.path.join(ROOT, "tests", "slab_proc_worker.py"), str(r), str(a.procs), session, str(m), str(m), str(m), out]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT,
                                      env={**os.environ, "F3D_TEST_HALO_CAPACITY": "32"}))
    t0 = time.time()
    rcs = [p.wait(timeout=600) for p in procs]
    print(f"{a.procs} processes {m}^3: {time.time() - t0:.2f} s, return codes {rcs}", flush=True)
    parts = [np.load(o) for o in outs]
    got = [sum(p[c] for p in parts) for c in "uvw"]
    print("overlapped iterations per rank", [int(p["overlapped"]) for p in parts], "batched", [int(p["batched"]) for p in parts])
ok2 = all(bool(np.array_equal(g, e)) for g, e in zip(got, exp))
print("processes:", "identical" if ok2 else "DIFFER", flush=True)
sys.exit(0 if ok and ok2 else 1)
