#!/usr/bin/env python3
"""Frame-sequence throughput of bin/flow3d: N frames of one size written as RAW float32, then `flow3d --frames ...` twice --
pipelined (default: the next frame uploads and the previous flow downloads and is written beside the running solve) and with
F3D_SEQ_SERIAL=1 (the same work one step after the other), and with --concurrent 2 / 3 (that many pairs solved at once on lanes of
their own: what sequences of SMALL volumes want).   python tools/sequence_bench.py [--size 384] [--frames 6] [--concurrent 2 3]"""
import argparse
import importlib
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=384)
ap.add_argument("--frames", type=int, default=6)
ap.add_argument("--concurrent", type=int, nargs="*", default=[2])
a = ap.parse_args()
pkg = importlib.import_module("cuda-flow3d_amd")
S, N = a.size, a.frames
f0, f1 = pkg.synth_pair(S, S, S)
exe = os.path.join(ROOT, "cuda-flow3d_amd", "bin", "flow3d")
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
    paths = []
    for k in range(N):
        t = k / max(1, N - 1)
        p = os.path.join(tmp, f"f{k}.raw")
        ((1 - t) * f0 + t * f1).astype(np.float32).tofile(p)
        paths.append(p)
    for tag, env in (("pipelined", {}), ("serial", {"F3D_SEQ_SERIAL": "1"})):
        t0 = time.time()
        run = subprocess.run([exe, "--dims", str(S), str(S), str(S), "--f32", "--frames", *paths, "--out", os.path.join(tmp, tag),
                              "--silent"], capture_output=True, text=True, env={**os.environ, **env})
        wall = time.time() - t0
        dev = [float(x) for x in re.findall(r"pair \d+ of \d+: ([\d.]+) s on the device", run.stdout)]
        print(f"{tag:10s} {S}^3 x {N} frames: wall {wall:.2f} s, device {sum(dev):.2f} s over {len(dev)} pairs "
              f"({wall / max(1, len(dev)):.3f} s per pair), rc {run.returncode}", flush=True)
    for n in a.concurrent:
        t0 = time.time()
        run = subprocess.run([exe, "--dims", str(S), str(S), str(S), "--f32", "--frames", *paths, "--out", os.path.join(tmp, f"c{n}"),
                              "--silent", "--concurrent", str(n)], capture_output=True, text=True)
        wall = time.time() - t0
        m = re.search(r"(\d+) pairs in ([\d.]+) s: ([\d.]+) pairs per second", run.stdout)
        print(f"concurrent {n}: {S}^3 x {N} frames: wall {wall:.2f} s, " + (f"{m.group(1)} pairs in {m.group(2)} s = {float(m.group(2)) / int(m.group(1)):.3f} s per pair"
              if m else "no summary line") + f", rc {run.returncode}", flush=True)
        if n == a.concurrent[0]:   # the same bits as the pipelined run
            same = all(open(os.path.join(tmp, f"c{n}_{k}_flow-{c}-{S}-{S}-{S}.raw"), "rb").read() ==
                       open(os.path.join(tmp, f"pipelined_{k}_flow-{c}-{S}-{S}-{S}.raw"), "rb").read() for k in range(N - 1) for c in "uvw")
            print(f"concurrent {n}: results identical to the pipelined run: {same}", flush=True)
