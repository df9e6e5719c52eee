#!/usr/bin/env python3
"""Frame-sequence throughput of bin/flow3d: N frames of one size written as RAW float32, then `flow3d --frames ...` twice --
pipelined (default: the next frame uploads and the previous flow downloads and is written beside the running solve) and with
F3D_SEQ_SERIAL=1 (the same work one step after the other).   python tools/sequence_bench.py [--size 384] [--frames 6]"""
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
