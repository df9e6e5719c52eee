#!/usr/bin/env python3
"""sha256 of the (u, v, w) a default solve of the synthetic pair produces: run it under different switches (F3D_UDIV=0,
F3D_FUSED_SWEEPS=0, F3D_ZCHUNK=..., F3D_XCD_REMAP=0) and compare the digests -- every one of them must leave the bits alone.
   python tools/digest_solve.py --size 640"""
import argparse
import hashlib
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=384)
a = ap.parse_args()
pkg = importlib.import_module("cuda-flow3d_amd")
n = a.size
f0, f1 = pkg.synth_pair(n, n, n)
flow = pkg.OpticalFlow()
flow.initialize(n, n, n)
flow.upload(f0, f1)
secs = flow.compute_resident(silent=True)
h = hashlib.sha256()
result = flow.download()
for x in result:
    h.update(np.ascontiguousarray(x + np.float32(0.0)).tobytes())
flow.destroy()
switches = {k: v for k, v in os.environ.items() if k.startswith("F3D_")}
print(f"{n}^3 {secs:.2f} s {h.hexdigest()} {switches}")
print(f"per-plane digest (tests/golden/config_digests.json *_plane_sha256) {pkg.combine_plane_digests(pkg.flow_plane_digests(result))}")
