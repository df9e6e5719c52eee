#!/usr/bin/env python3
"""Whole-volume host <-> device copies above 4 GiB (1024 x 1024 x 1200 floats), pageable and page-locked, whole and in
pieces: every plane must come back as it went.   python tools/big_copy_check.py"""
import importlib, os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("cuda-flow3d_amd")
hip = pkg.hip()
W, H, D = 1024, 1024, 1200
box = pkg.Containers(W, H, D)
rng = np.random.default_rng(7)
slab = rng.uniform(-1, 1, (16, H, W)).astype(np.float32)
vol = np.tile(slab, (D // 16, 1, 1)) + np.arange(D, dtype=np.float32)[:, None, None]   # every plane distinct
print("host volume", vol.nbytes / 2**30, "GiB", flush=True)
p = box.alloc(fill=0xFF)
for pinned in (False, True):
    if pinned:
        pkg.check(hip.f3d_host_register(vol.ctypes.data_as(C.c_void_p), vol.nbytes))
    box.upload(p, vol)            # one f3d_copy3d_h2d of the whole volume
    back = box.download(p, (W, H, D))
    bad = np.flatnonzero((back != vol).any(axis=(1, 2)))
    print(f"pinned={pinned}: {len(bad)} planes differ after upload+download", bad[:5], bad[-5:] if len(bad) else "", flush=True)
    # which direction: upload in pieces, download whole
    hipbox = box
    for z in range(0, D, 100):
        box.upload(p, vol[z:z + 100], plane0=z)
    back = box.download(p, (W, H, D))
    bad = np.flatnonzero((back != vol).any(axis=(1, 2)))
    print(f"pinned={pinned}: {len(bad)} planes differ after pieced upload + whole download", bad[:5], flush=True)
    pieces = np.concatenate([box.download(p, (W, H, 100), plane0=z) for z in range(0, D, 100)])
    bad = np.flatnonzero((pieces != vol).any(axis=(1, 2)))
    print(f"pinned={pinned}: {len(bad)} planes differ after pieced upload + pieced download", bad[:5], flush=True)
    pkg.check(hip.f3d_memset2d(p, box.size4.pitch, 0xFF, W * 4, H * D))
box.free()
