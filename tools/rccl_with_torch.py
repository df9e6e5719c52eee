#!/usr/bin/env python3
"""bench.py's multi-GPU start-up order on one GPU: native library and /opt/rocm's HIP first, then torch (gloo only), then a
one-rank RCCL communicator, the scalar all-reduce and a pack -> send/recv to itself -> unpack exchange -- checks that the
RCCL the library dlopens and the ROCm copies bundled with torch coexist in one process.   python tools/rccl_with_torch.py"""
import ctypes as C
import faulthandler
import importlib
import os
import sys

faulthandler.dump_traceback_later(420, exit=True)   # a hang names its line instead of eating the time limit

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

pkg = importlib.import_module("cuda-flow3d_amd")
pkg.check(pkg.hip().f3d_init(0), "f3d_init")
print("native library up", flush=True)
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
print("torch imported", flush=True)
dist.init_process_group(backend="gloo", rank=0, world_size=1)
print("gloo up", flush=True)
pkg.comm_init(pkg.comm_unique_id(), 0, 1, device=0)
print("rccl communicator up", flush=True)
v = C.c_float(3.5)
pkg.check(pkg.hip().f3d_comm_allreduce_max_f32(C.byref(v)), "allreduce")
assert v.value == 3.5
print("all-reduce ok", flush=True)
f0, f1 = pkg.synth_pair(48, 40, 32)
flow = pkg.SlabOpticalFlow(1, [0])
flow.initialize(48, 40, 32)
u, v_, w = flow.compute(f0, f1, warp_levels_count=4, outer_iterations_count=2)
flow.destroy()
ref = pkg.OpticalFlow()
ref.initialize(48, 40, 32)
e = ref.compute(f0, f1, silent=True, warp_levels_count=4, outer_iterations_count=2)
ref.destroy()
assert all(np.array_equal(a, b) for a, b in zip((u, v_, w), e))
pkg.comm_destroy()
dist.destroy_process_group()
print("rccl + torch in one process: ok; torch", torch.__version__)
