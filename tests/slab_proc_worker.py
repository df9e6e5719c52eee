"""One rank of the multi-process slab rehearsal (tests/test_gpu_slab_procs.py): its own process, its own HIP context on
GPU 0, halos through the shared-memory transport (F3D_COMM_BACKEND=shm; RCCL refuses two ranks on one device).
argv: rank n_ranks session W H D out.npz [key=value ...]"""
import importlib
import os
import sys

import numpy as np

os.environ["F3D_COMM_BACKEND"] = "shm"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, n = int(sys.argv[1]), int(sys.argv[2])
session = sys.argv[3]
W, H, D = (int(a) for a in sys.argv[4:7])
out = sys.argv[7]
kw = {}
for a in sys.argv[8:]:
    k, v = a.split("=")
    kw[k] = float(v) if "." in v else int(v)
f3d = importlib.import_module("cuda-flow3d_amd")
f3d.comm_init(("f3dshm:" + session).encode(), rank, n, device=0)
f0, f1 = f3d.synth_pair(W, H, D)
roll = int(os.environ.get("F3D_TEST_ROLL", "0"))   # frame 1 = frame 0 moved by this many planes along z (warp reach tests)
if roll:
    f1 = np.ascontiguousarray(np.roll(f0, roll, axis=0))
flow = f3d.SlabOpticalFlow(n, [rank], halo_capacity=int(os.environ.get("F3D_TEST_HALO_CAPACITY", "16")))
flow.initialize(W, H, D)
u, v, w = flow.compute(f0, f1, **kw)      # every rank fills the planes it owns, the rest stays zero
overlapped = flow.overlapped_iterations()
batched = flow.batched_exchanges()
gathered = flow.gathered_warps()
flow.destroy()
f3d.comm_destroy()
np.savez(out, u=u, v=v, w=w, overlapped=overlapped, batched=batched, gathered=gathered)
