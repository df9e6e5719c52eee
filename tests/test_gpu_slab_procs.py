"""The one-process-per-rank slab driver, for real: N processes, each with its own HIP context, drive OpticalFlowSlab with
local_ranks = {rank} exactly as bench.py --gpus N does; only the transport differs (shared-memory mailboxes instead of
RCCL, because RCCL refuses two ranks on one device and the box has one GPU).  The union of the slabs must equal the
single-GPU flow bit for bit."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

from conftest import same

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_ranks(n, dims, tmp_path, env=None, **kw):
    session = uuid.uuid4().hex[:12]
    procs, outs = [], []
    for r in range(n):
        out = str(tmp_path / f"rank{r}.npz")
        outs.append(out)
        cmd = [sys.executable, os.path.join(HERE, "slab_proc_worker.py"), str(r), str(n), session, *map(str, dims), out]
        cmd += [f"{k}={v}" for k, v in kw.items()]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env={**os.environ, **(env or {})}))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-2000:]}"
    parts = [np.load(o) for o in outs]
    run_ranks.overlapped = [int(p["overlapped"]) for p in parts]
    run_ranks.batched = [int(p["batched"]) for p in parts]
    run_ranks.gathered = [int(p["gathered"]) for p in parts]
    return [sum(p[c] for p in parts) for c in "uvw"]   # slabs are disjoint, the other planes are zero


@pytest.mark.parametrize("n_ranks,dims,kw", [
    (2, (48, 40, 44), dict(warp_levels_count=14, outer_iterations_count=4)),
    (3, (70, 33, 26), dict(warp_levels_count=9, outer_iterations_count=3, inner_iterations_count=3, median_radius=3)),
    (4, (40, 36, 40), dict()),   # full defaults: thin slabs, ranks that own nothing on coarse levels, multi-hop halos
])
def test_processes_equal_single_gpu(f3d, tmp_path, n_ranks, dims, kw):
    W, H, D = dims
    f0, f1 = f3d.synth_pair(W, H, D)
    flow = f3d.OpticalFlow()
    flow.initialize(W, H, D)
    exp = flow.compute(f0, f1, silent=True, **kw)
    flow.destroy()
    got = run_ranks(n_ranks, dims, tmp_path, **kw)
    for g, e, c in zip(got, exp, "uvw"):
        assert same(g, e), f"{n_ranks} processes: component {c} differs, max {np.abs(g - e).max():.3e}"


@pytest.mark.parametrize("n_ranks,dims,min_planes", [(2, (40, 36, 64), "24"), (3, (33, 30, 84), "24"), (2, (40, 36, 64), "0")])
def test_overlapped_exchange_order(f3d, tmp_path, n_ranks, dims, min_planes):
    """Slabs of 28-32 planes: the finest levels run the overlapped order (zones first, transfer beside the interior,
    optical_flow_slab.cpp SweepsOverlapped), the coarse ones the plain order; "0" switches the overlap off.  Same bits."""
    W, H, D = dims
    kw = dict(warp_levels_count=6, outer_iterations_count=5)
    f0, f1 = f3d.synth_pair(W, H, D)
    flow = f3d.OpticalFlow()
    flow.initialize(W, H, D)
    exp = flow.compute(f0, f1, silent=True, **kw)
    flow.destroy()
    got = run_ranks(n_ranks, dims, tmp_path, env={"F3D_OVERLAP_MIN_PLANES": min_planes}, **kw)
    if min_planes == "0":
        assert run_ranks.overlapped == [0] * n_ranks
    else:   # up to 6 levels x 4 of the 5 outer iterations; the coarsest levels of the 3-rank case fall below 24 planes
        assert all(8 <= c <= 24 for c in run_ranks.overlapped), run_ranks.overlapped
    for g, e, c in zip(got, exp, "uvw"):
        assert same(g, e), f"{n_ranks} processes: component {c} differs, max {np.abs(g - e).max():.3e}"


@pytest.mark.parametrize("n_ranks,forced,halo", [(4, "4", "32"), (3, "1", "16"), (4, "3", "40")])
def test_outer_iterations_per_exchange_across_processes(f3d, tmp_path, n_ranks, forced, halo):
    """Thin slabs with n (K + 1) halo planes per exchange (n forced to 4, 1 and 3): the pack / transfer / unpack path of the
    one-rank-per-process driver with halos that reach across several ranks.  Same bits as one GPU, same count on every rank."""
    dims = (44, 36, 50)
    kw = dict(warp_levels_count=10, outer_iterations_count=7)
    f0, f1 = f3d.synth_pair(*dims)
    flow = f3d.OpticalFlow()
    flow.initialize(*dims)
    exp = flow.compute(f0, f1, silent=True, **kw)
    flow.destroy()
    got = run_ranks(n_ranks, dims, tmp_path, env={"F3D_SLAB_OUTER_PER_EXCHANGE": forced, "F3D_TEST_HALO_CAPACITY": halo}, **kw)
    expect = 0 if forced == "1" else 10 * 2   # 7 outer iterations = 4 + 3 or 3 + 3 + 1: two groups of more than one per level
    assert run_ranks.batched == [expect] * n_ranks, run_ranks.batched
    for g, e, c in zip(got, exp, "uvw"):
        assert same(g, e), f"{n_ranks} processes, n = {forced}: component {c} differs, max {np.abs(g - e).max():.3e}"


def test_gathered_frame_across_processes(f3d, tmp_path):
    """The warp reaching beyond the halo room with one rank per process: frame 1 is gathered from every rank the reach spans through
    the transport (pack, grouped send / recv, unpack into the wide container) -- 4 processes, slabs of 10 planes and less, 7 planes
    of halo room, a pair that moves 3 planes along z.  Same bits as one GPU, and every rank took the road."""
    dims = (40, 36, 40)
    kw = dict(warp_levels_count=12, outer_iterations_count=4)
    f0, _ = f3d.synth_pair(*dims)
    f1 = np.ascontiguousarray(np.roll(f0, 3, axis=0))
    flow = f3d.OpticalFlow()
    flow.initialize(*dims)
    exp = flow.compute(f0, f1, silent=True, **kw)
    flow.destroy()
    got = run_ranks(4, dims, tmp_path, env={"F3D_TEST_HALO_CAPACITY": "7", "F3D_TEST_ROLL": "3"}, **kw)
    assert all(c >= 1 for c in run_ranks.gathered), run_ranks.gathered
    for g, e, c in zip(got, exp, "uvw"):
        assert same(g, e), f"4 processes with a gathered frame: component {c} differs, max {np.abs(g - e).max():.3e}"


def test_exchange_after_every_solver_stage_across_processes(f3d, tmp_path):
    """F3D_SLAB_EXCHANGE=stage with one rank per process (3 processes): pack / transfer / unpack after every solver stage"""
    dims = (44, 36, 50)
    kw = dict(warp_levels_count=10, outer_iterations_count=4)
    f0, f1 = f3d.synth_pair(*dims)
    flow = f3d.OpticalFlow()
    flow.initialize(*dims)
    exp = flow.compute(f0, f1, silent=True, **kw)
    flow.destroy()
    got = run_ranks(3, dims, tmp_path, env={"F3D_SLAB_EXCHANGE": "stage"}, **kw)
    assert run_ranks.overlapped == [0, 0, 0]
    for g, e, c in zip(got, exp, "uvw"):
        assert same(g, e), f"3 processes exchanging per stage: component {c} differs, max {np.abs(g - e).max():.3e}"


@pytest.mark.parametrize("n_ranks,dims", [(2, (40, 36, 64)), (3, (33, 30, 84))])
def test_stage_exchanges_hidden_behind_the_interior(f3d, tmp_path, n_ranks, dims):
    """F3D_SLAB_EXCHANGE=stage on slabs of 28-32 planes: after every solver stage the planes the neighbours wait for are computed
    first, the transfer runs beside the stage's interior, then the halos are unpacked (round 4; the per-outer-iteration order has
    hidden its exchanges since round 1).  Same bits as one GPU, and the path ran on every rank."""
    kw = dict(warp_levels_count=6, outer_iterations_count=5)
    f0, f1 = f3d.synth_pair(*dims)
    flow = f3d.OpticalFlow()
    flow.initialize(*dims)
    exp = flow.compute(f0, f1, silent=True, **kw)
    flow.destroy()
    got = run_ranks(n_ranks, dims, tmp_path, env={"F3D_SLAB_EXCHANGE": "stage", "F3D_OVERLAP_MIN_PLANES": "24"}, **kw)
    assert all(c >= 8 for c in run_ranks.overlapped), run_ranks.overlapped     # outer iterations whose stage exchanges were hidden
    for g, e, c in zip(got, exp, "uvw"):
        assert same(g, e), f"{n_ranks} processes, stage exchanges hidden: component {c} differs, max {np.abs(g - e).max():.3e}"
