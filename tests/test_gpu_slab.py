"""z-slab decomposition on ONE GPU: the multi-GPU driver run with every rank in this process (halo planes copied
between the ranks' containers instead of sent over RCCL) must reproduce the single-GPU driver bit for bit -- the sweep
is a Jacobi update, so widened windows and exchanged halos carry exactly the neighbour's values."""
import numpy as np
import pytest

from conftest import same

pytestmark = pytest.mark.gpu


def single(f3d, f0, f1, **kw):
    d, h, w = f0.shape
    flow = f3d.OpticalFlow()
    flow.initialize(w, h, d)
    out = flow.compute(f0, f1, silent=True, **kw)
    flow.destroy()
    return out


def slabbed(f3d, f0, f1, n_ranks, **kw):
    d, h, w = f0.shape
    flow = f3d.SlabOpticalFlow(n_ranks, list(range(n_ranks)))
    flow.initialize(w, h, d)
    out = flow.compute(f0, f1, **kw)
    flow.destroy()
    return out


@pytest.mark.parametrize("n_ranks", [1, 2, 3, 8])
def test_slabs_equal_single_gpu(f3d, n_ranks):
    f0, f1 = f3d.synth_pair(48, 40, 44)
    kw = dict(warp_levels_count=14, outer_iterations_count=4)
    exp = single(f3d, f0, f1, **kw)
    got = slabbed(f3d, f0, f1, n_ranks, **kw)
    for g, e, n in zip(got, exp, "uvw"):
        assert same(g, e), f"{n_ranks} slabs: component {n} differs, max {np.abs(g - e).max():.3e}"


def test_thin_slabs_full_defaults(f3d):
    """8 ranks on 40 planes: 5 planes per rank at the finest level, fewer than the 6-plane solver halo, and coarse
    levels where some ranks own nothing -- the exchange plan reaches across several ranks."""
    f0, f1 = f3d.synth_pair(40, 36, 40)
    exp = single(f3d, f0, f1)
    got = slabbed(f3d, f0, f1, 8)
    for g, e, n in zip(got, exp, "uvw"):
        assert same(g, e), f"component {n} differs, max {np.abs(g - e).max():.3e}"


def test_anisotropic_and_no_blur(f3d):
    f0, f1 = f3d.synth_pair(70, 33, 26)
    kw = dict(warp_levels_count=9, outer_iterations_count=3, inner_iterations_count=3, gaussian_sigma=0.0, median_radius=3)
    exp = single(f3d, f0, f1, **kw)
    got = slabbed(f3d, f0, f1, 4, **kw)
    for g, e, n in zip(got, exp, "uvw"):
        assert same(g, e), f"component {n} differs"


def test_single_rank_rccl_init_and_allreduce(f3d):
    """librccl loads, a one-rank communicator comes up and the scalar all-reduce round-trips (the N > 1 exchange itself
    needs N GPUs and is exercised by bench.py --gpus N on the multi-GPU node)."""
    import ctypes as C
    f3d.comm_init(f3d.comm_unique_id(), 0, 1)
    v = C.c_float(3.5)
    f3d.check(f3d.hip().f3d_comm_allreduce_max_f32(C.byref(v)))
    assert v.value == 3.5
    f3d.comm_destroy()
