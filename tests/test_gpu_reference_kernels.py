"""The reference's own kernels against the oracle and against the product -- on the MI355X.

oracle/_ref/*.hsaco are the reference's entire_data .cu files compiled for gfx950 from /root/reference by oracle/Makefile (as HIP
source, unmodified; contraction off, IEEE division and square root) and launched by tests/ref_kernels.py with the reference
operators' own block sizes, shared-memory sizes and argument lists.  For each of compute_phi_ksi_3d, solve_3d, median_3d (3, 5, 7),
registration_3d, resample_{x,y,z}_3d and the three convolution kernels, on the same seeded inputs:

    reference kernel == oracle      -- what pins oracle/f3d_oracle.c (A.1 - A.6 of SURVEY.md) to the reference's source;
    product kernel   == reference   -- the parity claim itself, without the oracle in between.

Bit for bit (uint32 views; the median and the warp pass values through, so +0 / -0 are compared as values there as in the rest of
the suite).  Skipped when the code objects are absent (a tree built where /root/reference does not exist)."""
import ctypes as C

import numpy as np
import pytest

import ref_kernels
from conftest import bit_same, box_in_container, same
from test_gpu_kernels import CASES, RESAMPLE, SPACINGS, Dev, solver_inputs

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_kernels.available(), reason="oracle/_ref/*.hsaco not built (no /root/reference here)")]


@pytest.fixture
def rig(f3d):
    made = []

    def make(cdims):
        dev = Dev(f3d, cdims)
        ref = ref_kernels.RefKernels(dev.cont)
        made.append((dev, ref))
        return dev, ref

    yield make
    for dev, ref in made:
        ref.close()
        dev.close()


def box(a, dims):
    w, h, d = dims
    return a[:d, :h, :w]


def differing(a, b):
    return int(np.count_nonzero(np.ascontiguousarray(a).view(np.uint32) != np.ascontiguousarray(b).view(np.uint32)))


@pytest.mark.parametrize("dims,cdims", CASES)
@pytest.mark.parametrize("h", SPACINGS)
def test_phi_ksi_and_two_sweeps(f3d, oracle, rig, dims, cdims, h):
    """compute_phi_ksi_3d, then solve_3d twice with the increments swapped in between (cuda_operation_solve.cpp:189-254)"""
    rng = np.random.default_rng(hash((dims, h, "ref")) % 2**32)
    W, H, D = dims
    arrs = solver_inputs(rng, dims, cdims)
    eps_s, eps_d, alpha = 0.001, 0.002, 7.5
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, eps_s, eps_d)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, alpha)
    s2 = oracle.solve_sweep(*arrs[:5], *s1, phi_o, ksi_o, dims, h, alpha)
    dev, ref = rig(cdims)
    hip = f3d.hip()
    ptr = [dev.put(a) for a in arrs]
    # the reference
    r_phi, r_ksi = dev.out(), dev.out()
    a_out, b_out = [dev.out() for _ in range(3)], [dev.out() for _ in range(3)]
    f3d.sync()
    ref.phi_ksi(*ptr, dims, h, eps_s, eps_d, r_phi, r_ksi)
    ref.solve_sweep(*ptr, r_phi, r_ksi, dims, h, alpha, *a_out)
    ref.solve_sweep(*ptr[:5], *a_out, r_phi, r_ksi, dims, h, alpha, *b_out)
    got = {"phi": dev.get(r_phi), "ksi": dev.get(r_ksi)}
    for n, p, q in zip("uvw", a_out, b_out):
        got["d" + n + " after one sweep"], got["d" + n + " after two"] = dev.get(p), dev.get(q)
    exp = {"phi": phi_o, "ksi": ksi_o}
    for n, e1, e2 in zip("uvw", s1, s2):
        exp["d" + n + " after one sweep"], exp["d" + n + " after two"] = e1, e2
    for name in exp:
        assert bit_same(box(got[name], dims), box(exp[name], dims)), \
            f"reference vs oracle, {name}: {differing(box(got[name], dims), box(exp[name], dims))} voxels differ"
    # the product: separate launches, then the fused ones
    p_phi, p_ksi = dev.out(), dev.out()
    f3d.check(hip.f3d_phi_ksi(*ptr, W, H, D, *h, eps_s, eps_d, p_phi, p_ksi, None))
    p1, p2 = [dev.out() for _ in range(3)], [dev.out() for _ in range(3)]
    f3d.check(hip.f3d_solve_sweep(*ptr, p_phi, p_ksi, W, H, D, *h, alpha, *p1, None))
    f3d.check(hip.f3d_solve_sweep2(*ptr, p_phi, p_ksi, W, H, D, *h, alpha, *p2, None))
    assert bit_same(box(dev.get(p_phi), dims), box(got["phi"], dims)) and bit_same(box(dev.get(p_ksi), dims), box(got["ksi"], dims))
    for n, a, b in zip("uvw", p1, p2):
        assert bit_same(box(dev.get(a), dims), box(got["d" + n + " after one sweep"], dims)), f"product vs reference: d{n}, one sweep"
        assert bit_same(box(dev.get(b), dims), box(got["d" + n + " after two"], dims)), f"product vs reference: d{n}, two fused sweeps"


@pytest.mark.parametrize("r", [3, 5, 7])
@pytest.mark.parametrize("dims,cdims", [((37, 21, 9), (64, 32, 16)), ((9, 6, 4), (9, 6, 4)), ((70, 33, 5), (70, 33, 5)), ((40, 17, 22), (64, 20, 22))])
def test_median(f3d, oracle, rig, dims, cdims, r):
    rng = np.random.default_rng(170 + r)
    W, H, D = dims
    inp = box_in_container(rng, dims, cdims, -2, 2)
    box(inp, dims)[rng.random((D, H, W)) < 0.3] = 0.5
    box(inp, dims)[rng.random((D, H, W)) < 0.1] = 0.0
    exp = oracle.median(inp, dims, r)
    dev, ref = rig(cdims)
    pin, r_out, p_out = dev.put(inp), dev.out(), dev.out()
    f3d.sync()
    ref.median(pin, dims, r, r_out)
    got = dev.get(r_out)
    assert same(box(got, dims), box(exp, dims)), "reference vs oracle"
    f3d.check(f3d.hip().f3d_median(pin, W, H, D, r, p_out, None))
    assert same(box(dev.get(p_out), dims), box(got, dims)), "product vs reference"


@pytest.mark.parametrize("dims,cdims", CASES[:5])
@pytest.mark.parametrize("h", SPACINGS)
def test_registration(f3d, oracle, rig, dims, cdims, h):
    """flows that leave the volume, land on integers, and a few that are NaN (registration_3d.cu:60-64)"""
    rng = np.random.default_rng(hash((dims, h, "warp")) % 2**32)
    W, H, D = dims
    f0, f1 = box_in_container(rng, dims, cdims, 0, 255), box_in_container(rng, dims, cdims, 0, 255)
    flows = [box_in_container(rng, dims, cdims, -6, 6) for _ in range(3)]
    for f in flows:
        b = box(f, dims)
        b[rng.random((D, H, W)) < 0.1] = np.float32(2.0)
        b[rng.random((D, H, W)) < 0.02] = np.nan
    exp = oracle.warp(f0, f1, *flows, dims, h)
    dev, ref = rig(cdims)
    p = [dev.put(a) for a in (f0, f1, *flows)]
    r_out, p_out = dev.out(), dev.out()
    f3d.sync()
    ref.warp(*p, dims, h, r_out)
    got = dev.get(r_out)
    assert bit_same(box(got, dims), box(exp, dims)), f"reference vs oracle: {differing(box(got, dims), box(exp, dims))} voxels differ"
    f3d.check(f3d.hip().f3d_warp(*p, W, H, D, *h, p_out, None))
    assert bit_same(box(dev.get(p_out), dims), box(got, dims)), "product vs reference"


@pytest.mark.parametrize("src,dst", RESAMPLE + [((150, 40, 33), (143, 38, 31)), ((200, 90, 70), (37, 17, 13))])
def test_resample(f3d, oracle, rig, src, dst):
    rng = np.random.default_rng(130)
    cdims = tuple(max(a, b) + 3 for a, b in zip(src, dst))
    inp = box_in_container(rng, src, cdims, -5, 5)
    exp = oracle.resample(inp, src, dst)
    dev, ref = rig(cdims)
    pin, r_out, r_tmp = dev.put(inp), dev.out(), dev.out()
    f3d.sync()
    ref.resample(pin, r_out, r_tmp, src, dst)
    got = dev.get(r_out)
    assert bit_same(box(got, dst), box(exp, dst)), f"reference vs oracle: {differing(box(got, dst), box(exp, dst))} voxels differ"
    op = f3d.Operation("resample")
    assert op.initialize(dev.cont)
    p_out, p_tmp = dev.out(), dev.out()
    op.execute(dev_input=pin, dev_output=p_out, dev_temp=p_tmp, data_size=src, resample_size=dst)
    assert bit_same(box(dev.get(p_out), dst), box(got, dst)), "product vs reference"
    op.destroy()


@pytest.mark.parametrize("sigma", [1.0, 2.0, 3.5])
@pytest.mark.parametrize("dims", [(37, 20, 9), (64, 8, 5), (130, 12, 33)])
def test_gaussian(f3d, oracle, rig, dims, sigma):
    """The container height equals the data height and is a multiple of four, as at the reference's only call site: outside that the
    reference's column kernel reads rows of the next plane (finding F8) and there is no reference value to compare with.
    Radius 3 sigma <= 16: the halo of the reference's tiles (convolution_3d.cu:49, :73)."""
    rng = np.random.default_rng(190)
    W, H, D = dims
    cdims = (W + 5, H, D)
    inp = box_in_container(rng, dims, cdims, 0, 255)
    exp = oracle.gaussian(inp, dims, sigma)
    radius, taps = oracle.gaussian_taps(sigma)
    dev, ref = rig(cdims)
    pin, r_out, r_tmp = dev.put(inp), dev.out(), dev.out()
    f3d.sync()
    ref.gaussian(pin, r_out, r_tmp, dims, [float(t) for t in taps], int(radius))
    got = dev.get(r_out)
    assert bit_same(box(got, dims), box(exp, dims)), f"reference vs oracle: {differing(box(got, dims), box(exp, dims))} voxels differ"
    op = f3d.Operation("convolution")
    assert op.initialize(dev.cont)
    p_out, p_tmp = dev.out(), dev.out()
    op.execute(dev_input=pin, dev_output=p_out, dev_temp=p_tmp, data_size=dims, gaussian_sigma=sigma)
    assert bit_same(box(dev.get(p_out), dims), box(got, dims)), "product vs reference"
    op.destroy()
