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
import os

import numpy as np
import pytest

import ref_kernels
from conftest import bit_same, box_in_container, same
from test_gpu_kernels import CASES, RESAMPLE, SPACINGS, Dev, solver_inputs

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_kernels.available(), reason="oracle/_ref/*.hsaco not built (no /root/reference here)")]


class GuardedDev:
    """`Dev` with room around every container.  The reference's kernels read outside their buffers where the values are never
    looked at: median_3d.cu:109-113 forms plane 2 D - offset - 2 = -1 for the rear halo of a five-plane volume (size_t arithmetic:
    the plane BEFORE the buffer), and the same pattern exists for rows and columns.  On the reference's own allocations that is a
    read of a neighbouring buffer; here an unmapped address is a GPU memory fault, so every container handed to a reference kernel
    sits `margin` planes inside a larger allocation (NaN-poisoned like the rest)."""

    def __init__(self, f3d, cdims, margin=8):
        wc, hc, dc = cdims
        self.f3d, self.cdims, self.margin = f3d, cdims, margin
        self.alloc = f3d.Containers(wc, hc, dc + 2 * margin)
        self.alloc.alloc(fill=0xFF)
        self.cont = f3d.Containers(wc, hc, dc)          # the view the kernels are told about: same pitch and height, dc planes
        self.cont.pitch = self.alloc.pitch
        self.cont.set_current()
        self.offset = margin * self.alloc.pitch * hc

    def put(self, host_container):
        p = self.alloc.new()
        self.alloc.upload(p, host_container, plane0=self.margin)
        return p + self.offset

    def out(self):
        return self.alloc.new() + self.offset

    def get(self, p):
        self.f3d.sync()
        return self.alloc.download(p - self.offset, self.cdims, plane0=self.margin)

    def close(self):
        self.f3d.sync()
        self.alloc.free()


@pytest.fixture
def rig(f3d):
    made = []

    def make(cdims):
        dev = GuardedDev(f3d, cdims)
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


def test_the_code_objects_about_to_be_loaded_are_the_pinned_build():
    """what ran is what the recipe gives: the .text of every oracle/_ref/*.hsaco on THIS box equals tests/golden/ref_hsaco_manifest.json
    (tests/test_oracle.py rebuilds them from the reference tree in the build container and holds the result to the same manifest)"""
    import hsaco_text
    want = {k: v for k, v in hsaco_text.manifest().items() if not k.startswith("_")}
    assert sorted(want) == sorted(m + ".hsaco" for m in ref_kernels.MODULES)
    for name, digest in want.items():
        assert hsaco_text.text_sha256(os.path.join(ref_kernels.REF_DIR, name)) == digest, name



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


def test_the_product_outruns_the_reference_kernels_on_the_same_gpu(f3d, rig, capsys):
    """SURVEY.md 8d's sample -- phi/ksi + 5 sweeps of one outer iteration -- on a 256^3 level: the reference's kernels (compiled for
    this GPU, launched as its operator launches them) against the product's three launches.  Same bits; the time ratio is what a
    straight recompile of the CUDA kernels would have left on the table (printed, and written to $F3D_OUT when set)."""
    import os
    import time
    S = 256
    dims = (S, S, S)
    rng = np.random.default_rng(7)
    plane = lambda lo, hi: np.repeat(rng.uniform(lo, hi, size=(1, S, S)).astype(np.float32), S, axis=0) + \
        rng.uniform(-0.01, 0.01, size=(S, 1, 1)).astype(np.float32)
    arrs = [plane(0, 255), plane(0, 255), plane(-3, 3), plane(-3, 3), plane(-3, 3), plane(-.5, .5), plane(-.5, .5), plane(-.5, .5)]
    h, eps_s, eps_d, alpha = (1.0, 1.0, 1.0), 0.001, 0.001, 7.5
    dev, ref = rig(dims)
    hip = f3d.hip()
    ptr = [dev.put(a) for a in arrs]
    del arrs

    bufs = {who: ([dev.out() for _ in range(3)], [dev.out() for _ in range(3)], [dev.out() for _ in range(4)]) for who in ("ref", "pro")}

    def reference():   # the inputs are read only: the increments ping-pong between two buffer triples of the run's own
        x, y, (phi, ksi, _, _) = bufs["ref"]
        f3d.sync()
        t = time.perf_counter()
        ref.phi_ksi(*ptr, dims, h, eps_s, eps_d, phi, ksi)
        src = ptr[5:8]
        for i in range(5):
            dst = x if i % 2 == 0 else y
            ref.solve_sweep(*ptr[:5], *src, phi, ksi, dims, h, alpha, *dst)
            src = dst
        return time.perf_counter() - t, src, (phi, ksi)

    def product():
        x, y, (phi, ksi, phi2, ksi2) = bufs["pro"]
        f3d.sync()
        t = time.perf_counter()
        f3d.check(hip.f3d_phi_ksi(*ptr, S, S, S, *h, eps_s, eps_d, phi, ksi, None))
        f3d.check(hip.f3d_solve_sweep2(*ptr[:5], *ptr[5:8], phi, ksi, S, S, S, *h, alpha, *x, None))
        f3d.check(hip.f3d_solve_sweep2(*ptr[:5], *x, phi, ksi, S, S, S, *h, alpha, *y, None))
        f3d.check(hip.f3d_solve_sweep_phi_ksi(*ptr[:5], *y, phi, ksi, S, S, S, *h, alpha, eps_s, eps_d, *x, phi2, ksi2, None))
        f3d.sync()
        return time.perf_counter() - t, x, (phi, ksi)

    reference()
    product()                     # warm: code objects loaded, scratch allocated
    t_ref, r_out, r_w = reference()
    t_pro, p_out, p_w = product()
    for r, p in zip(list(r_out) + list(r_w), list(p_out) + list(p_w)):
        assert bit_same(dev.get(r), dev.get(p))
    line = (f"phi/ksi + 5 sweeps on {S}^3: the reference's kernels {t_ref * 1e3:.2f} ms, the product {t_pro * 1e3:.2f} ms "
            f"(x {t_ref / t_pro:.1f}); {6 * S ** 3 / t_pro / 1e9:.1f} against {6 * S ** 3 / t_ref / 1e9:.1f} Gvoxel-updates/s")
    with capsys.disabled():
        print("\n" + line)
    if os.environ.get("F3D_OUT"):
        with open(os.path.join(os.environ["F3D_OUT"], "reference_kernels_vs_product.txt"), "a") as f:
            f.write(line + "\n")
    assert t_ref > 2.0 * t_pro, line


@pytest.mark.parametrize("dims,cdims", CASES[:4])
def test_add(f3d, oracle, rig, dims, cdims):
    rng = np.random.default_rng(5)
    W, H, D = dims
    a, b = box_in_container(rng, dims, cdims, -3, 3), box_in_container(rng, dims, cdims, -0.5, 0.5)
    exp = a.copy()
    oracle.add(exp, b, dims)
    dev, ref = rig(cdims)
    ra, pa, pb = dev.put(a), dev.put(a), dev.put(b)
    f3d.sync()
    ref.add(ra, pb, dims)
    got = dev.get(ra)
    assert bit_same(box(got, dims), box(exp, dims)), "reference vs oracle"
    f3d.check(f3d.hip().f3d_add(pa, pb, W, H, D, None))
    assert bit_same(box(dev.get(pa), dims), box(got, dims)), "product vs reference"


def reference_pyramid(f3d, oracle, dev, ref, f0, f1, prm):
    """OpticalFlowE::ComputeFlow (optical_flow_e.cpp:208-576) with EVERY kernel launch made on the reference's own kernels: the blur of
    both frames, and per level the two frame resamplings from the full-size frames, the flow of the level before brought up, the
    registration, outer x (phi/ksi + inner x sweep with the buffer swap), flow += increment, the median.  Clearing and copying are the
    library's (cuMemsetD2D8 / the container stack in the reference).  Returns (u, v, w) as host arrays."""
    hip = f3d.hip()
    d0, h0, w0 = f0.shape
    full = (w0, h0, d0)
    cont = dev.cont
    rows = cont.height * cont.depth

    def clear(p, width):
        f3d.check(hip.f3d_memset2d(p, cont.pitch, 0, width * 4, rows))
        f3d.sync()

    raw0, raw1 = dev.put(f0), dev.put(f1)
    fr0, fr1, tmp = dev.out(), dev.out(), dev.out()
    f3d.sync()
    if prm["gaussian_sigma"] > 0:
        radius, taps = oracle.gaussian_taps(prm["gaussian_sigma"])
        ref.gaussian(raw0, fr0, tmp, full, [float(t) for t in taps], int(radius))
        ref.gaussian(raw1, fr1, tmp, full, [float(t) for t in taps], int(radius))
    else:
        fr0, fr1 = raw0, raw1
    fr0_res, fr1_res = dev.out(), dev.out()
    flow = [dev.out() for _ in range(3)]
    incr = [dev.out() for _ in range(3)]
    phi, ksi = dev.out(), dev.out()
    tinc = [dev.out() for _ in range(3)]
    level = min(prm["warp_levels_count"], oracle.max_warp_level(w0, h0, d0, prm["warp_scale_factor"])) - 1
    prev = None
    while level >= 0:
        cur, h = oracle.level_geometry(w0, h0, d0, prm["warp_scale_factor"], level)
        if level == 0:
            fr0, fr0_res = fr0_res, fr0
            fr1, fr1_res = fr1_res, fr1
        else:
            ref.resample(fr0, fr0_res, tmp, full, cur)
            ref.resample(fr1, fr1_res, tmp, full, cur)
        if prev is None:
            for p in flow:
                clear(p, cont.width)
        else:
            for i in range(3):
                ref.resample(flow[i], incr[i], tmp, prev, cur)
                flow[i], incr[i] = incr[i], flow[i]
        ref.warp(fr0_res, fr1_res, *flow, cur, h, tmp)
        fr1_res, tmp = tmp, fr1_res
        for p in incr:
            clear(p, cur[0])
        for _ in range(prm["outer_iterations_count"]):
            ref.phi_ksi(fr0_res, fr1_res, *flow, *incr, cur, h, prm["equation_smoothness"], prm["equation_data"], phi, ksi)
            for _ in range(prm["inner_iterations_count"]):
                ref.solve_sweep(fr0_res, fr1_res, *flow, *incr, phi, ksi, cur, h, prm["equation_alpha"], *tinc)
                incr, tinc = tinc, incr
        for i in range(3):
            ref.add(flow[i], incr[i], cur)
        for i in range(3):
            ref.median(flow[i], cur, prm["median_radius"], tmp)
            flow[i], tmp = tmp, flow[i]
        prev = cur
        level -= 1
    return [dev.get(p) for p in flow]


@pytest.mark.parametrize("shape,prm", [
    ((24, 40, 48), dict(warp_levels_count=6, outer_iterations_count=5)),
    ((5, 64, 96), dict(warp_levels_count=4, outer_iterations_count=4, warp_scale_factor=0.8)),
    ((20, 36, 40), dict(warp_levels_count=3, outer_iterations_count=3, gaussian_sigma=0.0, median_radius=3)),
])
def test_whole_pyramid_on_the_reference_kernels(f3d, oracle, rig, shape, prm):
    """A whole ComputeFlow -- blur, several pyramid levels, registration, solver, flow update, median -- three ways on the same pair:
    the reference's kernels driven in the reference's order, the oracle, the product.  Bit for bit (the median passes values through:
    signs of zero compared as values)."""
    import importlib
    pkg = importlib.import_module("cuda-flow3d_amd")
    params = dict(pkg.DEFAULT_PARAMS)
    params.update(prm)
    d, h, w = shape
    zz, yy, xx = np.meshgrid(np.arange(d), np.arange(h), np.arange(w), indexing="ij")
    blob = lambda cx, cy, cz, s: np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2 + (zz - cz) ** 2) / (2.0 * s * s))
    f0 = (180 * blob(w * 0.4, h * 0.5, d * 0.5, w / 7) + 90 * blob(w * 0.7, h * 0.3, d * 0.4, w / 10) + 5 * np.sin(xx * 0.9) * np.cos(yy * 0.7)).astype(np.float32)
    f1 = (180 * blob(w * 0.4 + 1.5, h * 0.5 - 1.0, d * 0.5 + 0.4, w / 7) + 90 * blob(w * 0.7 + 1.5, h * 0.3 - 1.0, d * 0.4 + 0.4, w / 10) +
          5 * np.sin((xx - 1.5) * 0.9) * np.cos((yy + 1.0) * 0.7)).astype(np.float32)
    dev, ref = rig((w, h, d))
    got = reference_pyramid(f3d, oracle, dev, ref, f0, f1, params)
    exp, _ = oracle.compute_flow(f0, f1, **prm)
    for n, g, e in zip("uvw", got, exp):
        assert same(g, e), f"reference kernels vs oracle, {n}: {int((g != e).sum())} voxels differ, max {np.abs(g - e).max():.3e}"
    flow = f3d.OpticalFlow()
    flow.initialize(w, h, d)
    try:
        mine = flow.compute(f0, f1, silent=True, **prm)
    finally:
        flow.destroy()
    for n, g, m in zip("uvw", got, mine):
        assert same(m, g), f"product vs reference kernels, {n}: {int((g != m).sum())} voxels differ, max {np.abs(g - m).max():.3e}"
    assert max(float(np.abs(c).max()) for c in got) > 1e-3   # not the zero field (a few outer iterations recover a fraction of the shift)


@pytest.mark.parametrize("config", ["c1", "crop128", "croprub", "c2", "c3"])
def test_baseline_configs_on_the_reference_kernels(f3d, oracle, rig, config):
    """The committed golden results of BASELINE configs 1 - 3 and of the two full-default crops (tests/golden/expected_oracle.npz,
    made by the oracle) reproduced by the REFERENCE'S KERNELS driven through the reference's sequence: the digests of (u, v, w) the
    product is held to in tests/test_gpu_pipeline.py are the reference's, not only the restatement's."""
    import importlib
    import os
    from test_gpu_pipeline import GOLD, digest
    pkg = importlib.import_module("cuda-flow3d_amd")
    e = np.load(os.path.join(GOLD, "expected_oracle.npz"))
    i128 = np.load(os.path.join(GOLD, "inputs_128.npz"))
    irub = np.load(os.path.join(GOLD, "inputs_rub.npz"))
    f0, f1 = i128["frame_0"].astype(np.float32), i128["frame_1"].astype(np.float32)
    r0 = np.repeat(irub["slice_0"][None], int(irub["depth"]), axis=0).astype(np.float32)
    r1 = np.repeat(irub["slice_1"][None], int(irub["depth"]), axis=0).astype(np.float32)
    prm = {}
    if config == "c1":
        prm = dict(warp_levels_count=1, outer_iterations_count=1, inner_iterations_count=5)
    elif config == "crop128":
        crop = (slice(40, 64), slice(40, 80), slice(40, 88))
        f0, f1 = f0[crop].copy(), f1[crop].copy()
    elif config == "croprub":
        rc = (slice(0, 5), slice(100, 164), slice(200, 296))
        f0, f1 = r0[rc].copy(), r1[rc].copy()
    elif config == "c3":
        f0, f1 = r0, r1
    params = dict(pkg.DEFAULT_PARAMS)
    params.update(prm)
    d, h, w = f0.shape
    dev, ref = rig((w, h, d))
    got = reference_pyramid(f3d, oracle, dev, ref, f0, f1, params)
    for c in got:
        assert np.isfinite(c).all()
    if config in ("c1", "c2", "c3"):
        assert digest(*got) == str(e[config + "_sha256"])
    else:
        for n, g, x in zip("uvw", got, e[config + "_flow"]):
            assert same(g, x), f"{config}: {n} differs in {int((g != x).sum())} voxels"


def test_baseline_config_4_on_the_reference_kernels(f3d, oracle, rig):
    """BASELINE config 4 -- the 512^3 synthetic pair through the full default pyramid, the configuration bench.py times -- on the
    reference's kernels: the sha256 of (u, v, w) is the digest committed in tests/golden/config_digests.json, the one every bench
    line checks its own result against (`parity.match`).  The oracle cannot reach this size; the reference's kernels
    on the GPU can, in about a minute."""
    import importlib
    from test_gpu_configs import committed, digest
    pkg = importlib.import_module("cuda-flow3d_amd")
    f0, f1 = f3d.synth_pair(512, 512, 512)
    import os
    import time
    dev, ref = rig((512, 512, 512))
    t = time.perf_counter()
    got = reference_pyramid(f3d, oracle, dev, ref, f0, f1, dict(pkg.DEFAULT_PARAMS))
    seconds = time.perf_counter() - t
    for c in got:
        assert np.isfinite(c).all()
    assert digest(got) == committed("c4_512_default_sha256")
    line = (f"BASELINE config 4 on the reference's kernels (this GPU, upload and download included, one host wait per launch): "
            f"{seconds:.1f} s = {512 ** 3 / seconds / 1e6:.1f} Mvoxels/s")
    print("\n" + line)
    if os.environ.get("F3D_OUT"):
        with open(os.path.join(os.environ["F3D_OUT"], "reference_kernels_vs_product.txt"), "a") as f:
            f.write(line + "\n")


@pytest.mark.skipif(os.environ.get("F3D_REF_C5") == "0", reason="F3D_REF_C5=0: the 1024^3 run on the reference's kernels (3.5 minutes, 110 GB) was switched off")
def test_baseline_config_5_on_the_reference_kernels(f3d, oracle, rig):
    """BASELINE config 5 -- 1024^3, the configuration of the multi-GPU runs -- on the reference's kernels against the committed digest
    (the one the resident, 8-slab and rank-process runs and every `config5` record of bench.py are held to).  3.5 minutes on the
    reference's kernels; in the default run since round 4 (F3D_REF_C5=0 leaves it out of a quick local run)."""
    import importlib
    import os
    import time
    from test_gpu_configs import committed, digest
    pkg = importlib.import_module("cuda-flow3d_amd")
    f0, f1 = f3d.synth_pair(1024, 1024, 1024)
    dev, ref = rig((1024, 1024, 1024))
    t = time.perf_counter()
    got = reference_pyramid(f3d, oracle, dev, ref, f0, f1, dict(pkg.DEFAULT_PARAMS))
    seconds = time.perf_counter() - t
    assert digest(got) == committed("c5_1024_default_sha256")
    line = f"BASELINE config 5 on the reference's kernels: {seconds:.1f} s = {1024 ** 3 / seconds / 1e6:.1f} Mvoxels/s"
    print("\n" + line)
    if os.environ.get("F3D_OUT"):
        with open(os.path.join(os.environ["F3D_OUT"], "reference_kernels_vs_product.txt"), "a") as f:
            f.write(line + "\n")


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes_against_the_reference_kernels(f3d, rig, seed):
    """Differential run without the oracle in between: random box sizes (widths around the 64-lane tile edges included), random
    container slack, random spacings -- the product's launches (separate and fused, frames and frame derivatives) against the
    reference's kernels for the solver, and the median, the registration and the resampling on the same boxes."""
    rng = np.random.default_rng(1000 + seed)
    W = int(rng.choice([rng.integers(4, 40), rng.integers(60, 70), rng.integers(120, 135), rng.integers(180, 200)]))
    H, D = int(rng.integers(4, 60)), int(rng.integers(4, 36))
    dims = (W, H, D)
    cdims = (W + int(rng.integers(0, 9)), H + int(rng.integers(0, 5)), D + int(rng.integers(0, 3)))
    h = tuple(float(x) for x in rng.uniform(0.6, 8.0, size=3).astype(np.float32))
    eps_s, eps_d, alpha = 0.001, 0.0015, float(np.float32(rng.uniform(3, 12)))
    arrs = solver_inputs(rng, dims, cdims)
    dev, ref = rig(cdims)
    hip = f3d.hip()
    ptr = [dev.put(a) for a in arrs]
    # reference: phi/ksi, sweep, sweep, and the weights of the next outer iteration from the increments after ONE sweep
    r_phi, r_ksi, r_phi2, r_ksi2 = dev.out(), dev.out(), dev.out(), dev.out()
    r1, r2 = [dev.out() for _ in range(3)], [dev.out() for _ in range(3)]
    f3d.sync()
    ref.phi_ksi(*ptr, dims, h, eps_s, eps_d, r_phi, r_ksi)
    ref.solve_sweep(*ptr, r_phi, r_ksi, dims, h, alpha, *r1)
    ref.solve_sweep(*ptr[:5], *r1, r_phi, r_ksi, dims, h, alpha, *r2)
    ref.phi_ksi(*ptr[:5], *r1, dims, h, eps_s, eps_d, r_phi2, r_ksi2)
    want = {n: box(dev.get(p), dims) for n, p in zip(("phi", "ksi", "phi2", "ksi2"), (r_phi, r_ksi, r_phi2, r_ksi2))}
    for n, a, b in zip("uvw", r1, r2):
        want["1" + n], want["2" + n] = box(dev.get(a), dims), box(dev.get(b), dims)
    # product
    p_phi, p_ksi = dev.out(), dev.out()
    f3d.check(hip.f3d_phi_ksi(*ptr, W, H, D, *h, eps_s, eps_d, p_phi, p_ksi, None))
    assert bit_same(box(dev.get(p_phi), dims), want["phi"]) and bit_same(box(dev.get(p_ksi), dims), want["ksi"]), (dims, cdims, h)
    fd = [dev.out() for _ in range(4)]
    f3d.check(hip.f3d_frame_derivatives(ptr[0], ptr[1], W, H, D, *h, *fd, None))
    pitch_ok = dev.cont.pitch % 256 == 0
    launches = [("two sweeps", lambda o: hip.f3d_solve_sweep2(*ptr, p_phi, p_ksi, W, H, D, *h, alpha, *o[:3], None), "2")]
    if pitch_ok:
        launches += [
            ("sweep + phi/ksi", lambda o: hip.f3d_solve_sweep_phi_ksi(*ptr, p_phi, p_ksi, W, H, D, *h, alpha, eps_s, eps_d, *o, None), "1"),
            ("two sweeps on derivatives", lambda o: hip.f3d_solve_sweep2_fd(*fd, *ptr[2:], p_phi, p_ksi, W, H, D, *h, alpha, *o[:3], None), "2"),
            ("sweep + phi/ksi on derivatives",
             lambda o: hip.f3d_solve_sweep_phi_ksi_fd(*fd, *ptr[2:], p_phi, p_ksi, W, H, D, *h, alpha, eps_s, eps_d, *o, None), "1")]
    for label, launch, which in launches:
        outs = [dev.out() for _ in range(5)]
        f3d.check(launch(outs))
        for n, o in zip("uvw", outs):
            assert bit_same(box(dev.get(o), dims), want[which + n]), f"{label}: d{n} on {dims} in {cdims}, h = {h}"
        if which == "1":
            assert bit_same(box(dev.get(outs[3]), dims), want["phi2"]) and bit_same(box(dev.get(outs[4]), dims), want["ksi2"]), label
    # median, registration, resampling of one of the volumes
    r = int(rng.choice([3, 5, 7]))
    if min(dims) > r // 2:
        src = dev.put(box_in_container(rng, dims, cdims, -2, 2))
        a, b = dev.out(), dev.out()
        f3d.sync()
        ref.median(src, dims, r, a)
        f3d.check(hip.f3d_median(src, W, H, D, r, b, None))
        assert same(box(dev.get(b), dims), box(dev.get(a), dims)), f"median {r} on {dims}"
    a, b = dev.out(), dev.out()
    f3d.sync()
    ref.warp(*ptr[:5], dims, h, a)
    f3d.check(hip.f3d_warp(*ptr[:5], W, H, D, *h, b, None))
    assert bit_same(box(dev.get(b), dims), box(dev.get(a), dims)), f"registration on {dims}"
    dst = tuple(int(np.clip(round(n * s), 2, c)) for n, s, c in zip(dims, rng.uniform(0.45, 1.3, size=3), cdims))
    a, b, t1, t2 = dev.out(), dev.out(), dev.out(), dev.out()
    f3d.sync()
    ref.resample(ptr[0], a, t1, dims, dst)
    op = f3d.Operation("resample")
    assert op.initialize(dev.cont)
    op.execute(dev_input=ptr[0], dev_output=b, dev_temp=t2, data_size=dims, resample_size=dst)
    assert bit_same(box(dev.get(b), dst), box(dev.get(a), dst)), f"resampling {dims} -> {dst}"
    op.destroy()
