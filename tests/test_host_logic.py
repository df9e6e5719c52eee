"""Host-side logic of the product against the oracle and, where it was built, against the reference's own CUDA-free
sources compiled in place (oracle/_ref): level schedule, Gaussian taps, RAW/VTK I/O, the synthetic generator and the
z-slab plan.  No GPU needed."""
import os

import numpy as np
import pytest

from conftest import bit_same, flush_c_stdio

DIMS = [(128, 128, 128), (584, 388, 5), (512, 512, 512), (1024, 1024, 1024), (450, 180, 450), (40, 36, 32), (7, 9, 11),
        (4, 4, 4), (3, 10, 10), (100, 4, 100), (17, 1000, 33)]
FACTORS = [0.95, 0.9, 0.5, 0.99, 0.8]


@pytest.fixture(scope="module")
def ref(oracle):
    r = oracle.ref()
    if r is None:
        pytest.skip("oracle/_ref was not built (no reference tree on this machine)")
    return r


@pytest.mark.parametrize("sf", FACTORS)
def test_level_schedule_product_vs_oracle(f3d, oracle, sf):
    for dims in DIMS:
        n = f3d.max_warp_level(*dims, sf)
        assert n == oracle.max_warp_level(*dims, sf), (dims, sf)
        for level in {0, 1, n // 2, max(n - 1, 0)}:
            assert f3d.level_geometry(*dims, sf, level) == oracle.level_geometry(*dims, sf, level)


@pytest.mark.parametrize("sf", FACTORS)
def test_level_schedule_vs_reference_source(f3d, oracle, ref, sf):
    """optical_flow_base.cpp compiled verbatim: GetMaxWarpLevel is the pinned part of the oracle."""
    for dims in DIMS:
        expected = ref.ref_max_warp_level(*dims, sf)
        assert oracle.max_warp_level(*dims, sf) == expected, (dims, sf)
        assert f3d.max_warp_level(*dims, sf) == expected, (dims, sf)


def test_survey_level_counts(f3d):
    # 74 / 10 / 101 / 114 possible levels for the four benchmark volumes at 0.95
    assert [f3d.max_warp_level(*d, 0.95) for d in DIMS[:4]] == [74, 10, 101, 114]
    assert f3d.level_geometry(128, 128, 128, 0.95, 39)[0] == (18, 18, 18)
    assert f3d.level_geometry(584, 388, 5, 0.95, 9)[0] == (369, 245, 4)


def test_parameter_bag_semantics_of_the_reference(ref):
    assert ref.ref_params_first_push_wins() == 1


@pytest.mark.parametrize("sigma", [0.5, 1.0, 2.0, 3.5, 5.0, 8.0])
def test_gaussian_taps(f3d, oracle, sigma):
    r_p, t_p = f3d.gaussian_taps(sigma)
    r_o, t_o = oracle.gaussian_taps(sigma)
    assert r_p == r_o == int(3 * sigma)
    assert bit_same(t_p, t_o)
    assert abs(float(t_p.sum()) - 1.0) < 1e-6 and np.allclose(t_p, t_p[::-1])


def test_raw_io_roundtrip_and_reference_reader(f3d, oracle, tmp_path):
    rng = np.random.default_rng(3)
    vol = rng.uniform(-20, 300, size=(5, 7, 9)).astype(np.float32)
    p32 = str(tmp_path / "v-9-7-5.raw")
    f3d.write_raw(p32, vol)
    assert np.array_equal(f3d.read_raw(p32, (9, 7, 5), u8=False), vol)
    assert os.path.getsize(p32) == vol.size * 4
    p8 = str(tmp_path / "v8.raw")
    f3d.write_raw(p8, vol, u8=True)
    back = f3d.read_raw(p8, (9, 7, 5), u8=True)
    assert np.array_equal(back, np.clip(vol, 0, 255).astype(np.uint8).astype(np.float32))  # clamp, truncate
    with pytest.raises(f3d.F3dError):
        f3d.read_raw(p8, (9, 7, 4), u8=True)  # one plane too many in the file: "wrong dimensions"
    with pytest.raises(f3d.F3dError):
        f3d.read_raw(p8, (9, 7, 6), u8=True)  # file too short
    r = oracle.ref()
    if r is not None and r.ref_have_data3d():
        import ctypes as C
        fp = C.POINTER(C.c_float)
        out = np.empty_like(vol)
        assert r.ref_read_raw_u8(p8.encode(), 9, 7, 5, out.ctypes.data_as(fp)) == 1
        assert np.array_equal(out, back)
        assert r.ref_read_raw_f32(p32.encode(), 9, 7, 5, out.ctypes.data_as(fp)) == 1
        assert np.array_equal(out, vol)
        # writers: byte-identical files
        q8, q32, qv, pv = (str(tmp_path / n) for n in ("r8.raw", "r32.raw", "r.vtk", "p.vtk"))
        assert r.ref_write_raw(q8.encode(), vol.ctypes.data_as(fp), 9, 7, 5, 1)
        assert r.ref_write_raw(q32.encode(), vol.ctypes.data_as(fp), 9, 7, 5, 0)
        assert open(q8, "rb").read() == open(p8, "rb").read()
        assert open(q32, "rb").read() == open(p32, "rb").read()
        u, v, w = vol, vol * 2, vol - 1
        f3d.write_vtk(pv, u, v, w)
        assert r.ref_write_vtk(qv.encode(), u.ctypes.data_as(fp), v.ctypes.data_as(fp), w.ctypes.data_as(fp), 9, 7, 5)
        assert open(qv, "rb").read() == open(pv, "rb").read()


def test_reference_data_files_read_like_the_fixtures(f3d):
    path = "/root/reference/data/frame_0_128-128-128.raw"
    if not os.path.exists(path):
        pytest.skip("reference data not on this machine")
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "inputs_128.npz"))
    assert np.array_equal(f3d.read_raw(path, (128, 128, 128), u8=True), gold["frame_0"].astype(np.float32))


def test_synthetic_pair_is_deterministic_and_slabwise(f3d):
    a0, a1 = f3d.synth_pair(40, 36, 24)
    b0, b1 = f3d.synth_pair(40, 36, 24)
    assert np.array_equal(a0, b0) and np.array_equal(a1, b1)
    assert float(a0.max()) == 255.0 and float(a0.min()) >= 0.0
    f0 = np.zeros((24, 36, 40), np.float32)
    f1 = np.zeros_like(f0)
    m = max(f3d.synth_planes(40, 36, 24, lo, hi, f0, f1) for lo, hi in ((0, 7), (7, 19), (19, 24)))
    s = np.float32(255.0) / np.float32(m)
    assert np.array_equal(f0 * s, a0) and np.array_equal(f1 * s, a1)
    # frame_1 is frame_0 moved by (+2, -1, +0.5): the centre of mass moves by about that much
    def com(v):
        z, y, x = np.indices(v.shape)
        t = v.sum(dtype=np.float64)
        return np.array([(x * v).sum(dtype=np.float64) / t, (y * v).sum(dtype=np.float64) / t, (z * v).sum(dtype=np.float64) / t])
    assert np.allclose(com(a1) - com(a0), [2.0, -1.0, 0.5], atol=0.1)


@pytest.mark.parametrize("depth,n", [(512, 8), (70, 8), (18, 8), (5, 8), (4, 3), (129, 2), (40, 1)])
def test_slab_partition_covers_the_volume(f3d, depth, n):
    edges = [f3d.plan_owned(depth, r, n) for r in range(n)]
    assert edges[0][0] == 0 and edges[-1][1] == depth
    for (a, b), (c, d) in zip(edges, edges[1:]):
        assert b == c and a <= b
    sizes = [b - a for a, b in edges]
    assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("depth,n,need", [(512, 8, 6), (70, 8, 6), (18, 8, 6), (5, 8, 2), (40, 3, 16), (129, 2, 9)])
def test_halo_plan_is_consistent_and_complete(f3d, depth, n, need):
    plans = [f3d.plan_exchange(depth, r, n, need, need) for r in range(n)]
    for r in range(n):
        lo, hi = f3d.plan_owned(depth, r, n)
        got = set()
        for peer, send, recv in plans[r]:
            # what I receive from the peer is exactly what the peer says it sends to me, and it owns those planes
            back = [t for t in plans[peer] if t[0] == r]
            assert len(back) == 1 and back[0][1] == recv and back[0][2] == send
            plo, phi = f3d.plan_owned(depth, peer, n)
            assert recv[0] == recv[1] or (plo <= recv[0] and recv[1] <= phi)
            assert send[0] == send[1] or (lo <= send[0] and send[1] <= hi)
            got |= set(range(*recv))
        wanted = set(range(max(0, lo - need), lo)) | set(range(hi, min(depth, hi + need)))
        assert got == wanted, (r, sorted(wanted - got), sorted(got - wanted))


def test_resample_source_planes_match_the_kernel_rule(f3d):
    for din, dout in [(512, 70), (70, 74), (5, 4), (128, 122), (19, 19)]:
        delta = np.float32(din) / np.float32(dout)
        for lo, hi in [(0, dout), (3, min(9, dout)), (dout - 1, dout), (2, 2)]:
            got = f3d.plan_resample_source(din, dout, lo, hi)
            if lo == hi:
                assert got[0] == got[1]
                continue
            left = int(np.floor(np.float32(lo) * delta))
            right = int(min(np.float32(din), np.ceil(np.float32(hi) * delta)))
            assert got == (left, right)


def test_uninitialised_operator_reports_and_returns(f3d, capfd):
    op = f3d.Operation("median")
    assert op.name == "CUDA Median"
    op.execute(dev_input=1, dev_output=2, data_size=(4, 4, 4), radius=5)
    flush_c_stdio()
    assert "was not initialized" in capfd.readouterr().out
    assert not op.initialize(None)  # Initialize(nullptr): "Initialization parameters are missing."
    flush_c_stdio()
    assert "Initialization parameters are missing" in capfd.readouterr().out
    op.destroy()
    for name, shown in (("add", "CUDA Add"), ("convolution", "CUDA Convolution 3D"), ("registration", "CUDA Registration"),
                        ("resample", "CUDA Resample"), ("solve", "CUDA Solve")):
        o = f3d.Operation(name)
        assert o.name == shown
        o.destroy()
    with pytest.raises(f3d.F3dError):
        f3d.Operation("fft")


def test_uniform_divisor_identity():
    """The arithmetic fact the solver kernels' UDiv shortcut rests on (f3d_solve.hip): for binary32 x and d,
    (float)((double)x * RN64(1 / (double)d)) == x / d whenever the quotient is in the normal range.  Random significands over
    200 binades and numerators placed next to every kind of rounding boundary of the quotient."""
    rng = np.random.default_rng(0)
    bad = 0
    for trial in range(60):
        d = np.float32(rng.uniform(1, 64)) if trial % 3 else np.float32(rng.choice([2, 4, 3, 6, 14.2, 6.4, 2.0006, 63.99999]))
        r = np.float64(1.0) / np.float64(d)
        m = rng.integers(1 << 23, 1 << 24, size=100000).astype(np.float64)
        e = rng.integers(-123, 77, size=100000)
        x = (m * np.exp2(e.astype(np.float64))).astype(np.float32) * rng.choice([-1, 1], size=100000).astype(np.float32)
        mid = ((rng.integers(1 << 24, 1 << 25, size=100000) | 1).astype(np.float64)) * np.exp2(rng.integers(-60, 60, size=100000) - 24.0)
        near = (mid * np.float64(d)).astype(np.float32)           # numerators whose quotient sits next to a midpoint
        for xs in (x, near, np.nextafter(near, np.float32(np.inf)), np.nextafter(near, np.float32(-np.inf))):
            q_ref = (xs / d).astype(np.float32)
            q_fast = (xs.astype(np.float64) * r).astype(np.float32)
            bad += int(np.count_nonzero(q_ref.view(np.uint32) != q_fast.view(np.uint32)))
    assert bad == 0


def test_piecemeal_solver_plan_properties(f3d):
    """Chunk plan of the out-of-core solver (PlanSolvePiecemeal, pure arithmetic): a level that fits is one residency without
    halo; otherwise chunk + 2 * halo fills the planes the budget allows, halo = outer_per_pass * (inner + 1), forcing the
    number of outer iterations per residency is honoured, and a budget below one plane plus halos yields chunk 0."""
    f3d.host()
    def planes_budget(planes, w, h):
        pitch = (w * 4 + 255) // 256 * 256
        return 13 * (planes * pitch * h + 17 * 256 + 256)
    for (w, h, d) in [(100, 90, 80), (512, 512, 512), (2048, 2048, 2048), (37, 21, 27)]:
        for inner in (1, 2, 5):
            for outer in (1, 3, 40):
                chunk, per_pass, halo, max_planes, _ = f3d.plan_solve_piecemeal(planes_budget(d, w, h), w, h, d, inner, outer)
                assert (chunk, per_pass, halo) == (d, outer, 0) and max_planes >= d
                for planes in (2 * (inner + 1) + 1, 3 * (inner + 1) + 4, d // 2 + 2 * (inner + 1), d - 1):
                    if planes >= d:
                        continue
                    chunk, per_pass, halo, max_planes, overlapped = f3d.plan_solve_piecemeal(planes_budget(planes, w, h), w, h, d, inner, outer)
                    assert max_planes == planes and not overlapped
                    assert 1 <= per_pass <= outer and halo == per_pass * (inner + 1) and chunk == planes - 2 * halo >= 1
                    forced = f3d.plan_solve_piecemeal(planes_budget(planes, w, h), w, h, d, inner, outer, 1)
                    assert forced[1] == 1 and forced[2] == inner + 1 and forced[0] == planes - 2 * (inner + 1)
                too_small = f3d.plan_solve_piecemeal(planes_budget(2 * (inner + 1), w, h), w, h, d, inner, outer)
                assert too_small[0] == 0 or 2 * (inner + 1) >= d
    # the cost model: a large level on a 270 GB budget keeps many outer iterations per residency, a tiny budget few
    big = f3d.plan_solve_piecemeal(270 << 30, 2048, 2048, 2048, 5, 40)
    small = f3d.plan_solve_piecemeal(planes_budget(40, 2048, 2048), 2048, 2048, 2048, 5, 40)
    assert big[1] >= 8 and small[1] <= 2
    # two chunk sets hold the eight fields that travel twice and the five compute-only ones once: 13 / 21 of the planes per field; the
    # model takes them when chunks stay much thicker than their halos
    auto_big = f3d.plan_solve_piecemeal(270 << 30, 2048, 2048, 2048, 5, 40, 0, -1)
    forced_on = f3d.plan_solve_piecemeal(270 << 30, 2048, 2048, 2048, 5, 40, 0, 1)
    assert forced_on[4] == 1 and big[3] // 2 < forced_on[3] <= big[3] * 13 // 21 + 1 and forced_on[0] == forced_on[3] - 2 * forced_on[2]
    assert auto_big[:4] in (big[:4], forced_on[:4])
    # (a budget of 40 planes: two sets of 24 with a one-iteration halo, or one of 40 -- either way a plan that fits)
    auto_small = f3d.plan_solve_piecemeal(planes_budget(40, 2048, 2048), 2048, 2048, 2048, 5, 40, 0, -1)
    assert auto_small[0] >= 1 and auto_small[0] == auto_small[3] - 2 * auto_small[2] and (auto_small[3] == 40 or 20 <= auto_small[3] <= 13 * 40 // 21)


@pytest.mark.parametrize("planes,forced,outer,inner", [(20, 1, 3, 5), (28, 2, 5, 5), (17, 0, 4, 3)])
def test_piecemeal_solver_windows_reproduce_the_unsplit_solve(f3d, oracle, planes, forced, outer, inner):
    """The out-of-core solver's residency scheme on the CPU, with the ORACLE as the compute and the PRODUCT's plan
    (PlanSolvePiecemeal) deciding chunk, halo and outer iterations per pass: every chunk is staged with its halo in a
    container of its own, outer iteration j of a pass runs phi/ksi on the chunk widened by (n-1-j)(K+1) + K planes and sweep s
    on (n-1-j)(K+1) + K-1-s, only the owned planes go back.  Bit-identical to the unsplit oracle loop."""
    f3d.host()
    W, H, D = 19, 12, 31
    K = inner
    pitch = (W * 4 + 255) // 256 * 256
    budget = 13 * (planes * pitch * H + 17 * 256 + 256)
    chunk, n_pass, halo, max_planes, _ = f3d.plan_solve_piecemeal(budget, W, H, D, K, outer, forced)
    assert max_planes == planes and chunk == planes - 2 * halo and halo == n_pass * (K + 1)
    rng = np.random.default_rng(planes)
    dims, h = (W, H, D), (1.1, 0.9, 1.4)
    full = [rng.uniform(lo, hi, size=(D, H, W)).astype(np.float32) for lo, hi in [(0, 255), (0, 255), (-2, 2), (-2, 2), (-2, 2)]]
    # unsplit
    du, dv, dw = (np.zeros((D, H, W), np.float32) for _ in range(3))
    for _ in range(outer):
        phi, ksi = oracle.phi_ksi(*full, du, dv, dw, dims, h, 0.001, 0.001)
        for _ in range(K):
            du, dv, dw = oracle.solve_sweep(*full, du, dv, dw, phi, ksi, dims, h, 7.5)
    expect = (du, dv, dw)
    # chunked: host increments inc -> nxt per pass, like flow_du / temp_du
    inc = [np.zeros((D, H, W), np.float32) for _ in range(3)]
    nxt = [np.full((D, H, W), np.nan, np.float32) for _ in range(3)]
    it = 0
    while it < outer:
        n = min(n_pass, outer - it)
        reach = n * (K + 1)
        for z0 in range(0, D, chunk):
            z1 = min(D, z0 + chunk)
            base = z0 - halo
            lo, hi = max(0, z0 - reach), min(D, z1 + reach)
            stage = lambda vol: _staged(vol, base, planes, lo, hi)
            fixed = [stage(v) for v in full]
            d = [stage(v) for v in inc]
            tmp = [np.full_like(fixed[0], np.nan) for _ in range(3)]
            win = lambda grow: oracle.Geom(H, W, base, max(0, z0 - grow), min(D, z1 + grow))
            for j in range(n):
                g = (n - 1 - j) * (K + 1)
                phi, ksi = oracle.phi_ksi(*fixed, *d, dims, h, 0.001, 0.001, g=win(g + K))
                for s in range(K):
                    oracle.solve_sweep(*fixed, *d, phi, ksi, dims, h, 7.5, g=win(g + K - 1 - s), out=tuple(tmp))
                    d, tmp = tmp, d
            for k in range(3):
                nxt[k][z0:z1] = d[k][z0 - base:z1 - base]
        inc, nxt = nxt, inc
        it += n
    for got, e, name in zip(inc, expect, ("du", "dv", "dw")):
        assert bit_same(got, e), name


def _staged(vol, base, planes, lo, hi):
    """planes [lo, hi) of a volume in a NaN-poisoned container whose plane 0 holds global plane `base`"""
    c = np.full((planes,) + vol.shape[1:], np.nan, np.float32)
    c[lo - base:hi - base] = vol[lo:hi]
    return c


def test_generated_median_networks_are_current():
    """csrc/f3d_median_nets.h is what tools/gen_median_nets.py writes (the generator checks every network against sorted()
    on random inputs with ties before it emits it), so the committed header cannot drift from its checked source."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_median_nets", os.path.join(root, "tools", "gen_median_nets.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text, counts = gen.main()
    assert counts == [238, 244]
    with open(gen.HEADER) as f:
        assert f.read() == text
