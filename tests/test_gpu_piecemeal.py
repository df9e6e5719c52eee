"""Out-of-core ("piecemeal") path on the MI355X: the five host-volume operators and the OpticalFlowP driver, with the device
budget forced small (F3D_P_BUDGET_MB) so that every operator really cuts its level into several z-chunks.  Every result
must equal the oracle's whole-volume result bit for bit (sign of zero aside): chunking, halos, the in-place resample
order and the number of outer iterations per residency must not be visible in the output."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import box_in_container, same

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _env():
    keep = {k: os.environ.get(k) for k in ("F3D_P_BUDGET_MB", "F3D_P_OUTER_PER_PASS", "F3D_P_PIN", "F3D_P_OVERLAP", "F3D_P_RESIDENT")}
    yield
    for k, v in keep.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def budget_for(planes_total, w, h, buffers):
    """MB that let `buffers` buffers hold `planes_total` planes of a w x h level in total (mirrors ChunkBox::TotalPlanes)."""
    pitch = (w * 4 + 255) // 256 * 256
    return (planes_total * pitch * h + buffers * (17 * 256 + 256) + 1024) / (1024.0 * 1024.0)


def set_budget(mb):
    os.environ["F3D_P_BUDGET_MB"] = repr(float(mb))


def make_op(f3d, name):
    f3d.check(f3d.hip().f3d_init(-1), "f3d_init")
    op = f3d.Operation(name)
    assert op.initialize(None), name
    return op


def test_operator_names(f3d):
    names = {"add_p": "CUDA Add Piecemeal", "resample_p": "CUDA Resample Piecemeal", "registration_p": "CUDA Registration",
             "solve_p": "CUDA Sove Piecemeal", "stat_p": "CUDA Stat Piecemeal"}
    for key, name in names.items():
        op = make_op(f3d, key)
        assert op.name == name
        op.destroy()


def test_add_and_stat_in_chunks(f3d, oracle):
    rng = np.random.default_rng(3)
    W, H, D = 45, 22, 31
    cd = (64, 32, 40)
    a = box_in_container(rng, (W, H, D), cd, poison=False)
    b = box_in_container(rng, (W, H, D), cd, poison=False)
    c = box_in_container(rng, (W, H, D), cd, poison=False)
    expect = a.copy()
    expect[:D, :H, :W] += b[:D, :H, :W]
    set_budget(budget_for(2 * 7, W, H, 2))  # 7 planes per chunk -> 5 chunks
    va, vb, vc = (f3d.HostVolume(x) for x in (a, b, c))
    op = make_op(f3d, "add_p")
    op.execute(operand_0=va, operand_1=vb, data_size=(W, H, D))
    assert same(va.array, expect)
    op.destroy()

    set_budget(budget_for(3 * 4, W, H, 3))
    st = f3d.Stat3()
    op = make_op(f3d, "stat_p")
    op.execute(flow_u=va, flow_v=vb, flow_w=vc, data_size=(W, H, D), stat=st)
    mn, mx, avg, total = oracle.flow_stats(va.array, vb.array, vc.array, (W, H, D))
    assert st.min == mn and st.max == mx
    assert abs(st.avg - total / (W * H * D)) <= 1e-6 * abs(avg)
    op.destroy()
    for v in (va, vb, vc):
        v.destroy()


@pytest.mark.parametrize("in_dims,out_dims,planes", [
    ((50, 33, 23), (48, 32, 22), 20),     # one level down, several chunks
    ((50, 33, 23), (7, 5, 4), 40),        # frames from the original size: long source spans
    ((19, 14, 11), (20, 15, 12), 12),     # flow upsampling
    ((30, 20, 16), (30, 20, 16), 10),     # identity sizes
])
def test_resample_matches_oracle(f3d, oracle, in_dims, out_dims, planes):
    rng = np.random.default_rng(5)
    cd = (64, 40, 32)
    src = box_in_container(rng, in_dims, cd, lo=-2, hi=2, poison=True)
    expect = oracle.resample(src, in_dims, out_dims)
    ow, oh, od = out_dims
    wc, hc = max(in_dims[0], ow), max(in_dims[1], oh)
    op = make_op(f3d, "resample_p")
    # separate output
    set_budget(budget_for(planes, wc, hc, 4))
    vin, vout = f3d.HostVolume(src.copy()), f3d.HostVolume(np.full(src.shape, np.nan, np.float32))
    op.execute(input=vin, output=vout, data_size=in_dims, resample_size=out_dims)
    assert same(vout.array[:od, :oh, :ow], expect[:od, :oh, :ow])
    assert same(vin.array[:in_dims[2], :in_dims[1], :in_dims[0]], src[:in_dims[2], :in_dims[1], :in_dims[0]])
    # in place, like the driver resamples the flow
    vio = f3d.HostVolume(src.copy())
    op.execute(input=vio, output=vio, data_size=in_dims, resample_size=out_dims)
    assert same(vio.array[:od, :oh, :ow], expect[:od, :oh, :ow])
    # page-locked volumes: two buffer sets, the upload of the next chunk beside the kernels and the download of this one (three chunks
    # and more; the same planes in total now make two sets of half the size).  Separate output and in place; F3D_P_OVERLAP=0 = in order.
    for planes2, overlap in ((planes, None), (2 * planes, None), (2 * planes, "0")):
        set_budget(budget_for(planes2, wc, hc, 8))
        if overlap is not None:
            os.environ["F3D_P_OVERLAP"] = overlap
        try:
            pin, pout = f3d.HostVolume(src.copy(), pin=True), f3d.HostVolume(np.full(src.shape, np.nan, np.float32), pin=True)
            op.execute(input=pin, output=pout, data_size=in_dims, resample_size=out_dims)
            assert same(pout.array[:od, :oh, :ow], expect[:od, :oh, :ow]), (planes2, overlap)
            assert same(pin.array[:in_dims[2], :in_dims[1], :in_dims[0]], src[:in_dims[2], :in_dims[1], :in_dims[0]])
            pio = f3d.HostVolume(src.copy(), pin=True)
            op.execute(input=pio, output=pio, data_size=in_dims, resample_size=out_dims)
            assert same(pio.array[:od, :oh, :ow], expect[:od, :oh, :ow]), (planes2, overlap, "in place")
        finally:
            os.environ.pop("F3D_P_OVERLAP", None)
            for v in (pin, pout, pio):
                v.destroy()
    op.destroy()
    for v in (vin, vout, vio):
        v.destroy()


@pytest.mark.parametrize("w_amp,planes", [(0.4, 60), (3.0, 60), (9.0, 120), (40.0, 6 * 26)])
def test_registration_matches_oracle(f3d, oracle, w_amp, planes):
    """Flows of up to w_amp planes in z: the chunk halo follows max |w| of each chunk; large flows force sub-chunks."""
    rng = np.random.default_rng(int(w_amp * 10))
    W, H, D = 37, 21, 26
    cd = (48, 24, 32)
    h = (1.3, 0.9, 1.1)
    f0 = box_in_container(rng, (W, H, D), cd, 0, 255)
    f1 = box_in_container(rng, (W, H, D), cd, 0, 255)
    u = box_in_container(rng, (W, H, D), cd, -6, 6)
    v = box_in_container(rng, (W, H, D), cd, -6, 6)
    w = box_in_container(rng, (W, H, D), cd, -w_amp, w_amp)
    w[3, 4, 5] = np.nan          # NaN flow -> frame_0 (registration_3d.cu:60-64)
    w[10:14] *= 0.1              # a quiet stretch: its chunks need a thinner halo
    expect = oracle.warp(f0, f1, u, v, w, (W, H, D), h)
    set_budget(budget_for(planes, W, H, 6))
    vols = [f3d.HostVolume(x.copy()) for x in (f0, f1, u, v, w)]
    temp = f3d.HostVolume(np.full(f0.shape, np.nan, np.float32))
    op = make_op(f3d, "registration_p")
    op.execute(frame_0=vols[0], frame_1=vols[1], flow_u=vols[2], flow_v=vols[3], flow_w=vols[4], temp=temp,
               hx=h[0], hy=h[1], hz=h[2], data_size=(W, H, D), max_mag=0)
    # the warped frame is now frame_1's storage, the old frame_1 is in temp (Data3D::Swap)
    assert same(vols[1].array[:D, :H, :W], expect[:D, :H, :W])
    assert same(temp.array[:D, :H, :W], f1[:D, :H, :W])
    op.destroy()
    for x in vols + [temp]:
        x.destroy()


def test_swapped_pair_survives_the_destruction_of_one_volume(f3d):
    """registration_p swaps the storage of frame_1 and temp (Data3D::Swap).  Destroying one of the two afterwards must
    release the array THAT volume holds now, not the one it was created with: the survivor keeps its memory, its
    page-lock and its .array; and the storage table does not leak when both are gone."""
    rng = np.random.default_rng(5)
    W, H, D = 24, 12, 10
    mk = lambda lo, hi: rng.uniform(lo, hi, size=(D, H, W)).astype(np.float32)
    f0, f1, u, v, w = mk(0, 255), mk(0, 255), mk(-1, 1), mk(-1, 1), mk(-1, 1)
    set_budget(budget_for(1000, W, H, 6))
    before = len(f3d.HostVolume._storage)
    vols = [f3d.HostVolume(x.copy(), pin=(i == 1)) for i, x in enumerate((f0, f1, u, v, w))]   # frame_1 page-locked
    temp = f3d.HostVolume(np.full(f0.shape, np.nan, np.float32))
    f1_addr = vols[1].array.ctypes.data
    op = make_op(f3d, "registration_p")
    op.execute(frame_0=vols[0], frame_1=vols[1], flow_u=vols[2], flow_v=vols[3], flow_w=vols[4], temp=temp,
               hx=1.0, hy=1.0, hz=1.0, data_size=(W, H, D), max_mag=0)
    assert temp.array.ctypes.data == f1_addr          # temp now holds frame_1's original (page-locked) array
    warped = vols[1].array.copy()
    vols[1].destroy()                                  # drops the array it holds NOW (temp's original)
    assert same(temp.array, f1)                        # the survivor still reads its data ...
    yes = C.c_int()
    f3d.check(f3d.hip().f3d_host_is_pinned(C.c_void_p(f1_addr), C.byref(yes)))
    assert yes.value == 1                              # ... and keeps its page-lock
    temp.array[0, 0, 0] = 7.0                          # the memory is alive and writable
    assert np.isfinite(warped).all()
    op.destroy()
    for x in [vols[0]] + vols[2:] + [temp]:
        x.destroy()
    f3d.check(f3d.hip().f3d_host_is_pinned(C.c_void_p(f1_addr), C.byref(yes)))
    assert yes.value == 0
    assert len(f3d.HostVolume._storage) == before
    vols[1].destroy()                                  # idempotent


def oracle_solve(oracle, f0, f1, u, v, w, dims, h, outer, inner, alpha, eps_s, eps_d):
    du, dv, dw = (np.zeros_like(f0) for _ in range(3))
    for _ in range(outer):
        phi, ksi = oracle.phi_ksi(f0, f1, u, v, w, du, dv, dw, dims, h, eps_s, eps_d)
        for _ in range(inner):
            du, dv, dw = oracle.solve_sweep(f0, f1, u, v, w, du, dv, dw, phi, ksi, dims, h, alpha)
    return du, dv, dw


@pytest.mark.parametrize("planes,forced,outer,inner,overlap,D", [
    (1000, 0, 3, 5, 0, 27),   # fits: one residency
    (22, 1, 3, 5, 0, 27),     # 10-plane chunks, increments go home after every outer iteration
    (34, 2, 5, 5, 0, 40),     # two outer iterations per residency, last pass shorter
    (30, 0, 4, 3, 0, 33),     # planner's own choice, odd inner count
    (26, 3, 3, 2, 0, 27),     # three per residency with pairs only
    (13, 1, 4, 5, 1, 27),     # two chunk sets of 13 planes: one-plane chunks, 27 of them through the pipeline per pass
    (28, 2, 5, 5, 1, 61),     # overlapped, two outer iterations per residency, odd number of chunks per pass
    (26, 0, 3, 5, 0, 27),     # 26 of 27 planes: almost fits
    (1000, 0, 3, 5, 1, 27),   # overlap asked for but the level fits: still one residency
    (13, 1, 4, 5, 2, 27),     # overlap asked for on pageable volumes: the solver keeps the serial schedule
])
@pytest.mark.parametrize("constants", ["0", "1"])
def test_solve_matches_oracle(f3d, oracle, planes, forced, outer, inner, overlap, D, constants, monkeypatch):
    # constants "0": every field in the chunk sets (the plan geometry below is asserted for that layout); "1": the two frames and
    # u, v, w held on the device for the whole level wherever the budget allows it (same bits, other chunks)
    monkeypatch.setenv("F3D_P_CONSTANTS", constants)
    rng = np.random.default_rng(planes)
    W, H = 37, 21
    cd = (40, 24, D + 3)
    dims, h = (W, H, D), (1.25, 1.0, 1.6)
    f0 = box_in_container(rng, dims, cd, 0, 255)
    f1 = np.full_like(f0, np.nan)
    f1[:D, :H, :W] = f0[:D, :H, :W] + rng.uniform(-8, 8, size=(D, H, W)).astype(np.float32)
    u, v, w = (box_in_container(rng, dims, cd, -2, 2) for _ in range(3))
    expect = oracle_solve(oracle, f0, f1, u, v, w, dims, h, outer, inner, 7.5, 0.001, 0.001)
    # 13 fields per chunk set, 15 with the second weight pair of the fused last sweep (odd inner count, another outer iteration to
    # fuse with inside the residency -- forced == 1 leaves none).  By default the operator fuses only where the level fits in one
    # residency; F3D_P_FUSED=1 makes it fuse inside the residencies of chunked levels too, which is what this test wants to see.
    os.environ["F3D_P_FUSED"] = "1"
    want_fused = inner % 2 == 1 and outer > 1 and forced != 1
    per_set = 15 if want_fused else 13
    # (two chunk sets hold the eight fields that travel twice, the compute-only ones once)
    fields = per_set + 8 if overlap and planes < D else per_set
    # ... and keep three staging buffers of 2 x halo planes for the increments neighbouring chunks share
    staging = 6 * forced * (inner + 1) if overlap and planes < D else 0
    set_budget(budget_for(fields * planes + staging, W, H, fields + (3 if staging else 0)))
    os.environ["F3D_P_OUTER_PER_PASS"] = str(forced)
    os.environ["F3D_P_OVERLAP"] = str(min(overlap, 1))
    pinned = overlap == 1
    names = ["frame_0", "frame_1", "flow_u", "flow_v", "flow_w", "flow_du", "flow_dv", "flow_dw", "temp_du", "temp_dv", "temp_dw"]
    arrays = [f0, f1, u, v, w] + [np.full(f0.shape, np.nan, np.float32) for _ in range(6)]
    vols = {n: f3d.HostVolume(a.copy(), pin=pinned) for n, a in zip(names, arrays)}  # overlap needs page-locked volumes
    op = make_op(f3d, "solve_p")
    op.execute(outer_iterations_count=outer, inner_iterations_count=inner, equation_alpha=7.5, equation_smoothness=0.001,
               equation_data=0.001, hx=h[0], hy=h[1], hz=h[2], data_size=dims, **vols)
    chunk, per_pass, halo, passes, overlapped = op.solve_p_last()
    fused = op.solve_p_fused_weights()
    assert not fused or want_fused
    if want_fused and not fused:
        # the 15-field plan came out with one outer iteration per residency (nothing to fuse): the operator took the 13-field plan
        # of the same budget instead, i.e. 15/13 of the planes this test budgeted per field
        assert passes == -(-outer // per_pass) and halo == (0 if chunk == D else per_pass * (inner + 1))
    elif planes >= D:
        assert (chunk, per_pass, halo, passes, overlapped) == (D, outer, 0, 1, False)
    elif constants == "1":
        assert chunk < D and halo == per_pass * (inner + 1) and passes == -(-outer // per_pass)
    else:
        assert overlapped == pinned
        if pinned or not overlap:
            assert chunk == planes - 2 * halo
        else:   # (overlap asked for on pageable volumes: one set gets the whole budget)
            assert 0 <= chunk - ((fields * planes + staging) // per_set - 2 * halo) <= 1   # (+ the other buffers' alignment slack)
        assert chunk < D and halo == per_pass * (inner + 1) and passes == -(-outer // per_pass)
        if forced:
            assert per_pass == forced
    for n, e in zip(("flow_du", "flow_dv", "flow_dw"), expect):
        got = vols[n].array[:D, :H, :W]
        assert same(got, e[:D, :H, :W]), f"{n}: max diff {np.nanmax(np.abs(got - e[:D, :H, :W]))}"
    # inputs untouched
    for n, a in zip(names[:5], arrays[:5]):
        assert same(vols[n].array[:D, :H, :W], a[:D, :H, :W])
    op.destroy()
    for x in vols.values():
        x.destroy()
    del os.environ["F3D_P_FUSED"]


def test_solve_reports_low_memory(f3d, capfd):
    W, H, D = 37, 21, 27
    set_budget(budget_for(13 * 8, W, H, 13))   # 8 planes per field: a 6-plane halo on either side cannot fit
    vols = {n: f3d.HostVolume(np.zeros((D, H, W), np.float32)) for n in
            ["frame_0", "frame_1", "flow_u", "flow_v", "flow_w", "flow_du", "flow_dv", "flow_dw", "temp_du", "temp_dv", "temp_dw"]}
    op = make_op(f3d, "solve_p")
    op.execute(outer_iterations_count=2, inner_iterations_count=5, equation_alpha=7.5, equation_smoothness=0.001,
               equation_data=0.001, hx=1.0, hy=1.0, hz=1.0, data_size=(W, H, D), **vols)
    assert op.solve_p_last()[0] == 0
    op.destroy()
    for x in vols.values():
        x.destroy()


def run_p(f3d, f0, f1, resident=True, full=False, **kw):
    d, h, w = f0.shape
    flow = f3d.PiecemealOpticalFlow()
    flow.initialize(w, h, d)
    flow.set_resident(resident)
    flow.set_full_pipeline(full)
    try:
        out = flow.compute(f0, f1, silent=True, **kw)
        run_p.originals_on_device = flow.originals_on_device()
        return out, flow.stats()
    finally:
        flow.destroy()


def test_registration_inside_the_first_residency(f3d, monkeypatch):
    """Frame 1 registered inside the solver's first residency (CudaOperationSolveP::register_frame_1, the driver's default) against the
    separate registration operator (F3D_P_FUSED_WARP=0): same flow bit for bit, the caller's frames untouched -- on a pair that moves
    two planes along z (the unregistered planes arrive in several pieces) and on one that moves seven (the reach does not leave room in
    the chunk buffers of the finest levels: the solver declines there and the driver registers the classical way)."""
    W, H, D = 40, 36, 48
    f0, _ = f3d.synth_pair(W, H, D)
    kw = dict(warp_levels_count=10, outer_iterations_count=4, inner_iterations_count=5)
    for shift, planes in ((2, 26), (2, 17), (7, 26)):
        f1 = np.ascontiguousarray(np.roll(f0, shift, axis=0))
        keep0, keep1 = f0.copy(), f1.copy()
        set_budget(budget_for(13 * planes, W, H, 13))
        runs = {}
        for fused in ("1", "0"):
            monkeypatch.setenv("F3D_P_FUSED_WARP", fused)
            flow = f3d.PiecemealOpticalFlow()
            flow.initialize(W, H, D)
            flow.set_resident(False)
            try:
                runs[fused] = (flow.compute(f0, f1, silent=True, **kw), flow.levels_registered_inside(), flow.stats())
            finally:
                flow.destroy()
            assert same(f0, keep0) and same(f1, keep1), "the caller's frames must come back unchanged"
        assert runs["0"][1] == 0 and runs["1"][2][1] >= 2, (shift, planes, runs["0"][1:], runs["1"][1:])
        assert runs["1"][1] >= (runs["1"][2][1] if shift == 2 else 1), (shift, planes, runs["1"][1:])
        if shift == 7:
            assert runs["1"][1] < 10, "the deep reach was expected to make the solver decline on the finest levels"
        assert np.abs(runs["0"][0][2]).max() > 0.25, "the pair was meant to move along z"
        for a, b, n in zip(runs["1"][0], runs["0"][0], "uvw"):
            assert same(a, b), f"shift {shift}, {planes} planes per field, {n}: registration inside the solver differs"


def test_constant_fields_held_on_the_device(f3d, monkeypatch):
    """The two frames and u, v, w of a level held on the device for the whole level beside smaller chunk sets (three fields up per
    residency instead of eight; SolvePiecemealPlan::constants_on_device): pinned on, pinned off and left to the cost model, with the
    registration inside the solver and by the operator -- the same flow bit for bit, the caller's frames untouched."""
    W, H, D = 40, 36, 48
    f0, _ = f3d.synth_pair(W, H, D)
    f1 = np.ascontiguousarray(np.roll(f0, 2, axis=0))
    keep0, keep1 = f0.copy(), f1.copy()
    kw = dict(warp_levels_count=10, outer_iterations_count=4, inner_iterations_count=5)
    for planes_total in (5 * 48 + 16 * 20, 5 * 48 + 8 * 16):      # room for two chunk sets beside the constants / for one
        set_budget(budget_for(planes_total, W, H, 21))
        runs = {}
        for constants in ("0", "1", None):
            for fused_warp in ("1", "0"):
                if constants is None:
                    monkeypatch.delenv("F3D_P_CONSTANTS", raising=False)
                else:
                    monkeypatch.setenv("F3D_P_CONSTANTS", constants)
                monkeypatch.setenv("F3D_P_FUSED_WARP", fused_warp)
                flow = f3d.PiecemealOpticalFlow()
                flow.initialize(W, H, D)
                flow.set_resident(False)
                try:
                    runs[(constants, fused_warp)] = (flow.compute(f0, f1, silent=True, **kw), flow.levels_with_constants_on_device(), flow.stats())
                finally:
                    flow.destroy()
                assert same(f0, keep0) and same(f1, keep1), "the caller's frames must come back unchanged"
        ref = runs[("0", "0")]
        assert ref[1] == 0 and runs[("0", "1")][1] == 0 and ref[2][1] >= 1, (planes_total, ref[1:])
        assert runs[("1", "1")][1] >= 1 and runs[("1", "0")][1] >= 1, (planes_total, runs[("1", "1")][1:], "no level held its constants")
        for key, (got, _, _) in runs.items():
            for a, b, n in zip(got, ref[0], "uvw"):
                assert same(a, b), f"{planes_total} planes, F3D_P_CONSTANTS={key[0]}, F3D_P_FUSED_WARP={key[1]}, {n} differs"


def test_driver_matches_oracle_small(f3d, oracle):
    """Whole pyramid on 40x36x32 with a budget that streams the upper levels: equals the oracle's pipeline without blur and
    median, which is what the reference's piecemeal driver computes."""
    f0, f1 = f3d.synth_pair(40, 36, 32)
    keep0, keep1 = f0.copy(), f1.copy()
    (exp, levels) = oracle.compute_flow(f0, f1, gaussian_sigma=0.0, median_radius=1)
    set_budget(budget_for(13 * 26, 40, 36, 13))
    for resident in (False, True):
        got, (passes, streamed, on_device) = run_p(f3d, f0, f1, resident=resident)
        assert streamed >= 1 and passes > levels
        assert (on_device >= 10) if resident else (on_device == 0)
        assert resident or not run_p.originals_on_device
        assert on_device + streamed <= levels
        for g, e, n in zip(got, exp, "uvw"):
            assert same(g, e), f"resident={resident} {n}: max diff {np.abs(g - e).max()}"
        assert same(f0, keep0) and same(f1, keep1), "the caller's frames must come back unchanged"
    # a budget that holds everything: all levels on the device, the originals uploaded once for level 0
    set_budget(64.0)
    got, (passes, streamed, on_device) = run_p(f3d, f0, f1)
    assert (streamed, on_device, passes) == (0, levels, levels) and run_p.originals_on_device
    for g, e in zip(got, exp):
        assert same(g, e)
    # budgets in between: the coarsest levels beside device copies of the originals, then levels with the originals streamed,
    # then levels through the host -- wherever the boundaries fall
    seen = set()
    for planes in (14, 18, 22, 30, 40):
        set_budget(budget_for(13 * planes, 40, 36, 13))
        got, (passes, streamed, on_device) = run_p(f3d, f0, f1)
        seen.add(run_p.originals_on_device)
        assert 0 < on_device < levels
        for g, e in zip(got, exp):
            assert same(g, e), f"{planes} planes per field"
    assert seen == {False, True}, "the budgets should cover levels with and without device copies of the originals"


def test_driver_without_levels_returns_zero_flow(f3d):
    f0, f1 = f3d.synth_pair(24, 20, 16)
    got, (passes, streamed, on_device) = run_p(f3d, f0, f1, warp_levels_count=0)
    assert (passes, streamed, on_device) == (0, 0, 0)
    for g in got:
        assert not g.any()


def test_driver_matches_golden_crops(f3d):
    """The committed piecemeal fixtures (tests/golden/expected_piecemeal.npz: oracle, no blur, no median) on the crops of the
    reference's data volumes, with a budget that puts the finest levels through chunks; the thin 96x64x5 crop runs with
    chunks of single planes' worth of halo."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    e = np.load(os.path.join(gold, "expected_piecemeal.npz"))
    i128 = np.load(os.path.join(gold, "inputs_128.npz"))
    irub = np.load(os.path.join(gold, "inputs_rub.npz"))
    crop = (slice(40, 64), slice(40, 80), slice(40, 88))
    f0 = i128["frame_0"].astype(np.float32)[crop].copy()
    f1 = i128["frame_1"].astype(np.float32)[crop].copy()
    set_budget(budget_for(13 * 20, 48, 40, 13))
    got, (passes, streamed, on_device) = run_p(f3d, f0, f1)
    assert streamed >= 1 and on_device >= 1
    for g, x, n in zip(got, e["crop128_flow"], "uvw"):
        assert same(g, x), f"crop128 {n}: max diff {np.abs(g - x).max()}"
    rc = (slice(0, 5), slice(100, 164), slice(200, 296))
    r0 = np.repeat(irub["slice_0"][None], int(irub["depth"]), axis=0).astype(np.float32)[rc].copy()
    r1 = np.repeat(irub["slice_1"][None], int(irub["depth"]), axis=0).astype(np.float32)[rc].copy()
    set_budget(64.0)
    got, _ = run_p(f3d, r0, r1)
    for g, x, n in zip(got, e["croprub_flow"], "uvw"):
        assert same(g, x), f"croprub {n}: max diff {np.abs(g - x).max()}"


def test_driver_matches_resident_driver(f3d):
    """120^3 default schedule: streamed (about 1/3 of the level resident at a time) vs everything resident, no blur/median."""
    n = 120
    f0, f1 = f3d.synth_pair(n, n, n)
    flow = f3d.OpticalFlow()
    flow.initialize(n, n, n)
    try:
        exp = flow.compute(f0, f1, silent=True, gaussian_sigma=0.0, median_radius=1, outer_iterations_count=6)
    finally:
        flow.destroy()
    set_budget(budget_for(13 * 52, n, n, 13))
    got, (passes, streamed, on_device) = run_p(f3d, f0, f1, outer_iterations_count=6)
    assert streamed >= 3 and on_device >= 20
    for g, e, c in zip(got, exp, "uvw"):
        assert same(g, e), f"{c}: max diff {np.abs(g - e).max()}"
    os.environ["F3D_P_PIN"] = "0"   # staged copies give the same result
    got2, (_, _, on_device) = run_p(f3d, f0, f1, resident=False, outer_iterations_count=6)
    assert on_device == 0
    for g, e in zip(got2, exp):
        assert same(g, e)
    os.environ["F3D_P_OVERLAP"] = "1"   # copies beside the kernels; without page-locked volumes the solver stays serial
    set_budget(budget_for(26 * 40, n, n, 26))
    for pin in ("1", "0"):
        os.environ["F3D_P_PIN"] = pin
        got3, (passes, streamed, _) = run_p(f3d, f0, f1, outer_iterations_count=6)
        assert streamed >= 3
        for g, e in zip(got3, exp):
            assert same(g, e)


def test_cli_partial_mode(f3d, tmp_path):
    """bin/flow3d --partial on two frame pairs with a small --budget-mb: files end in -partial.raw and equal the python
    binding's result for the same pairs; the second pair starts from the swapped frame like the resident mode."""
    import subprocess
    W, H, D = 48, 40, 24
    f0, f1 = f3d.synth_pair(W, H, D)
    frames = [np.round(np.clip(f0, 0, 255)), np.round(np.clip(f1, 0, 255)), np.round(np.clip(0.5 * (f0 + f1), 0, 255))]
    paths = []
    for k, fr in enumerate(frames):
        p = tmp_path / f"frame{k}.raw"
        fr.astype(np.uint8).tofile(p)
        paths.append(str(p))
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-flow3d_amd", "bin", "flow3d")
    prefix = str(tmp_path / "seq")
    mb = budget_for(13 * 20, W, H, 13)
    run = subprocess.run([exe, "--dims", str(W), str(H), str(D), "--frames", *paths, "--out", prefix, "--levels", "6", "--outer", "3",
                          "--partial", "--budget-mb", repr(mb), "--stats", "--silent"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "Mode: Partial processing mode" in run.stdout and "levels streamed" in run.stdout, run.stdout
    set_budget(mb)
    for k in range(2):
        a = frames[k].astype(np.uint8).astype(np.float32)
        b = frames[k + 1].astype(np.uint8).astype(np.float32)
        exp, (_, streamed, _) = run_p(f3d, a, b, warp_levels_count=6, outer_iterations_count=3)
        assert streamed >= 1
        got = [np.fromfile(f"{prefix}_{k}_flow-{c}-{W}-{H}-{D}-partial.raw", np.float32).reshape(D, H, W) for c in "uvw"]
        for g, e in zip(got, exp):
            assert same(g, e)
    # --partial --full writes what the resident mode writes
    common = [exe, "--dims", str(W), str(H), str(D), "--frames", paths[0], paths[1], "--levels", "6", "--outer", "3", "--silent"]
    for extra, tag in ((["--partial", "--full", "--budget-mb", repr(mb)], "pf"), ([], "res")):
        run = subprocess.run(common + ["--out", str(tmp_path / tag)] + extra, capture_output=True, text=True, timeout=120)
        assert run.returncode == 0, run.stdout + run.stderr
    for c in "uvw":
        a = np.fromfile(tmp_path / f"pf_flow-{c}-{W}-{H}-{D}-partial.raw", np.float32)
        b = np.fromfile(tmp_path / f"res_flow-{c}-{W}-{H}-{D}.raw", np.float32)
        assert same(a, b), c


@pytest.mark.parametrize("sigma,planes", [(2.0, 3 * 20), (1.0, 3 * 9), (3.5, 3 * 40), (2.0, 3 * 36), (2.0, 3 * 13)])  # 36 of 37 planes: almost fits
def test_gaussian_in_chunks(f3d, oracle, sigma, planes):
    """convolution_p: rows and columns on the chunk widened by the tap radius, slices on the chunk; zero padding at the ends
    of the volume only.  Height a multiple of 4 and equal to the host volume's (the reference's precondition)."""
    rng = np.random.default_rng(int(sigma * 10))
    W, H, D = 41, 24, 37
    src = rng.uniform(0, 255, size=(D, H, W)).astype(np.float32)
    expect = oracle.gaussian(src, (W, H, D), sigma)
    set_budget(budget_for(planes, W, H, 3))
    vin, vout = f3d.HostVolume(src.copy()), f3d.HostVolume(np.full(src.shape, np.nan, np.float32))
    op = make_op(f3d, "convolution_p")
    assert op.name == "CUDA Convolution 3D Piecemeal"
    op.execute(input=vin, output=vout, data_size=(W, H, D), gaussian_sigma=sigma)
    assert same(vout.array, expect), f"max diff {np.nanmax(np.abs(vout.array - expect))}"
    assert same(vin.array, src)
    op.destroy()
    vin.destroy()
    vout.destroy()


# 51 planes in total for a depth of 26: the volume ALMOST fits (23-plane chunk + a 3-plane one whose halo ends at the volume)
@pytest.mark.parametrize("radius,planes", [(5, 2 * 9), (3, 2 * 5), (7, 2 * 12), (5, 2 * 100), (4, 2 * 8), (1, 2 * 8), (5, 51), (7, 50)])
def test_median_in_chunks_and_in_place(f3d, oracle, radius, planes):
    """median_p on a sub-box of a larger host volume: separate output, and in place (the planes a later chunk needs of what
    an earlier chunk replaced are carried over on the device)."""
    rng = np.random.default_rng(radius)
    W, H, D = 37, 21, 26
    cd = (48, 24, 30)
    src = box_in_container(rng, (W, H, D), cd, -3, 3)
    eff = radius - 1 if radius % 2 == 0 else radius
    expect = src if eff == 1 else oracle.median(src, (W, H, D), eff)
    set_budget(budget_for(planes, W, H, 2))
    op = make_op(f3d, "median_p")
    assert op.name == "CUDA Median Piecemeal"
    vin, vout = f3d.HostVolume(src.copy()), f3d.HostVolume(np.full(src.shape, np.nan, np.float32))
    op.execute(input=vin, output=vout, data_size=(W, H, D), radius=radius)
    assert same(vout.array[:D, :H, :W], expect[:D, :H, :W])
    vio = f3d.HostVolume(src.copy())
    op.execute(input=vio, output=vio, data_size=(W, H, D), radius=radius)
    assert same(vio.array[:D, :H, :W], expect[:D, :H, :W])
    op.destroy()
    for v in (vin, vout, vio):
        v.destroy()


def test_full_pipeline_equals_the_resident_driver_and_the_golden_crop(f3d):
    """full_pipeline: pre-blur and per-level median on host volumes as well -- OpticalFlowE's whole pipeline.  Against the
    committed default-pipeline fixture (48x40x24 crop of the reference's 128^3 pair) and against OpticalFlowE on a 120^3
    synthetic pair, with budgets that stream the finest levels; with and without the resident coarse levels."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    e = np.load(os.path.join(gold, "expected_oracle.npz"))
    i128 = np.load(os.path.join(gold, "inputs_128.npz"))
    crop = (slice(40, 64), slice(40, 80), slice(40, 88))
    f0 = i128["frame_0"].astype(np.float32)[crop].copy()
    f1 = i128["frame_1"].astype(np.float32)[crop].copy()
    for resident in (True, False):
        set_budget(budget_for(13 * 20, 48, 40, 13))
        got, (passes, streamed, on_device) = run_p(f3d, f0, f1, resident=resident, full=True)
        assert streamed >= 1
        for g, x, n in zip(got, e["crop128_flow"], "uvw"):
            assert same(g, x), f"crop128 resident={resident} {n}: max diff {np.abs(g - x).max()}"
    n = 120
    a, b = f3d.synth_pair(n, n, n)
    keep = a.copy(), b.copy()
    flow = f3d.OpticalFlow()
    flow.initialize(n, n, n)
    try:
        exp = flow.compute(a, b, silent=True, outer_iterations_count=5)
    finally:
        flow.destroy()
    set_budget(budget_for(13 * 52, n, n, 13))
    got, (passes, streamed, on_device) = run_p(f3d, a, b, full=True, outer_iterations_count=5)
    assert streamed >= 3 and on_device >= 20
    for g, x, c in zip(got, exp, "uvw"):
        assert same(g, x), f"{c}: max diff {np.abs(g - x).max()}"
    assert same(a, keep[0]) and same(b, keep[1])
