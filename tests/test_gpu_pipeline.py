"""Whole-path parity on the MI355X: the driver (OpticalFlowE::ComputeFlow through the C ABI) and the operator classes
against the oracle and the committed golden fixtures.  Tolerance: the north star allows RMS 1e-4; because every kernel
is bit-faithful the tests demand max |diff| == 0 (sign of zero aside) and report the RMS in the assertion message."""
import hashlib
import os

import numpy as np
import pytest

from conftest import flush_c_stdio, same

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def digest(*vols):
    h = hashlib.sha256()
    for v in vols:
        h.update(np.ascontiguousarray(v + np.float32(0.0)).tobytes())
    return h.hexdigest()


def run_flow(f3d, f0, f1, **kw):
    d, h, w = f0.shape
    flow = f3d.OpticalFlow()
    flow.initialize(w, h, d)
    try:
        return flow.compute(f0, f1, silent=True, **kw)
    finally:
        flow.destroy()


@pytest.fixture(scope="module")
def gold():
    e = np.load(os.path.join(GOLD, "expected_oracle.npz"))
    i128 = np.load(os.path.join(GOLD, "inputs_128.npz"))
    irub = np.load(os.path.join(GOLD, "inputs_rub.npz"))
    f0 = i128["frame_0"].astype(np.float32)
    f1 = i128["frame_1"].astype(np.float32)
    r0 = np.repeat(irub["slice_0"][None], int(irub["depth"]), axis=0).astype(np.float32)
    r1 = np.repeat(irub["slice_1"][None], int(irub["depth"]), axis=0).astype(np.float32)
    return dict(e=e, f0=f0, f1=f1, r0=r0, r1=r1)


def check3(got, exp, what):
    for g, e, n in zip(got, exp, "uvw"):
        assert np.isfinite(g).all(), f"{what}: {n} not finite"
        assert same(g, e), f"{what}: component {n} differs, rms {rms(g, e):.3e}, max {np.abs(g - e).max():.3e}"


def test_small_synthetic_pair_matches_oracle_live(f3d, oracle):
    f0, f1 = f3d.synth_pair(40, 36, 32)
    got = run_flow(f3d, f0, f1)
    (exp, levels) = oracle.compute_flow(f0, f1)
    assert levels == f3d.max_warp_level(40, 36, 32, 0.95) or levels == 40
    check3(got, exp, "40x36x32 synthetic, defaults")
    # property: the known translation (+2, -1, +0.5) is recovered in the textured interior
    inner = (slice(8, -8),) * 3
    means = [float(g[inner].mean()) for g in got]
    assert abs(means[0] - 2.0) < 0.15 and abs(means[1] + 1.0) < 0.15 and abs(means[2] - 0.5) < 0.15, means


def test_c1_plumbing_config(f3d, gold):
    """BASELINE config 1: 128^3 pair, 1 level, 1 x 5 iterations."""
    u, v, w = run_flow(f3d, gold["f0"], gold["f1"], warp_levels_count=1, outer_iterations_count=1, inner_iterations_count=5)
    c = slice(48, 80)
    check3([a[c, c, c] for a in (u, v, w)], gold["e"]["c1_crop"], "C1 centre crop")
    assert digest(u, v, w) == str(gold["e"]["c1_sha256"])


def test_crop_pipelines_match_golden(f3d, gold):
    crop = (slice(40, 64), slice(40, 80), slice(40, 88))
    got = run_flow(f3d, gold["f0"][crop].copy(), gold["f1"][crop].copy())
    check3(got, gold["e"]["crop128_flow"], "48x40x24 crop of the 128^3 pair")
    rc = (slice(0, 5), slice(100, 164), slice(200, 296))
    got = run_flow(f3d, gold["r0"][rc].copy(), gold["r1"][rc].copy())
    check3(got, gold["e"]["croprub_flow"], "96x64x5 crop of the rub pair")


def test_c2_full_128_defaults(f3d, gold):
    """BASELINE config 2: 128^3 pair, full pyramid (40 levels, 40 x 5)."""
    u, v, w = run_flow(f3d, gold["f0"], gold["f1"])
    check3([a[64] for a in (u, v, w)], gold["e"]["c2_slice_z"], "C2 z-slice")
    check3([a[:, 64] for a in (u, v, w)], gold["e"]["c2_slice_y"], "C2 y-slice")
    assert digest(u, v, w) == str(gold["e"]["c2_sha256"])


def test_c3_thin_slab_defaults(f3d, gold):
    """BASELINE config 3: 584 x 388 x 5, 10 levels, depth 5 -> 4, anisotropic spacing."""
    u, v, w = run_flow(f3d, gold["r0"], gold["r1"])
    check3([a[2][130:258, 228:356] for a in (u, v, w)], gold["e"]["c3_slice_z"], "C3 z-slice window")
    check3([a[:, 194] for a in (u, v, w)], gold["e"]["c3_slice_y"], "C3 y-slice")
    assert digest(u, v, w) == str(gold["e"]["c3_sha256"])
    # five identical slices in: w stays tiny
    assert float(np.abs(w).max()) < 1.0


def test_identical_frames_give_zero_flow(f3d):
    f0, _ = f3d.synth_pair(48, 40, 24)
    u, v, w = run_flow(f3d, f0, f0.copy(), warp_levels_count=8, outer_iterations_count=5)
    assert float(np.abs(u).max()) == 0.0 and float(np.abs(v).max()) == 0.0 and float(np.abs(w).max()) == 0.0


def test_resident_path_equals_host_path(f3d):
    f0, f1 = f3d.synth_pair(40, 36, 24)
    kw = dict(warp_levels_count=10, outer_iterations_count=6)
    a = run_flow(f3d, f0, f1, **kw)
    flow = f3d.OpticalFlow()
    flow.initialize(40, 36, 24)
    flow.upload(f0, f1)
    flow.compute_resident(**kw)
    flow.compute_resident(**kw)  # repeatable: the raw frames are not consumed
    b = flow.download()
    flow.destroy()
    check3(b, a, "resident vs host entry point")


def test_compute_into_the_callers_page_locked_volumes(f3d):
    """ComputeFlow with the caller's own (here page-locked) flow volumes, as the reference's caller has them: the same bits
    as into fresh arrays, wrong shapes refused before anything runs."""
    import ctypes as C
    f0, f1 = f3d.synth_pair(40, 36, 24)
    kw = dict(warp_levels_count=10, outer_iterations_count=6)
    a = run_flow(f3d, f0, f1, **kw)
    out = tuple(np.zeros((24, 36, 40), np.float32) for _ in range(3))
    hip = f3d.hip()
    pinned = [o for o in out if hip.f3d_host_register(C.c_void_p(o.ctypes.data), o.nbytes) == 0]
    flow = f3d.OpticalFlow()
    flow.initialize(40, 36, 24)
    try:
        got = flow.compute(f0, f1, out=out, **kw)
        assert all(g is o for g, o in zip(got, out))
        check3(got, a, "caller-owned outputs")
        with pytest.raises(ValueError):
            flow.compute(f0, f1, out=(out[0], out[1], np.zeros((24, 36, 41), np.float32)), **kw)
    finally:
        flow.destroy()
        for o in pinned:
            hip.f3d_host_unregister(C.c_void_p(o.ctypes.data))


@pytest.mark.parametrize("outer", [3, 4])
def test_solve_operator_through_the_bag(f3d, oracle, outer):
    """CudaOperationSolve with the reference's parameter keys; the swapped du/temp pointers come back through the bag, and the
    weights of the LAST outer iteration are in the caller's dev_phi / dev_ksi (the fused schedule ping-pongs with a pair the
    operator owns: 2 hand-overs for outer = 3, 3 for outer = 4)."""
    rng = np.random.default_rng(5)
    dims, cdims = (37, 21, 9), (64, 24, 12)
    W, H, D = dims
    cont = f3d.Containers(*cdims)

    def put(lo, hi):
        c = np.full((cdims[2], cdims[1], cdims[0]), np.nan, np.float32)
        c[:D, :H, :W] = rng.uniform(lo, hi, size=(D, H, W)).astype(np.float32)
        return c, cont.new(c)

    hosts, ptrs = zip(*[put(*r) for r in [(0, 255), (0, 255), (-2, 2), (-2, 2), (-2, 2)]])
    names = ["dev_flow_du", "dev_flow_dv", "dev_flow_dw", "dev_phi", "dev_ksi", "dev_temp_du", "dev_temp_dv", "dev_temp_dw"]
    extra = {n: cont.new() for n in names}
    op = f3d.Operation("solve")
    assert op.name == "CUDA Solve" and op.initialize(cont)
    h = (1.5, 1.2, 2.0)
    inner = 5
    vals = op.execute(dev_frame_0=ptrs[0], dev_frame_1=ptrs[1], dev_flow_u=ptrs[2], dev_flow_v=ptrs[3], dev_flow_w=ptrs[4],
                      outer_iterations_count=outer, inner_iterations_count=inner, equation_alpha=7.5,
                      equation_smoothness=0.001, equation_data=0.001, hx=h[0], hy=h[1], hz=h[2], data_size=dims, **extra)
    f3d.sync()
    # oracle
    du = np.full_like(hosts[0], np.nan); du[:, :, :W] = 0
    dv, dw = du.copy(), du.copy()
    for _ in range(outer):
        phi, ksi = oracle.phi_ksi(*hosts, du, dv, dw, dims, h, 0.001, 0.001)
        for _ in range(inner):
            du, dv, dw = oracle.solve_sweep(*hosts, du, dv, dw, phi, ksi, dims, h, 7.5)
    # one swap per launch: how many there are depends on how the level's sweeps are cut (two launches per outer iteration on a level
    # this small, three on large ones), so the roles may or may not have changed hands -- but the two containers are still the pair
    for c in ("u", "v", "w"):
        assert {vals[f"dev_flow_d{c}"], vals[f"dev_temp_d{c}"]} == {extra[f"dev_flow_d{c}"], extra[f"dev_temp_d{c}"]}
    for key, e in (("dev_flow_du", du), ("dev_flow_dv", dv), ("dev_flow_dw", dw), ("dev_phi", phi), ("dev_ksi", ksi)):
        g = cont.download(vals[key], cdims)
        assert same(g[:D, :H, :W], e[:D, :H, :W]), key
    op.destroy()
    cont.free()


@pytest.mark.parametrize("alpha", [-0.02, float("inf")])
def test_solve_operator_with_weights_the_fused_launches_refuse(f3d, oracle, alpha):
    """alpha / h^2 negative or not finite: the fused launches select w or +0 where solve_3d.cu:437-445 multiplies by (float)(flag), which
    is the same float only for a finite w that is not negative -- so their entry points refuse such parameters (include/f3d.h) and the
    operator takes the one-sweep launches by itself; the result is what the reference's arithmetic gives, bit pattern for bit pattern
    (signed zeros and NaN included)."""
    rng = np.random.default_rng(6)
    dims, cdims = (37, 21, 9), (64, 24, 12)
    W, H, D = dims
    cont = f3d.Containers(*cdims)

    def put(lo, hi):
        c = np.full((cdims[2], cdims[1], cdims[0]), np.nan, np.float32)
        c[:D, :H, :W] = rng.uniform(lo, hi, size=(D, H, W)).astype(np.float32)
        return c, cont.new(c)

    hosts, ptrs = zip(*[put(*r) for r in [(0, 255), (0, 255), (-2, 2), (-2, 2), (-2, 2)]])
    names = ["dev_flow_du", "dev_flow_dv", "dev_flow_dw", "dev_phi", "dev_ksi", "dev_temp_du", "dev_temp_dv", "dev_temp_dw"]
    extra = {n: cont.new() for n in names}
    h = (1.5, 1.2, 2.0)
    # the fused entries say no ...
    hip = f3d.hip()
    outs = [extra[n] for n in ("dev_temp_du", "dev_temp_dv", "dev_temp_dw")]
    incs = [extra[n] for n in ("dev_flow_du", "dev_flow_dv", "dev_flow_dw")]
    assert hip.f3d_solve_sweep2(*ptrs, *incs, extra["dev_phi"], extra["dev_ksi"], W, H, D, *h, alpha, *outs, None) != 0
    assert b"finite and not negative" in hip.f3d_last_error()
    # ... and the operator computes what the reference would
    op = f3d.Operation("solve")
    assert op.initialize(cont)
    outer, inner = 2, 3
    vals = op.execute(dev_frame_0=ptrs[0], dev_frame_1=ptrs[1], dev_flow_u=ptrs[2], dev_flow_v=ptrs[3], dev_flow_w=ptrs[4],
                      outer_iterations_count=outer, inner_iterations_count=inner, equation_alpha=alpha,
                      equation_smoothness=0.001, equation_data=0.001, hx=h[0], hy=h[1], hz=h[2], data_size=dims, **extra)
    f3d.sync()
    du = np.full_like(hosts[0], np.nan); du[:, :, :W] = 0
    dv, dw = du.copy(), du.copy()
    for _ in range(outer):
        phi, ksi = oracle.phi_ksi(*hosts, du, dv, dw, dims, h, 0.001, 0.001)
        for _ in range(inner):
            du, dv, dw = oracle.solve_sweep(*hosts, du, dv, dw, phi, ksi, dims, h, alpha)
    for key, e in (("dev_flow_du", du), ("dev_flow_dv", dv), ("dev_flow_dw", dw), ("dev_phi", phi), ("dev_ksi", ksi)):
        g = cont.download(vals[key], cdims)[:D, :H, :W]
        e = e[:D, :H, :W]
        nan_g, nan_e = np.isnan(g), np.isnan(e)
        assert np.array_equal(nan_g, nan_e), key
        assert np.array_equal(g.view(np.uint32)[~nan_g], e.view(np.uint32)[~nan_e]), key
    op.destroy()
    cont.free()


def test_missing_key_is_reported_not_fatal(f3d, capfd):
    cont = f3d.Containers(16, 8, 8)
    p = cont.new()
    op = f3d.Operation("add")
    assert op.initialize(cont)
    op.execute(operand_0=p, data_size=(8, 8, 8))  # operand_1 missing: prints and returns, like the reference macro
    op.destroy()
    cont.free()
    flush_c_stdio()
    assert "Missing parameter 'operand_1'" in capfd.readouterr().out


def test_cli_frame_sequence_and_stats(f3d, tmp_path):
    """bin/flow3d on four frames: one Initialize, three rotating frame containers (every frame uploaded once, the next one
    while the current pair solves, the previous flow downloaded and written beside the next solve), the flow of every
    consecutive pair written, --stats printing what the device statistics return; each pair equals a fresh
    OpticalFlow.compute of the same two frames."""
    import re
    import subprocess
    W, H, D = 48, 40, 24
    f0, f1 = f3d.synth_pair(W, H, D)
    frames = [np.round(np.clip(f0, 0, 255)), np.round(np.clip(f1, 0, 255)), np.round(np.clip(0.5 * (f0 + f1), 0, 255)),
              np.round(np.clip(0.25 * f0 + 0.75 * f1, 0, 255))]
    paths = []
    for k, fr in enumerate(frames):
        p = tmp_path / f"frame{k}.raw"
        fr.astype(np.uint8).tofile(p)
        paths.append(str(p))
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-flow3d_amd", "bin", "flow3d")
    prefix = str(tmp_path / "seq")
    kw = dict(warp_levels_count=5, outer_iterations_count=3)
    run = subprocess.run([exe, "--dims", str(W), str(H), str(D), "--frames", *paths, "--out", prefix, "--levels", "5", "--outer", "3",
                          "--stats", "--silent"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    stats = re.findall(r"Flow magnitude\s+min:\s*([\d.]+)\s+max:\s*([\d.]+)\s+avg:\s*([\d.]+)", run.stdout)
    assert len(stats) == 3, run.stdout
    residual = re.findall(r"Registration residual .*?rms:\s*([\d.]+).*?unregistered\s+rms:\s*([\d.]+)", run.stdout)
    assert len(residual) == 3 and all(float(a) < float(b) for a, b in residual), run.stdout
    assert len(re.findall(r"^level\s+\d+ \(", run.stdout, flags=re.M)) == 3 * 5, run.stdout   # five levels per pair
    assert len(re.findall(r"^pair \d of 3:", run.stdout, flags=re.M)) == 3, run.stdout
    for k in range(3):
        a = frames[k].astype(np.uint8).astype(np.float32)
        b = frames[k + 1].astype(np.uint8).astype(np.float32)
        flow = f3d.OpticalFlow()
        flow.initialize(W, H, D)
        exp = flow.compute(a, b, silent=True, **kw)
        flow.destroy()
        got = [np.fromfile(f"{prefix}_{k}_flow-{c}-{W}-{H}-{D}.raw", np.float32).reshape(D, H, W) for c in "uvw"]
        for g, e in zip(got, exp):
            assert same(g, e)
        mag = np.sqrt(exp[0] * exp[0] + exp[1] * exp[1] + exp[2] * exp[2])
        mn, mx, avg = (float(v) for v in stats[k])
        assert abs(mn - mag.min()) < 1e-3 and abs(mx - mag.max()) < 1e-3 and abs(avg - mag.mean()) < 1e-3


def test_level_statistics_and_final_residual(f3d, oracle):
    """The driver's diagnostics: one record per pyramid level with the level geometry of the schedule, a
    residual that the registration brings down, flow statistics equal to the statistics operator's; and the final residual --
    frame_1 registered with the computed flow against frame_0 -- equal to the oracle's warp + scan of the same volumes."""
    W, H, D = 48, 40, 36
    f0, f1 = f3d.synth_pair(W, H, D)
    kw = dict(warp_levels_count=12, outer_iterations_count=6)
    flow = f3d.OpticalFlow()
    flow.initialize(W, H, D)
    try:
        flow.upload(f0, f1)
        flow.set_level_stats(True)
        flow.compute_resident(silent=True, **kw)
        u, v, w = flow.download()
        levels = flow.level_stats()
        reg, unreg = flow.final_residual()
        flow.set_level_stats(False)
        flow.compute_resident(silent=True, **kw)
        assert flow.level_stats() == []                      # nothing is collected unless asked for
        for a, b in zip(flow.download(), (u, v, w)):
            assert same(a, b)                                # ... and collecting changes nothing
    finally:
        flow.destroy()
    assert [st["level"] for st in levels] == list(range(11, -1, -1))
    for st in levels:
        (cw, ch, cd), _ = oracle.level_geometry(W, H, D, 0.95, st["level"])
        assert (st["width"], st["height"], st["depth"]) == (cw, ch, cd)
        assert st["residual_rms"] >= 0 and st["residual_max_abs"] >= st["residual_mean_abs"] >= 0
        assert 0 <= st["flow_min"] <= st["flow_avg"] <= st["flow_max"]
    assert levels[0]["residual_rms"] > 2 * levels[-1]["residual_rms"]      # the flow handed down registers better and better
    mag = np.sqrt(u.astype(np.float64) ** 2 + v.astype(np.float64) ** 2 + w.astype(np.float64) ** 2)
    assert abs(levels[-1]["flow_avg"] - mag.mean()) < 1e-5 * mag.mean()
    assert levels[-1]["flow_max"] == np.float32(np.sqrt(u * u + v * v + w * w).max())
    # final residual: oracle warp (pinned to the reference's host warp) + oracle scan
    warped = oracle.warp(f0, f1, u, v, w, (W, H, D), (1.0, 1.0, 1.0))
    n = W * H * D
    for got, (ssq, sab, mx) in ((reg, oracle.residual_stats(f0, warped, (W, H, D))), (unreg, oracle.residual_stats(f0, f1, (W, H, D)))):
        assert abs(got[0] - np.sqrt(ssq / n)) <= 1e-10 * got[0] and abs(got[1] - sab / n) <= 1e-10 * got[1] and got[2] == mx
    assert reg[0] < 0.5 * unreg[0]


def test_fused_pairs_equal_single_sweeps_end_to_end(tmp_path):
    """A 200^3 synthetic pair through bin/flow3d twice: inner sweeps in fused pairs (default) and one launch per sweep
    (F3D_FUSED_SWEEPS=0), exact uniform-divisor division on and off (F3D_UDIV=0).  Byte-identical flow files; the
    recovered flow approaches the synthetic translation (2, -1, 0.5) in the textured interior."""
    import subprocess
    S = 200
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-flow3d_amd", "bin", "flow3d")
    digests = []
    for tag, env in (("fused", {}), ("plain", {"F3D_FUSED_SWEEPS": "0", "F3D_UDIV": "0"})):
        prefix = str(tmp_path / tag)
        run = subprocess.run([exe, "--dims", str(S), str(S), str(S), "--synthetic", "--out", prefix, "--levels", "24", "--outer", "10",
                              "--silent"], capture_output=True, text=True, timeout=300, env={**os.environ, **env})
        assert run.returncode == 0, run.stdout + run.stderr
        digests.append([hashlib.sha256(open(f"{prefix}_flow-{c}-{S}-{S}-{S}.raw", "rb").read()).hexdigest() for c in "uvw"])
    assert digests[0] == digests[1]
    u, v, w = (np.fromfile(str(tmp_path / f"fused_flow-{c}-{S}-{S}-{S}.raw"), np.float32).reshape(S, S, S) for c in "uvw")
    core = (slice(60, 140),) * 3
    # 24 levels x 10 outer iterations are not converged: the translation is recovered to ~10 %
    assert abs(u[core].mean() - 2.0) < 0.4 and abs(v[core].mean() + 1.0) < 0.3 and abs(w[core].mean() - 0.5) < 0.2


def test_two_drivers_on_lanes_of_their_own_run_side_by_side(f3d):
    """Two host threads, each with a lane of its own (stream + container geometry) and a driver of its own, solve different pairs of
    DIFFERENT sizes at the same time: every result equals what the same pair gives alone on the default lane -- the launches of the
    two drivers share nothing but the device (and interleave freely on it)."""
    import threading
    cases = [((40, 36, 24), dict(warp_levels_count=8, outer_iterations_count=6)),
             ((56, 30, 33), dict(warp_levels_count=10, outer_iterations_count=5))]
    pairs = [f3d.synth_pair(*dims) for dims, _ in cases]
    alone = []
    for (dims, kw), (f0, f1) in zip(cases, pairs):
        flow = f3d.OpticalFlow()
        flow.initialize(*dims)
        alone.append(flow.compute(f0, f1, silent=True, **kw))
        flow.destroy()
    results, errors = [None, None], []

    def work(i):
        try:
            with f3d.Lane():
                assert f3d.hip().f3d_lane_is_private() == 1
                (dims, kw), (f0, f1) = cases[i], pairs[i]
                flow = f3d.OpticalFlow()
                flow.initialize(*dims)
                for _ in range(3):          # several solves per thread: the two chains of launches overlap for a while
                    results[i] = flow.compute(f0, f1, silent=True, **kw)
                flow.destroy()
        except Exception as e:              # noqa: BLE001 - reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert f3d.hip().f3d_lane_is_private() == 0
    for i in range(2):
        for g, e, c in zip(results[i], alone[i], "uvw"):
            assert same(g, e), f"driver {i} on its own lane: component {c} differs, max {np.abs(g - e).max():.3e}"


def test_cli_concurrent_pairs(f3d, tmp_path):
    """bin/flow3d --concurrent 2 on five frames: two worker threads, each with a lane and a driver of its own, take the four pairs
    alternately; every pair equals a fresh OpticalFlow.compute of its two frames."""
    import re
    import subprocess
    W, H, D = 48, 40, 24
    f0, f1 = f3d.synth_pair(W, H, D)
    mix = [0.0, 1.0, 0.5, 0.75, 0.2]
    frames = [np.round(np.clip((1 - a) * f0 + a * f1, 0, 255)) for a in mix]
    paths = []
    for k, fr in enumerate(frames):
        p = tmp_path / f"frame{k}.raw"
        fr.astype(np.uint8).tofile(p)
        paths.append(str(p))
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-flow3d_amd", "bin", "flow3d")
    prefix = str(tmp_path / "par")
    kw = dict(warp_levels_count=6, outer_iterations_count=4)
    run = subprocess.run([exe, "--dims", str(W), str(H), str(D), "--frames", *paths, "--out", prefix, "--levels", "6", "--outer", "4",
                          "--silent", "--concurrent", "2"], capture_output=True, text=True, timeout=180)
    assert run.returncode == 0, run.stdout + run.stderr
    done = re.findall(r"^pair (\d) of 4 done by worker (\d)", run.stdout, flags=re.M)
    assert sorted(int(p) for p, _ in done) == [1, 2, 3, 4] and {w for _, w in done} == {"0", "1"}, run.stdout
    assert re.search(r"4 pairs in [\d.]+ s: [\d.]+ pairs per second with 2 at a time", run.stdout), run.stdout
    for k in range(4):
        a = frames[k].astype(np.uint8).astype(np.float32)
        b = frames[k + 1].astype(np.uint8).astype(np.float32)
        flow = f3d.OpticalFlow()
        flow.initialize(W, H, D)
        exp = flow.compute(a, b, silent=True, **kw)
        flow.destroy()
        got = [np.fromfile(f"{prefix}_{k}_flow-{c}-{W}-{H}-{D}.raw", np.float32).reshape(D, H, W) for c in "uvw"]
        for g, e in zip(got, exp):
            assert same(g, e)
