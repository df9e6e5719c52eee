"""The product's HOST code without a GPU: libf3d_host.so (drivers, operators, slab planner and exchanges, out-of-core chunking)
built against tests/cpu_device -- the C ABI of include/f3d.h on host memory with the oracle's kernels as the compute -- once
plainly and once under AddressSanitizer + UndefinedBehaviorSanitizer.  Whatever the drivers do with containers, swaps, windows,
halos and chunks, a whole ComputeFlow must come out bit-identical to the oracle's own whole-pipeline function; under the
sanitizers the same runs must finish without a report.

Each case runs in a process of its own: the package is pointed at the stand-in library directory BEFORE it loads anything
(a module global of the binding, patched here by the test -- the product never does that), and the sanitizer runtime has to
be preloaded into the interpreter."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU = os.path.join(ROOT, "tests", "cpu_device")

CASE = textwrap.dedent('''
    import importlib, os, sys
    import numpy as np
    sys.path.insert(0, os.environ["F3D_ROOT"])
    pkg = importlib.import_module("cuda-flow3d_amd")
    pkg._LIBDIR = os.environ["F3D_TEST_LIBDIR"]          # test-only: the host-memory stand-in
    from oracle import oracle as orc
    what = sys.argv[1]
    W, H, D = 26, 22, 20
    f0, f1 = pkg.synth_pair(W, H, D)
    kw = dict(warp_levels_count=5, outer_iterations_count=3, inner_iterations_count=5)
    (eu, ev, ew), _ = orc.compute_flow(f0, f1, **kw)
    same = lambda a, b: a.shape == b.shape and bool(np.all(a == b)) and not np.isnan(a).any()
    if what == "resident":
        flow = pkg.OpticalFlow(); flow.initialize(W, H, D)
        got = flow.compute(f0, f1, silent=True, **kw)
        flow.upload(f0, f1); flow.set_level_stats(True); flow.compute_resident(silent=True, **kw)
        got2 = flow.download(); stats = flow.level_stats(); reg, unreg = flow.final_residual()
        flow.destroy()
        assert all(same(g, e) for g, e in zip(got, (eu, ev, ew))), "OpticalFlowE on the host backend differs from the oracle"
        assert all(same(g, e) for g, e in zip(got2, (eu, ev, ew)))
        assert len(stats) == 5 and reg[0] < unreg[0]
        # even inner count (pairs only) and a single sweep per outer iteration (no pair at all)
        for inner in (4, 1):
            kw2 = dict(kw, inner_iterations_count=inner)
            (xu, xv, xw), _ = orc.compute_flow(f0, f1, **kw2)
            flow = pkg.OpticalFlow(); flow.initialize(W, H, D)
            got = flow.compute(f0, f1, silent=True, **kw2); flow.destroy()
            assert all(same(g, e) for g, e in zip(got, (xu, xv, xw))), inner
    elif what == "slabs":
        for ranks in (2, 3):
            flow = pkg.SlabOpticalFlow(ranks, list(range(ranks)), halo_capacity=16); flow.initialize(W, H, D)
            got = flow.compute(f0, f1, **kw); flow.destroy()
            assert all(same(g, e) for g, e in zip(got, (eu, ev, ew))), f"{ranks} slabs on the host backend differ from the oracle"
        # one exchange per solver stage (2 / 1 / 3 planes) instead of six planes once per outer iteration: same bits
        os.environ["F3D_SLAB_EXCHANGE"] = "stage"
        for ranks in (2, 3, 4):
            flow = pkg.SlabOpticalFlow(ranks, list(range(ranks)), halo_capacity=16); flow.initialize(W, H, D)
            got = flow.compute(f0, f1, **kw); n_stage = flow.stage_exchanges(); flow.destroy()
            assert n_stage == 5 * (3 * 3 - 1), n_stage      # 5 levels x (3 outer iterations x 3 stages, none after the last)
            assert all(same(g, e) for g, e in zip(got, (eu, ev, ew))), f"{ranks} slabs exchanging per stage differ from the oracle"
        del os.environ["F3D_SLAB_EXCHANGE"]
        # a rank list that is not 0 .. n-1 is refused
        bad = pkg.SlabOpticalFlow(2, [1, 0], halo_capacity=16)
        try:
            bad.initialize(W, H, D); raise SystemExit("a permuted rank list was accepted")
        except pkg.F3dError:
            pass
        bad.destroy()
    elif what == "gather":
        # a pair that moves by ~7 planes along z: with 4 slabs of 8-10 planes and 8 planes of halo room the warp needs frame 1 from
        # ranks beyond the neighbour -- the driver gathers it into a container of its own instead of giving up (SURVEY 8e fallback)
        W2, H2, D2 = 24, 20, 40
        g0, _ = pkg.synth_pair(W2, H2, D2)
        g1 = np.ascontiguousarray(np.roll(g0, 7, axis=0))
        kw3 = dict(warp_levels_count=12, outer_iterations_count=3, inner_iterations_count=5)
        (xu, xv, xw), _ = orc.compute_flow(g0, g1, **kw3)
        assert np.abs(xw).max() > 0.5, np.abs(xw).max()
        # halo room 7 = the K + 1 = 6 planes of the widened sweeps + 1: any z motion makes the warp reach beyond it, and on the
        # coarse levels (slabs of 5-6 planes) the gathered planes come from ranks beyond the neighbour
        for ranks, halo in ((4, 7), (3, 7)):
            flow = pkg.SlabOpticalFlow(ranks, list(range(ranks)), halo_capacity=halo); flow.initialize(W2, H2, D2)
            got = flow.compute(g0, g1, **kw3); gathered = flow.gathered_warps(); flow.destroy()
            assert gathered >= 1, "the reach never exceeded the halo room: the case does not test the gather"
            assert all(same(g, e) for g, e in zip(got, (xu, xv, xw))), f"{ranks} slabs with a gathered frame differ from the oracle"
    elif what == "piecemeal":
        os.environ["F3D_P_PIN"] = "2"                     # (the driver page-locks only volumes of 32 MiB and more by itself: here every size, so that the two-set schedule runs)
        os.environ["F3D_P_BUDGET_MB"] = "1.3"             # the finest levels go through the "device" in chunks
        flow = pkg.PiecemealOpticalFlow(); flow.initialize(W, H, D); flow.set_full_pipeline(True)
        got = flow.compute(f0, f1, silent=True, **kw); passes, streamed, resident = flow.stats(); flow.destroy()
        assert streamed >= 1, (passes, streamed, resident)
        assert all(same(g, e) for g, e in zip(got, (eu, ev, ew))), "OpticalFlowP --full on the host backend differs from the oracle"
        # frame 1 registered inside the solver's first residency (default) against the separate registration operator, on a pair that
        # moves along z (a warp reach of several planes: the unregistered frame arrives in more than one piece, or the solver declines
        # and the driver registers the classical way), with and without the caller's frames surviving
        W2, H2, D2 = 24, 20, 40
        g0, _ = pkg.synth_pair(W2, H2, D2)
        for shift, budget in ((2, "2.0"), (2, "1.0"), (7, "1.3")):
            g1 = np.ascontiguousarray(np.roll(g0, shift, axis=0))
            # (14 levels: at 2 MB the coarsest run wholly on the "device" before the host levels: both parts of the driver in one run)
            kw3 = dict(warp_levels_count=14, outer_iterations_count=3, inner_iterations_count=5)
            os.environ["F3D_P_BUDGET_MB"] = budget
            runs = {}
            for fused in ("1", "0"):
                os.environ["F3D_P_FUSED_WARP"] = fused
                a0, a1 = g0.copy(), g1.copy()
                flow = pkg.PiecemealOpticalFlow(); flow.initialize(W2, H2, D2)
                runs[fused] = (flow.compute(a0, a1, silent=True, **kw3), flow.levels_registered_inside(), flow.stats())
                flow.destroy()
                assert same(a0, g0) and same(a1, g1), "the out-of-core driver changed the caller's frames"
            del os.environ["F3D_P_FUSED_WARP"]
            assert runs["0"][1] == 0 and runs["1"][2][1] >= 1, (shift, budget, runs["0"][1:], runs["1"][1:])
            assert budget != "2.0" or runs["1"][2][2] >= 1, ("no level ran wholly on the device", runs["1"][2])
            if shift == 2:
                assert runs["1"][1] >= 1, (shift, budget, runs["1"][1:])
            assert np.abs(runs["0"][0][2]).max() > 0.3 * shift / 7
            assert all(same(a, b) for a, b in zip(runs["1"][0], runs["0"][0])), (shift, budget, "registration inside the solver differs")
            print(f"piecemeal shift {shift} budget {budget} MB: {runs['1'][1]} level(s) registered inside, stats {runs['1'][2]}")
            # the two frames and u, v, w held on the device for the whole level beside smaller chunk sets (three fields up per residency
            # instead of eight): pinned on wherever the budget allows it, pinned off, and left to the cost model -- same bits
            kept = {}
            for mode in ("1", "0", None):
                if mode is None:
                    os.environ.pop("F3D_P_CONSTANTS", None)
                else:
                    os.environ["F3D_P_CONSTANTS"] = mode
                for fused_warp in ("1", "0"):
                    os.environ["F3D_P_FUSED_WARP"] = fused_warp
                    a0, a1 = g0.copy(), g1.copy()
                    flow = pkg.PiecemealOpticalFlow(); flow.initialize(W2, H2, D2)
                    got = flow.compute(a0, a1, silent=True, **kw3); kept[(mode, fused_warp)] = flow.levels_with_constants_on_device(); flow.destroy()
                    assert same(a0, g0) and same(a1, g1), "the out-of-core driver changed the caller's frames"
                    assert all(same(a, b) for a, b in zip(got, runs["0"][0])), (shift, budget, mode, fused_warp, "constants on the device differ")
                del os.environ["F3D_P_FUSED_WARP"]
            os.environ.pop("F3D_P_CONSTANTS", None)
            assert kept[("0", "1")] == 0 and kept[("0", "0")] == 0, kept
            assert budget != "2.0" or kept[("1", "1")] >= 1, ("no level could hold its constant fields: the layout was not exercised", kept)
            print(f"   constants on the device: {kept}")
    elif what == "reinit":
        # a solve operator initialised twice WITHOUT Destroy in between, the second time for a bigger container: the second weight
        # pair of the fused last sweep must follow the container (under the sanitizers a stale, smaller buffer is a heap overflow),
        # and the weights of the last outer iteration must end in the caller's dev_phi / dev_ksi for odd and even hand-over counts
        op = pkg.Operation("solve")
        rng = np.random.default_rng(3)
        for cdims, dims, outer in (((64, 12, 6), (30, 11, 6), 3), ((128, 24, 12), (70, 21, 9), 4), ((64, 12, 6), (30, 11, 6), 2)):
            cw, ch, cd = cdims; w, h, d = dims
            cont = pkg.Containers(*cdims)
            def put(lo, hi):
                c = np.zeros((cd, ch, cw), np.float32)
                c[:d, :h, :w] = rng.uniform(lo, hi, size=(d, h, w)).astype(np.float32)
                return c, cont.new(c)
            hosts, ptrs = zip(*[put(*r) for r in [(0, 255), (0, 255), (-2, 2), (-2, 2), (-2, 2)]])
            names = ["dev_flow_du", "dev_flow_dv", "dev_flow_dw", "dev_phi", "dev_ksi", "dev_temp_du", "dev_temp_dv", "dev_temp_dw"]
            extra = {n: cont.new() for n in names}
            assert op.initialize(cont)
            sp = (1.5, 1.2, 2.0)
            vals = op.execute(dev_frame_0=ptrs[0], dev_frame_1=ptrs[1], dev_flow_u=ptrs[2], dev_flow_v=ptrs[3], dev_flow_w=ptrs[4],
                              outer_iterations_count=outer, inner_iterations_count=5, equation_alpha=7.5, equation_smoothness=0.001,
                              equation_data=0.001, hx=sp[0], hy=sp[1], hz=sp[2], data_size=dims, **extra)
            pkg.sync()
            du = np.zeros_like(hosts[0]); dv, dw = du.copy(), du.copy()
            for _ in range(outer):
                phi, ksi = orc.phi_ksi(*hosts, du, dv, dw, dims, sp, 0.001, 0.001)
                for _ in range(5):
                    du, dv, dw = orc.solve_sweep(*hosts, du, dv, dw, phi, ksi, dims, sp, 7.5)
            for key, e in (("dev_flow_du", du), ("dev_flow_dv", dv), ("dev_flow_dw", dw), ("dev_phi", phi), ("dev_ksi", ksi)):
                g = cont.download(vals[key], cdims)
                assert same(g[:d, :h, :w], e[:d, :h, :w]), (cdims, key)
            cont.free()
        op.destroy()
    elif what == "solve_slab":
        # the Solve operator under a z-slab whose container starts BEFORE the volume (rank 0 of the z-slab driver: z_base = own.lo -
        # halo < 0): the increments must be cleared on the planes the container really holds, and the result must be the unsplit one
        # (round-3 advisor finding: the box fill was handed z_lo = z_base < 0, failed, and the sweeps ran on uncleared increments)
        op = pkg.Operation("solve")
        rng = np.random.default_rng(11)
        dims = (30, 11, 12); w, h, d = dims
        for z_base, cd in ((-7, 26), (-3, 15), (0, 12)):
            cdims = (64, 12, cd); cw, ch, _ = cdims
            cont = pkg.Containers(*cdims)
            def put(lo, hi):
                v = np.zeros((d, ch, cw), np.float32)
                v[:, :h, :w] = rng.uniform(lo, hi, size=(d, h, w)).astype(np.float32)
                p = cont.new()
                cont.upload(p, v, plane0=-z_base)       # volume plane 0 sits at container plane -z_base
                return v, p
            hosts, ptrs = zip(*[put(*r) for r in [(0, 255), (0, 255), (-2, 2), (-2, 2), (-2, 2)]])
            names = ["dev_flow_du", "dev_flow_dv", "dev_flow_dw", "dev_phi", "dev_ksi", "dev_temp_du", "dev_temp_dv", "dev_temp_dw"]
            extra = {n: cont.new() for n in names}     # NaN-poisoned: an increment that is not cleared shows
            assert op.initialize(cont)
            slab = pkg.Slab(z_base, 0, d)
            op.set_slab(slab)
            sp = (1.5, 1.2, 2.0)
            vals = op.execute(dev_frame_0=ptrs[0], dev_frame_1=ptrs[1], dev_flow_u=ptrs[2], dev_flow_v=ptrs[3], dev_flow_w=ptrs[4],
                              outer_iterations_count=3, inner_iterations_count=5, equation_alpha=7.5, equation_smoothness=0.001,
                              equation_data=0.001, hx=sp[0], hy=sp[1], hz=sp[2], data_size=dims, **extra)
            pkg.sync()
            op.set_slab(None)
            du = np.zeros_like(hosts[0]); dv, dw = du.copy(), du.copy()
            for _ in range(3):
                phi, ksi = orc.phi_ksi(*hosts, du, dv, dw, dims, sp, 0.001, 0.001)
                for _ in range(5):
                    du, dv, dw = orc.solve_sweep(*hosts, du, dv, dw, phi, ksi, dims, sp, 7.5)
            for key, e in (("dev_flow_du", du), ("dev_flow_dv", dv), ("dev_flow_dw", dw), ("dev_phi", phi), ("dev_ksi", ksi)):
                g = cont.download(vals[key], (cw, ch, d), plane0=-z_base)
                assert same(g[:d, :h, :w], e[:d, :h, :w]), (z_base, key)
            cont.free()
        op.destroy()
    elif what == "batch":
        # ExecuteBatch of the add / median / resample operators: bags that describe one box go out together (f3d_*_n), bags that do
        # not -- another size, a shared temp, an output that is another bag's input -- one after the other; the oracle's values either way
        rng = np.random.default_rng(5)
        cdims = (40, 24, 12); cw, ch, cd = cdims
        cont = pkg.Containers(*cdims)
        def vol(dims):
            w, h, d = dims
            c = np.zeros((cd, ch, cw), np.float32)
            c[:d, :h, :w] = rng.uniform(-2, 2, size=(d, h, w)).astype(np.float32)
            return c
        box = lambda a, dims: a[:dims[2], :dims[1], :dims[0]]
        ops = {n: pkg.Operation(n) for n in ("add", "median", "resample")}
        cont.new()                                        # the first allocation fixes the pitch the operators are initialised with
        assert all(o.initialize(cont) for o in ops.values())
        for sizes in (((30, 21, 9),) * 3, ((30, 21, 9), (30, 21, 9), (17, 9, 6)), ((30, 21, 9),) * 2):
            a = [vol(s) for s in sizes]; b = [vol(s) for s in sizes]
            pa = [cont.new(x) for x in a]; pb = [cont.new(x) for x in b]
            ops["add"].execute_batch([dict(operand_0=pa[i], operand_1=pb[i], data_size=sizes[i]) for i in range(len(sizes))])
            sums = []
            for i, s in enumerate(sizes):
                e = a[i].copy(); orc.add(e, b[i], s); sums.append(e)
                assert same(box(cont.download(pa[i], cdims), s), box(e, s)), ("add", sizes, i)
            po = [cont.new() for _ in sizes]
            ops["median"].execute_batch([dict(dev_input=pa[i], dev_output=po[i], data_size=sizes[i], radius=5) for i in range(len(sizes))])
            for i, s in enumerate(sizes):
                assert same(box(cont.download(po[i], cdims), s), box(orc.median(sums[i], s, 5), s)), ("median", sizes, i)
            to = (26, 24, 11)
            for shared in (False, True):
                pr = [cont.new() for _ in sizes]
                one = cont.new()
                pt = [one if shared else cont.new() for _ in sizes]
                ops["resample"].execute_batch([dict(dev_input=pa[i], dev_output=pr[i], dev_temp=pt[i], data_size=sizes[i],
                                                    resample_size=to) for i in range(len(sizes))])
                for i, s in enumerate(sizes):
                    assert same(box(cont.download(pr[i], cdims), to), box(orc.resample(sums[i], s, to), to)), ("resample", sizes, i, shared)
        # an output that is another bag's input: the median goes bag by bag (and the second bag sees the first one's result)
        x = [vol((30, 21, 9)) for _ in range(2)]
        px = [cont.new(v) for v in x]; py = cont.new()
        ops["median"].execute_batch([dict(dev_input=px[0], dev_output=px[1], data_size=(30, 21, 9), radius=3),
                                     dict(dev_input=px[1], dev_output=py, data_size=(30, 21, 9), radius=3)])
        m1 = orc.median(x[0], (30, 21, 9), 3)
        assert same(box(cont.download(py, cdims), (30, 21, 9)), box(orc.median(m1, (30, 21, 9), 3), (30, 21, 9)))
        for o in ops.values():
            o.destroy()
        cont.free()
    pkg.shutdown()
    print("ok", what)
''')


def build(target):
    subprocess.run(["make", "-C", CPU, target, "-j4"], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(CPU, "_build", "asan" if target == "asan" else "plain")


def run_case(what, libdir, sanitized):
    env = dict(os.environ, F3D_ROOT=ROOT, F3D_TEST_LIBDIR=libdir, OMP_NUM_THREADS="2")
    if sanitized:
        asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
        ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
        if not os.path.isabs(asan):
            pytest.skip("no libasan in this toolchain")
        env.update(LD_PRELOAD=f"{asan} {ubsan}", ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:handle_segv=1",
                   UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-c", CASE, what], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and f"ok {what}" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error:" not in out.stderr, out.stderr[-3000:]


@pytest.mark.parametrize("what", ["resident", "slabs", "piecemeal", "reinit", "gather", "batch", "solve_slab"])
def test_host_drivers_equal_the_oracle_on_the_cpu_backend(what):
    run_case(what, build("all"), sanitized=False)


@pytest.mark.parametrize("what", ["resident", "slabs", "piecemeal", "reinit", "gather", "batch", "solve_slab"])
def test_host_drivers_are_clean_under_asan_and_ubsan(what):
    run_case(what, build("asan"), sanitized=True)


def test_out_of_core_operator_tests_on_the_cpu_backend():
    """The GPU tests of the out-of-core operators and driver layouts (tests/test_gpu_piecemeal.py: solver plans and residencies with
    both field layouts, the resample operator on two buffer sets, registration inside the solver, constant fields held on the device)
    run against the host-memory stand-in as well: the chunk, halo, hand-over and storage-trading logic is host code, and this is
    where a caller volume lost among the scratch volumes was found in round 4.  (Copy queues are synchronous here: orderings between
    queues are the GPU run's to check.)"""
    libdir = build("all")
    env = dict(os.environ, F3D_LIBDIR=libdir, OMP_NUM_THREADS="4", F3D_P_PIN="2")
    pick = "test_solve_matches_oracle or test_resample_matches_oracle or test_registration_inside or test_constant_fields_held"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_piecemeal.py"), "-q", "-x", "-m", "gpu", "-k", pick,
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    # (the operators' own console lines reach the pipe after pytest's summary: look at all of it)
    import re
    summary = re.findall(r"(\d+) passed", out.stdout)
    assert out.returncode == 0 and summary and int(summary[-1]) >= 25 and " failed" not in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])
