"""Per-kernel parity on the MI355X: every launcher of include/f3d.h against the CPU oracle on the same seeded
inputs.  Float work, but the kernels keep the reference's expression trees with contraction off, so the bar is
EXACT equality (tolerance 0; only the sign of a zero may differ in the median).  Boxes have odd sizes and sit in
the corner of a larger NaN-poisoned container, so partial tiles, mirror halos, depth < tile and stale reads
outside the sub-box are all exercised."""
import ctypes as C

import numpy as np
import pytest

from conftest import bit_same, box_in_container, same

pytestmark = pytest.mark.gpu

# (W,H,D) box, (Wc,Hc,Dc) container
CASES = [
    ((37, 21, 9), (64, 32, 16)),
    ((64, 8, 5), (64, 8, 5)),
    ((70, 70, 70), (128, 72, 70)),
    ((5, 4, 4), (16, 8, 8)),
    ((131, 7, 13), (192, 8, 16)),
    ((62, 4, 4), (62, 4, 4)),
    ((63, 5, 6), (63, 5, 6)),
    ((125, 9, 7), (125, 9, 7)),
    ((129, 10, 9), (192, 12, 9)),   # W = 64 k + 1: the last column of the volume is a tile's halo column
    ((65, 6, 5), (65, 6, 5)),
]
SPACINGS = [(1.0, 1.0, 1.0), (7.1, 1.6, 1.25)]


class Dev:
    """Uploads host containers, runs a launcher, downloads the box."""

    def __init__(self, f3d, cdims):
        self.f3d = f3d
        self.cdims = cdims
        self.cont = f3d.Containers(*cdims)
        self.cont.alloc(fill=0xFF)
        self.cont.set_current()

    def put(self, host_container):
        p = self.cont.new()
        self.cont.upload(p, host_container)
        return p

    def out(self):
        return self.cont.new()

    def get(self, p):
        self.f3d.sync()
        return self.cont.download(p, self.cdims)

    def close(self):
        self.f3d.sync()
        self.cont.free()


def solver_inputs(rng, dims, cdims):
    mk = lambda lo, hi: box_in_container(rng, dims, cdims, lo, hi)
    f0, f1 = mk(0, 255), mk(0, 255)
    u, v, w = mk(-3, 3), mk(-3, 3), mk(-3, 3)
    du, dv, dw = mk(-0.5, 0.5), mk(-0.5, 0.5), mk(-0.5, 0.5)
    return [f0, f1, u, v, w, du, dv, dw]


@pytest.mark.parametrize("dims,cdims", CASES)
@pytest.mark.parametrize("h", SPACINGS)
def test_phi_ksi_and_sweeps(f3d, oracle, dims, cdims, h):
    rng = np.random.default_rng(hash((dims, h)) % 2**32)
    W, H, D = dims
    arrs = solver_inputs(rng, dims, cdims)
    eps_s, eps_d, alpha = 0.001, 0.001, 7.5
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, eps_s, eps_d)

    dev = Dev(f3d, cdims)
    try:
        ptr = [dev.put(a) for a in arrs]
        phi, ksi = dev.out(), dev.out()
        f3d.check(f3d.hip().f3d_phi_ksi(*ptr, W, H, D, *h, eps_s, eps_d, phi, ksi, None))
        got_phi, got_ksi = dev.get(phi)[:D, :H, :W], dev.get(ksi)[:D, :H, :W]
        assert bit_same(got_phi, phi_o[:D, :H, :W])
        assert bit_same(got_ksi, ksi_o[:D, :H, :W])

        # 5 ping-pong sweeps from these phi/ksi (tolerance 0)
        du, dv, dw = arrs[5], arrs[6], arrs[7]
        tmp = [dev.out() for _ in range(3)]
        cur = ptr[5:8]
        for it in range(5):
            o = oracle.solve_sweep(*arrs[:5], du, dv, dw, phi_o, ksi_o, dims, h, alpha)
            f3d.check(f3d.hip().f3d_solve_sweep(*ptr[:5], *cur, phi, ksi, W, H, D, *h, alpha, *tmp, None))
            cur, tmp = tmp, cur
            du, dv, dw = o
            if it in (0, 4):
                for g, e in zip(cur, (du, dv, dw)):
                    assert bit_same(dev.get(g)[:D, :H, :W], e[:D, :H, :W]), f"sweep {it}"
    finally:
        dev.close()


@pytest.mark.parametrize("dims,cdims", CASES[:5])
def test_solver_slab_window(f3d, oracle, dims, cdims):
    """A z-slab launch (container plane 0 = global plane z_base) equals the same planes of the whole-volume result."""
    rng = np.random.default_rng(7)
    W, H, D = dims
    h = (1.3, 0.9, 2.0)
    arrs = solver_inputs(rng, dims, cdims)
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, 0.001, 0.001)
    sw_o = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, 7.5)
    z_lo, z_hi = 1, D - 1
    z_base = 0 if z_lo - 1 < 0 else z_lo - 1
    planes = z_hi + 1 - z_base  # halo planes z_lo-1 .. z_hi
    sub = lambda a: np.ascontiguousarray(a[z_base:z_base + planes])
    dev = Dev(f3d, (cdims[0], cdims[1], planes))
    try:
        ptr = [dev.put(sub(a)) for a in arrs]
        slab = f3d.Slab(z_base, z_lo, z_hi)
        phi, ksi = dev.out(), dev.out()
        f3d.check(f3d.hip().f3d_phi_ksi(*ptr, W, H, D, *h, 0.001, 0.001, phi, ksi, C.byref(slab)))
        got = dev.get(phi)
        assert bit_same(got[z_lo - z_base:z_hi - z_base, :H, :W], phi_o[z_lo:z_hi, :H, :W])
        pphi, pksi = dev.put(sub(phi_o)), dev.put(sub(ksi_o))
        outs = [dev.out() for _ in range(3)]
        f3d.check(f3d.hip().f3d_solve_sweep(*ptr, pphi, pksi, W, H, D, *h, 7.5, *outs, C.byref(slab)))
        for g, e in zip(outs, sw_o):
            assert bit_same(dev.get(g)[z_lo - z_base:z_hi - z_base, :H, :W], e[z_lo:z_hi, :H, :W])
    finally:
        dev.close()


@pytest.mark.parametrize("dims,cdims", CASES[:5])
def test_warp(f3d, oracle, dims, cdims):
    rng = np.random.default_rng(11)
    W, H, D = dims
    h = (2.0, 1.0, 0.7)
    f0 = box_in_container(rng, dims, cdims, 0, 255)
    f1 = box_in_container(rng, dims, cdims, 0, 255)
    # flows large enough that ~40 % of targets leave the domain, plus exact-integer landings and NaN
    u = box_in_container(rng, dims, cdims, -0.6 * W, 0.6 * W)
    v = box_in_container(rng, dims, cdims, -0.6 * H, 0.6 * H)
    w = box_in_container(rng, dims, cdims, -0.6 * D, 0.6 * D)
    u[:D, :H, :W][::2, ::3, ::2] = 2.0
    v[:D, :H, :W][::2, ::3, ::2] = -1.0
    w[:D, :H, :W][::2, ::3, ::2] = 0.0
    u[0, 0, 0] = np.nan
    v[D - 1, H - 1, W - 1] = np.nan
    exp = oracle.warp(f0, f1, u, v, w, dims, h)
    dev = Dev(f3d, cdims)
    try:
        p = [dev.put(a) for a in (f0, f1, u, v, w)]
        out = dev.out()
        f3d.check(f3d.hip().f3d_warp(*p, W, H, D, *h, out, None))
        assert bit_same(dev.get(out)[:D, :H, :W], exp[:D, :H, :W])
    finally:
        dev.close()


RESAMPLE = [
    ((37, 21, 9), (36, 20, 9)),      # down ~0.95
    ((36, 20, 9), (37, 21, 9)),      # up
    ((50, 33, 23), (7, 5, 4)),       # strong down, > 7 cells per output
    ((64, 40, 5), (61, 38, 4)),      # thin slab
    ((19, 19, 19), (19, 19, 19)),    # identity
    ((18, 18, 18), (128, 20, 40)),   # strong anisotropic up
]


@pytest.mark.parametrize("src,dst", RESAMPLE)
def test_resample(f3d, oracle, src, dst):
    rng = np.random.default_rng(13)
    cdims = tuple(max(a, b) + 3 for a, b in zip(src, dst))
    inp = box_in_container(rng, src, cdims, -5, 5)
    exp = oracle.resample(inp, src, dst)
    dev = Dev(f3d, cdims)
    try:
        op = f3d.Operation("resample")
        assert op.initialize(dev.cont)
        pin, pout, ptmp = dev.put(inp), dev.out(), dev.out()
        op.execute(dev_input=pin, dev_output=pout, dev_temp=ptmp, data_size=src, resample_size=dst)
        got = dev.get(pout)
        W, H, D = dst
        assert bit_same(got[:D, :H, :W], exp[:D, :H, :W])
        op.destroy()
    finally:
        dev.close()


@pytest.mark.parametrize("r", [3, 5, 7])
@pytest.mark.parametrize("dims,cdims", [((37, 21, 9), (64, 32, 16)), ((9, 6, 4), (9, 6, 4)), ((70, 33, 5), (70, 33, 5)),
                                        ((4, 4, 4), (8, 8, 8))])
def test_median(f3d, oracle, dims, cdims, r):
    rng = np.random.default_rng(17 + r)
    W, H, D = dims
    inp = box_in_container(rng, dims, cdims, -2, 2)
    # plateaus of equal values and exact zeros
    inp[:D, :H, :W][rng.random((D, H, W)) < 0.3] = 0.5
    inp[:D, :H, :W][rng.random((D, H, W)) < 0.1] = 0.0
    exp = oracle.median(inp, dims, r)
    dev = Dev(f3d, cdims)
    try:
        pin, pout = dev.put(inp), dev.out()
        f3d.check(f3d.hip().f3d_median(pin, W, H, D, r, pout, None))
        assert same(dev.get(pout)[:D, :H, :W], exp[:D, :H, :W])
    finally:
        dev.close()


@pytest.mark.parametrize("variant", ["0", "1", "2"])
@pytest.mark.parametrize("dims,window", [((70, 9, 23), None), ((37, 21, 9), None), ((130, 6, 14), None), ((9, 6, 4), None),
                                         ((66, 10, 31), (7, 24)), ((20, 12, 17), (0, 5)), ((20, 12, 17), (12, 17))])
def test_median_kernel_variants(f3d, oracle, dims, window, variant, monkeypatch):
    """The three 5^3 kernels -- one output per step, two outputs sharing the network of the four common planes, sorted planes kept
    in registers (three steps unrolled, odd plane counts, chunks) -- give the oracle's values, also on a slab window whose
    container holds nothing beyond the two halo planes."""
    monkeypatch.setenv("F3D_MEDIAN_PAIR", variant)
    rng = np.random.default_rng(23)
    W, H, D = dims
    inp = box_in_container(rng, dims, dims, -2, 2)
    inp[rng.random((D, H, W)) < 0.3] = 0.5
    inp[rng.random((D, H, W)) < 0.1] = 0.0
    exp = oracle.median(inp, dims, 5)
    z_lo, z_hi = window or (0, D)
    z_base, top = max(0, z_lo - 2), min(D, z_hi + 2)
    dev = Dev(f3d, (W, H, top - z_base))
    try:
        pin, pout = dev.put(np.ascontiguousarray(inp[z_base:top])), dev.out()
        slab = f3d.Slab(z_base, z_lo, z_hi)
        f3d.check(f3d.hip().f3d_median(pin, W, H, D, 5, pout, C.byref(slab)))
        assert same(dev.get(pout)[z_lo - z_base:z_hi - z_base], exp[z_lo:z_hi])
    finally:
        dev.close()


@pytest.mark.parametrize("sigma", [1.0, 2.0, 3.5, 5.0, 8.0])
@pytest.mark.parametrize("dims", [(37, 20, 9), (64, 8, 5), (130, 12, 33)])
def test_gaussian(f3d, oracle, dims, sigma):
    """Clean zero-padded spec; the container height equals the data height as at the reference's only call site."""
    rng = np.random.default_rng(19)
    W, H, D = dims
    cdims = (W + 5, H, D)
    inp = box_in_container(rng, dims, cdims, 0, 255)
    exp = oracle.gaussian(inp, dims, sigma)
    r_o, taps_o = oracle.gaussian_taps(sigma)
    r_p, taps_p = f3d.gaussian_taps(sigma)
    assert r_o == r_p and bit_same(taps_o, taps_p)
    dev = Dev(f3d, cdims)
    try:
        op = f3d.Operation("convolution")
        assert op.initialize(dev.cont)
        pin, pout, ptmp = dev.put(inp), dev.out(), dev.out()
        op.execute(dev_input=pin, dev_output=pout, dev_temp=ptmp, data_size=dims, gaussian_sigma=sigma)
        assert bit_same(dev.get(pout)[:D, :H, :W], exp[:D, :H, :W])
        op.destroy()
        # the launchers one by one: single passes (f3d_conv_rows, f3d_conv_cols), the fused rows + columns launch and the
        # z march give the same bits as the oracle's three passes, also on a window of planes
        hip = f3d.hip()
        taps = np.ascontiguousarray(taps_p, np.float32)
        f3d.check(hip.f3d_set_conv_taps(taps.ctypes.data_as(C.POINTER(C.c_float)), len(taps)))
        g = oracle.geom(inp, z_hi=D)
        ex = np.full_like(inp, np.nan)
        oracle.conv_axis(ex, inp, dims, r_o, taps_o, 0)
        exy = np.full_like(inp, np.nan)
        oracle.conv_axis(exy, ex, dims, r_o, taps_o, 1)
        a, b = dev.out(), dev.out()
        f3d.check(hip.f3d_conv_rows(a, pin, W, H, D, r_p, None))
        assert bit_same(dev.get(a)[:D, :H, :W], ex[:D, :H, :W])
        f3d.check(hip.f3d_conv_cols(b, a, W, H, D, r_p, None))
        assert bit_same(dev.get(b)[:D, :H, :W], exy[:D, :H, :W])
        f3d.check(hip.f3d_conv_rows_cols(a, pin, W, H, D, r_p, None))
        assert bit_same(dev.get(a)[:D, :H, :W], exy[:D, :H, :W])
        if D > 2:
            slab = f3d.Slab(0, 1, D - 1)
            f3d.check(hip.f3d_memset2d(b, dev.cont.pitch, 0xFF, dev.cont.pitch, cdims[1] * cdims[2]))
            f3d.check(hip.f3d_conv_slices(b, a, W, H, D, r_p, C.byref(slab)))
            got = dev.get(b)
            assert bit_same(got[1:D - 1, :H, :W], exp[1:D - 1, :H, :W]) and np.isnan(got[0]).all() and np.isnan(got[D - 1]).all()
        assert hip.f3d_conv_rows_cols(a, a, W, H, D, r_p, None) != 0
    finally:
        dev.close()


def test_add(f3d, oracle):
    rng = np.random.default_rng(23)
    dims, cdims = (37, 21, 9), (64, 32, 16)
    W, H, D = dims
    a = box_in_container(rng, dims, cdims)
    b = box_in_container(rng, dims, cdims)
    exp = a.copy()
    oracle.add(exp, b, dims)
    dev = Dev(f3d, cdims)
    try:
        pa, pb = dev.put(a), dev.put(b)
        f3d.check(f3d.hip().f3d_add(pa, pb, W, H, D, None))
        got = dev.get(pa)
        assert bit_same(got[:D, :H, :W], exp[:D, :H, :W])
        # nothing outside the box was touched
        assert np.isnan(got[D:]).all() and np.isnan(got[:, H:]).all() and np.isnan(got[:, :, W:]).all()
    finally:
        dev.close()


def test_errors_are_loud(f3d):
    dev = Dev(f3d, (16, 8, 8))
    try:
        p = dev.out()
        assert f3d.hip().f3d_median(p, 8, 8, 8, 5, p, None) != 0           # in == out
        assert f3d.hip().f3d_median(p, 8, 8, 8, 4, dev.out(), None) != 0   # unsupported window
        assert f3d.hip().f3d_add(p, p, 32, 8, 8, None) != 0                # box larger than the container
        assert b"container" in f3d.hip().f3d_last_error()
    finally:
        dev.close()


BIG_CASES = [((200, 45, 40), (256, 48, 40)), ((64, 64, 130), (64, 64, 130)), ((130, 19, 23), (130, 19, 23))]


@pytest.mark.parametrize("dims,cdims", CASES + BIG_CASES)
@pytest.mark.parametrize("h", SPACINGS)
def test_two_fused_sweeps(f3d, oracle, dims, cdims, h):
    """f3d_solve_sweep2 = two f3d_solve_sweep calls with a swap in between, bit for bit (and equal to two oracle sweeps):
    the intermediate du, dv, dw stay on chip, halo rows/columns of the first sweep are recomputed per tile."""
    rng = np.random.default_rng(hash((dims, h, 2)) % 2**32)
    W, H, D = dims
    arrs = solver_inputs(rng, dims, cdims)
    alpha = 7.5
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, 0.001, 0.001)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, alpha)
    s2 = oracle.solve_sweep(*arrs[:5], *s1, phi_o, ksi_o, dims, h, alpha)
    dev = Dev(f3d, cdims)
    try:
        ptr = [dev.put(a) for a in arrs]
        phi, ksi = dev.put(phi_o), dev.put(ksi_o)
        outs = [dev.out() for _ in range(3)]
        f3d.check(f3d.hip().f3d_solve_sweep2(*ptr, phi, ksi, W, H, D, *h, alpha, *outs, None))
        for name, g, e in zip("uvw", outs, s2):
            got = dev.get(g)[:D, :H, :W]
            if not bit_same(got, e[:D, :H, :W]):
                # say which side moved: a second launch and a second oracle pass on the same inputs
                bad = np.argwhere(got.view(np.uint32) != np.ascontiguousarray(e[:D, :H, :W]).view(np.uint32))
                f3d.check(f3d.hip().f3d_solve_sweep2(*ptr, phi, ksi, W, H, D, *h, alpha, *outs, None))
                again = dev.get(g)[:D, :H, :W]
                o1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, alpha)
                o2 = oracle.solve_sweep(*arrs[:5], *o1, phi_o, ksi_o, dims, h, alpha)["uvw".index(name)]
                ee = np.ascontiguousarray(e[:D, :H, :W])
                zs = {int(z): int((bad[:, 0] == z).sum()) for z in sorted(set(bad[:, 0]))}
                vals = [(tuple(int(i) for i in b), float(got[tuple(b)]), float(ee[tuple(b)])) for b in bad[:4]]
                raise AssertionError(
                    f"d{name}: {len(bad)} voxels differ, per plane {zs}, NaN in result {int(np.isnan(got).sum())}, samples "
                    f"(index, got, expected) {vals}; second launch equals first: "
                    f"{bit_same(again, got)}, second launch equals oracle: {bit_same(again, e[:D, :H, :W])}, "
                    f"second oracle pass equals first: {bit_same(o2[:D, :H, :W], e[:D, :H, :W])}")
    finally:
        dev.close()


# widths around the edges of k_tri's tiles (64 lanes, 56 owned columns, tile column t > 0 starts at 56 t - 4), rows around its 4- and
# 7-row tiles, depths from 2 planes (every plane a face) up
TRI_CASES = CASES + BIG_CASES + [
    ((56, 7, 6), (64, 8, 6)), ((57, 14, 5), (64, 16, 6)), ((60, 9, 4), (64, 12, 4)), ((61, 3, 3), (64, 4, 4)),
    ((112, 8, 5), (128, 8, 5)), ((113, 11, 4), (128, 12, 4)), ((116, 5, 7), (128, 8, 8)), ((120, 29, 3), (128, 32, 4)),
    ((168, 6, 2), (192, 8, 2)), ((24, 24, 24), (128, 128, 128)), ((18, 18, 18), (64, 20, 20)), ((100, 66, 17), (128, 70, 20)),
]


@pytest.mark.parametrize("ty", ["4", "7", None])
@pytest.mark.parametrize("dims,cdims", TRI_CASES)
def test_three_stage_launches(f3d, oracle, monkeypatch, dims, cdims, ty):
    """f3d_solve_sweep3 = three f3d_solve_sweep calls with the swaps in between, and f3d_solve_sweep2_phi_ksi = two sweeps followed by
    f3d_phi_ksi on their result -- bit for bit, against the oracle's separate passes.  k_tri computes the three columns either side of a
    tile's 56 owned columns and the two rows above and below its owned rows redundantly; widths and heights around those edges, both
    tile heights (F3D_TRI_TY) and the launcher's own choice."""
    if ty:
        monkeypatch.setenv("F3D_TRI_TY", ty)
    else:
        monkeypatch.delenv("F3D_TRI_TY", raising=False)
    W, H, D = dims
    for h in SPACINGS:
        rng = np.random.default_rng(hash((dims, h, 33)) % 2**32)
        arrs = solver_inputs(rng, dims, cdims)
        alpha, eps_s, eps_d = 7.5, 0.001, 0.002
        phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, eps_s, eps_d)
        s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, alpha)
        s2 = oracle.solve_sweep(*arrs[:5], *s1, phi_o, ksi_o, dims, h, alpha)
        s3 = oracle.solve_sweep(*arrs[:5], *s2, phi_o, ksi_o, dims, h, alpha)
        phi_n, ksi_n = oracle.phi_ksi(*arrs[:5], *s2, dims, h, eps_s, eps_d)
        dev = Dev(f3d, cdims)
        try:
            ptr = [dev.put(a) for a in arrs]
            phi, ksi = dev.put(phi_o), dev.put(ksi_o)
            outs = [dev.out() for _ in range(5)]
            f3d.check(f3d.hip().f3d_solve_sweep3(*ptr, phi, ksi, W, H, D, *h, alpha, *outs[:3], None))
            for name, g, e in zip(("du", "dv", "dw"), outs, s3):
                got = dev.get(g)[:D, :H, :W]
                bad = np.argwhere(got.view(np.uint32) != np.ascontiguousarray(e[:D, :H, :W]).view(np.uint32))
                assert len(bad) == 0, (f"three sweeps, {name}, h {h}: {len(bad)} voxels differ; first (z, y, x) {bad[:6].tolist()}, "
                                       f"x range {bad[:, 2].min()}..{bad[:, 2].max()}, y {bad[:, 1].min()}..{bad[:, 1].max()}, "
                                       f"z {bad[:, 0].min()}..{bad[:, 0].max()}")
            outs = [dev.out() for _ in range(5)]
            f3d.check(f3d.hip().f3d_solve_sweep2_phi_ksi(*ptr, phi, ksi, W, H, D, *h, alpha, eps_s, eps_d, *outs, None))
            for name, g, e in zip(("du", "dv", "dw", "phi", "ksi"), outs, list(s2) + [phi_n, ksi_n]):
                got = dev.get(g)[:D, :H, :W]
                bad = np.argwhere(got.view(np.uint32) != np.ascontiguousarray(e[:D, :H, :W]).view(np.uint32))
                assert len(bad) == 0, (f"two sweeps + phi/ksi, {name}, h {h}: {len(bad)} voxels differ; first (z, y, x) {bad[:6].tolist()}, "
                                       f"x range {bad[:, 2].min()}..{bad[:, 2].max()}, y {bad[:, 1].min()}..{bad[:, 1].max()}, "
                                       f"z {bad[:, 0].min()}..{bad[:, 0].max()}")
            # an output that is also an input is refused (other tiles would still be reading it)
            assert f3d.hip().f3d_solve_sweep3(*ptr, phi, ksi, W, H, D, *h, alpha, ptr[5], *outs[1:3], None) != 0
            assert f3d.hip().f3d_solve_sweep2_phi_ksi(*ptr, phi, ksi, W, H, D, *h, alpha, eps_s, eps_d, *outs[:3], phi, outs[4], None) != 0
        finally:
            dev.close()


@pytest.mark.parametrize("dims,cdims", [CASES[0], CASES[2], CASES[4], BIG_CASES[0], ((113, 11, 14), (128, 12, 16))])
@pytest.mark.parametrize("zchunk", ["1", "2", "5", None])
def test_three_stage_launches_on_slab_windows_and_forced_chunks(f3d, oracle, monkeypatch, dims, cdims, zchunk):
    """k_tri on a z window [z_lo, z_hi) with three halo planes either side inside the container, and with the z-chunk pinned (one
    plane per chunk: every chunk is all prologue and tail; F3D_ZCHUNK is read once per process, so the pinned cases run in a child)"""
    rng = np.random.default_rng(17)
    W, H, D = dims
    h = (1.3, 0.9, 2.0)
    arrs = solver_inputs(rng, dims, cdims)
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, 0.001, 0.001)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, 7.5)
    s2 = oracle.solve_sweep(*arrs[:5], *s1, phi_o, ksi_o, dims, h, 7.5)
    s3 = oracle.solve_sweep(*arrs[:5], *s2, phi_o, ksi_o, dims, h, 7.5)
    phi_n, ksi_n = oracle.phi_ksi(*arrs[:5], *s2, dims, h, 0.001, 0.001)
    if zchunk is not None:
        # the tuning switches are read once per process: run this case in a child process that pins the chunk
        import pickle, subprocess, sys, tempfile, os
        with tempfile.TemporaryDirectory() as tmp:
            blob = os.path.join(tmp, "case.pkl")
            pickle.dump(dict(arrs=arrs, phi=phi_o, ksi=ksi_o, dims=dims, cdims=cdims, h=h, s3=s3, s2=s2, phi_n=phi_n, ksi_n=ksi_n), open(blob, "wb"))
            code = ("import importlib, pickle, sys, numpy as np\n"
                    f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
                    f"sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})\n"
                    "f3d = importlib.import_module('cuda-flow3d_amd')\n"
                    "from test_gpu_kernels import Dev\n"
                    f"c = pickle.load(open({blob!r}, 'rb'))\n"
                    "W, H, D = c['dims']; dev = Dev(f3d, c['cdims'])\n"
                    "ptr = [dev.put(a) for a in c['arrs']]; phi, ksi = dev.put(c['phi']), dev.put(c['ksi'])\n"
                    "outs = [dev.out() for _ in range(5)]\n"
                    "f3d.check(f3d.hip().f3d_solve_sweep3(*ptr, phi, ksi, W, H, D, *c['h'], 7.5, *outs[:3], None))\n"
                    "same = lambda g, e: np.array_equal(np.ascontiguousarray(dev.get(g)[:D, :H, :W]).view(np.uint32), np.ascontiguousarray(e[:D, :H, :W]).view(np.uint32))\n"
                    "assert all(same(g, e) for g, e in zip(outs, c['s3'])), 'three sweeps'\n"
                    "outs = [dev.out() for _ in range(5)]\n"
                    "f3d.check(f3d.hip().f3d_solve_sweep2_phi_ksi(*ptr, phi, ksi, W, H, D, *c['h'], 7.5, 0.001, 0.001, *outs, None))\n"
                    "assert all(same(g, e) for g, e in zip(outs, list(c['s2']) + [c['phi_n'], c['ksi_n']])), 'two sweeps + phi/ksi'\n"
                    "dev.close(); print('ok')\n")
            out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, F3D_ZCHUNK=zchunk), capture_output=True, text=True, timeout=600)
            assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
        return
    z_lo, z_hi = (1, D - 1) if D < 10 else (3, D - 2)
    z_base = max(0, z_lo - 3)
    top = min(D, z_hi + 3)
    planes = top - z_base
    sub = lambda a: np.ascontiguousarray(a[z_base:top])
    dev = Dev(f3d, (cdims[0], cdims[1], planes))
    try:
        ptr = [dev.put(sub(a)) for a in arrs] + [dev.put(sub(phi_o)), dev.put(sub(ksi_o))]
        slab = f3d.Slab(z_base, z_lo, z_hi)
        outs = [dev.out() for _ in range(5)]
        f3d.check(f3d.hip().f3d_solve_sweep3(*ptr, W, H, D, *h, 7.5, *outs[:3], C.byref(slab)))
        for g, e in zip(outs, s3):
            assert bit_same(dev.get(g)[z_lo - z_base:z_hi - z_base, :H, :W], e[z_lo:z_hi, :H, :W])
        outs = [dev.out() for _ in range(5)]
        f3d.check(f3d.hip().f3d_solve_sweep2_phi_ksi(*ptr, W, H, D, *h, 7.5, 0.001, 0.001, *outs, C.byref(slab)))
        for g, e in zip(outs, list(s2) + [phi_n, ksi_n]):
            assert bit_same(dev.get(g)[z_lo - z_base:z_hi - z_base, :H, :W], e[z_lo:z_hi, :H, :W])
    finally:
        dev.close()


@pytest.mark.parametrize("dims,cdims", CASES[:5] + BIG_CASES[:1])
def test_two_fused_sweeps_slab_window(f3d, oracle, dims, cdims):
    """Slab launch of the fused pair: window [z_lo, z_hi) with two halo planes on either side inside the container."""
    rng = np.random.default_rng(11)
    W, H, D = dims
    h = (1.3, 0.9, 2.0)
    arrs = solver_inputs(rng, dims, cdims)
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, 0.001, 0.001)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, 7.5)
    s2 = oracle.solve_sweep(*arrs[:5], *s1, phi_o, ksi_o, dims, h, 7.5)
    z_lo, z_hi = (1, D - 1) if D < 10 else (3, D - 2)
    z_base = max(0, z_lo - 2)
    top = min(D, z_hi + 2)
    planes = top - z_base
    sub = lambda a: np.ascontiguousarray(a[z_base:top])
    dev = Dev(f3d, (cdims[0], cdims[1], planes))
    try:
        ptr = [dev.put(sub(a)) for a in arrs] + [dev.put(sub(phi_o)), dev.put(sub(ksi_o))]
        outs = [dev.out() for _ in range(3)]
        slab = f3d.Slab(z_base, z_lo, z_hi)
        f3d.check(f3d.hip().f3d_solve_sweep2(*ptr, W, H, D, *h, 7.5, *outs, C.byref(slab)))
        for g, e in zip(outs, s2):
            assert bit_same(dev.get(g)[z_lo - z_base:z_hi - z_base, :H, :W], e[z_lo:z_hi, :H, :W])
    finally:
        dev.close()


@pytest.mark.parametrize("dims,cdims", CASES + BIG_CASES)
@pytest.mark.parametrize("h", SPACINGS)
def test_sweep_and_next_phi_ksi_fused(f3d, oracle, dims, cdims, h):
    """f3d_solve_sweep_phi_ksi = f3d_solve_sweep followed by f3d_phi_ksi on its result, bit for bit (and equal to the oracle's
    sweep + phi/ksi): the increments reach the weights of the next outer iteration without a trip through HBM."""
    rng = np.random.default_rng(hash((dims, h, 3)) % 2**32)
    W, H, D = dims
    arrs = solver_inputs(rng, dims, cdims)
    alpha, eps_s, eps_d = 7.5, 0.001, 0.002
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, eps_s, eps_d)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, alpha)
    phi_n, ksi_n = oracle.phi_ksi(*arrs[:5], *s1, dims, h, eps_s, eps_d)
    dev = Dev(f3d, cdims)
    try:
        ptr = [dev.put(a) for a in arrs]
        phi, ksi = dev.put(phi_o), dev.put(ksi_o)
        outs = [dev.out() for _ in range(5)]
        f3d.check(f3d.hip().f3d_solve_sweep_phi_ksi(*ptr, phi, ksi, W, H, D, *h, alpha, eps_s, eps_d, *outs, None))
        for name, g, e in zip(("du", "dv", "dw", "phi", "ksi"), outs, list(s1) + [phi_n, ksi_n]):
            got = dev.get(g)[:D, :H, :W]
            assert bit_same(got, e[:D, :H, :W]), \
                f"{name}: {np.count_nonzero(got.view(np.uint32) != np.ascontiguousarray(e[:D, :H, :W]).view(np.uint32))} voxels differ"
        # the weights must go to buffers of their own
        assert f3d.hip().f3d_solve_sweep_phi_ksi(*ptr, phi, ksi, W, H, D, *h, alpha, eps_s, eps_d, *outs[:3], phi, outs[4], None) != 0
        assert b"alias" in f3d.hip().f3d_last_error()
    finally:
        dev.close()


@pytest.mark.parametrize("dims,cdims", CASES + BIG_CASES)
@pytest.mark.parametrize("h", SPACINGS)
def test_fused_launches_on_precomputed_frame_derivatives(f3d, oracle, dims, cdims, h):
    """f3d_frame_derivatives once, then f3d_solve_sweep2_fd / f3d_solve_sweep_phi_ksi_fd: the same bits as the launches that
    read the frames (and as the oracle).  The derivative volumes are checked on their own against numpy with the reference's
    association order."""
    rng = np.random.default_rng(hash((dims, h, 4)) % 2**32)
    W, H, D = dims
    arrs = solver_inputs(rng, dims, cdims)
    alpha, eps_s, eps_d = 7.5, 0.001, 0.002
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, eps_s, eps_d)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, alpha)
    s2 = oracle.solve_sweep(*arrs[:5], *s1, phi_o, ksi_o, dims, h, alpha)
    phi_n, ksi_n = oracle.phi_ksi(*arrs[:5], *s1, dims, h, eps_s, eps_d)
    # expected derivatives: mirrored neighbours, ((F0+ - F0-) + F1+) - F1-, one float32 division by 4h
    f0, f1 = arrs[0][:D, :H, :W], arrs[1][:D, :H, :W]
    def mirrored(a, axis, off):
        n = a.shape[axis]
        idx = np.arange(n) + off
        idx = np.where(idx < 0, -idx, np.where(idx >= n, 2 * n - idx - 2, idx))
        return np.take(a, idx, axis=axis)
    def deriv(axis, hh):
        num = ((mirrored(f0, axis, 1) - mirrored(f0, axis, -1)) + mirrored(f1, axis, 1)) - mirrored(f1, axis, -1)
        return (num / (np.float32(4.0) * np.float32(hh))).astype(np.float32)
    exp_d = [deriv(2, h[0]), deriv(1, h[1]), deriv(0, h[2]), (f1 - f0).astype(np.float32)]
    dev = Dev(f3d, cdims)
    try:
        ptr = [dev.put(a) for a in arrs]
        phi, ksi = dev.put(phi_o), dev.put(ksi_o)
        fd = [dev.out() for _ in range(4)]
        f3d.check(f3d.hip().f3d_frame_derivatives(ptr[0], ptr[1], W, H, D, *h, *fd, None))
        for name, g, e in zip(("fx", "fy", "fz", "ft"), fd, exp_d):
            assert bit_same(dev.get(g)[:D, :H, :W], e), name
        outs = [dev.out() for _ in range(5)]
        f3d.check(f3d.hip().f3d_solve_sweep2_fd(*fd, *ptr[2:], phi, ksi, W, H, D, *h, alpha, *outs[:3], None))
        for name, g, e in zip("uvw", outs, s2):
            assert bit_same(dev.get(g)[:D, :H, :W], e[:D, :H, :W]), f"two sweeps on derivatives: d{name}"
        f3d.check(f3d.hip().f3d_solve_sweep_phi_ksi_fd(*fd, *ptr[2:], phi, ksi, W, H, D, *h, alpha, eps_s, eps_d, *outs, None))
        for name, g, e in zip(("du", "dv", "dw", "phi", "ksi"), outs, list(s1) + [phi_n, ksi_n]):
            assert bit_same(dev.get(g)[:D, :H, :W], e[:D, :H, :W]), f"sweep + phi/ksi on derivatives: {name}"
    finally:
        dev.close()


@pytest.mark.parametrize("dims,cdims", CASES[:5] + BIG_CASES[:1])
def test_sweep_and_next_phi_ksi_slab_window(f3d, oracle, dims, cdims):
    """Slab launch of the fused sweep + phi/ksi: window [z_lo, z_hi) with two halo planes on either side inside the container."""
    rng = np.random.default_rng(12)
    W, H, D = dims
    h = (1.3, 0.9, 2.0)
    arrs = solver_inputs(rng, dims, cdims)
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, 0.001, 0.001)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, 7.5)
    phi_n, ksi_n = oracle.phi_ksi(*arrs[:5], *s1, dims, h, 0.001, 0.001)
    z_lo, z_hi = (1, D - 1) if D < 10 else (3, D - 2)
    z_base = max(0, z_lo - 2)
    top = min(D, z_hi + 2)
    planes = top - z_base
    sub = lambda a: np.ascontiguousarray(a[z_base:top])
    dev = Dev(f3d, (cdims[0], cdims[1], planes))
    try:
        ptr = [dev.put(sub(a)) for a in arrs] + [dev.put(sub(phi_o)), dev.put(sub(ksi_o))]
        outs = [dev.out() for _ in range(5)]
        slab = f3d.Slab(z_base, z_lo, z_hi)
        f3d.check(f3d.hip().f3d_solve_sweep_phi_ksi(*ptr, W, H, D, *h, 7.5, 0.001, 0.001, *outs, C.byref(slab)))
        for g, e in zip(outs, list(s1) + [phi_n, ksi_n]):
            assert bit_same(dev.get(g)[z_lo - z_base:z_hi - z_base, :H, :W], e[z_lo:z_hi, :H, :W])
    finally:
        dev.close()


@pytest.mark.parametrize("dims,cdims,zones", [((37, 20, 19), (64, 32, 19), ((0, 6), (13, 19))), ((129, 10, 12), (192, 12, 12), ((1, 3), (7, 11))),
                                              ((70, 70, 30), (128, 72, 30), ((2, 8), (20, 26))), ((64, 8, 9), (64, 8, 9), ((0, 1), (8, 9)))])
def test_phi_ksi_on_two_zones_in_one_launch(f3d, oracle, dims, cdims, zones):
    """f3d_phi_ksi_zones = f3d_phi_ksi on each of two disjoint windows of one container; planes outside both stay untouched."""
    rng = np.random.default_rng(14)
    W, H, D = dims
    h = (1.3, 0.9, 2.0)
    arrs = solver_inputs(rng, dims, cdims)
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, 0.001, 0.001)
    dev = Dev(f3d, cdims)
    try:
        ptr = [dev.put(a) for a in arrs]
        phi, ksi = dev.out(), dev.out()
        za, zb = f3d.Slab(0, *zones[0]), f3d.Slab(0, *zones[1])
        f3d.check(f3d.hip().f3d_phi_ksi_zones(*ptr, W, H, D, *h, 0.001, 0.001, phi, ksi, C.byref(za), C.byref(zb)))
        for got, exp in ((dev.get(phi), phi_o), (dev.get(ksi), ksi_o)):
            inside = np.zeros(D, bool)
            for lo, hi in zones:
                inside[lo:hi] = True
                assert bit_same(got[lo:hi, :H, :W], exp[lo:hi, :H, :W])
            assert np.isnan(got[:D][~inside]).all()
        # an empty window: the ordinary launch on the other one
        phi2 = dev.out()
        empty = f3d.Slab(0, 3, 3)
        f3d.check(f3d.hip().f3d_phi_ksi_zones(*ptr, W, H, D, *h, 0.001, 0.001, phi2, ksi, C.byref(empty), C.byref(zb)))
        assert bit_same(dev.get(phi2)[zones[1][0]:zones[1][1], :H, :W], phi_o[zones[1][0]:zones[1][1], :H, :W])
        # overlapping windows are refused
        bad = f3d.Slab(0, zones[1][0] - 1 if zones[1][0] > 0 else 0, zones[1][1])
        assert f3d.hip().f3d_phi_ksi_zones(*ptr, W, H, D, *h, 0.001, 0.001, phi2, ksi, C.byref(bad), C.byref(zb)) != 0
    finally:
        dev.close()


@pytest.mark.parametrize("keep", [(1, 1), (1, 0), (0, 1)])
@pytest.mark.parametrize("dims,cdims,window", [((37, 20, 9), (64, 32, 16), (2, 7)), ((129, 10, 9), (192, 12, 9), (0, 6)),
                                               ((65, 6, 5), (128, 8, 8), (2, 5)), ((64, 8, 5), (64, 8, 8), (1, 4))])
def test_sweep_and_next_phi_ksi_keeps_the_edge_planes(f3d, oracle, dims, cdims, window, keep):
    """f3d_solve_sweep_phi_ksi_edges: the weights on [z_lo, z_hi) as before, the sweep ALSO on plane z_lo-1 / z_hi where asked for
    and where such a plane exists (the z-slab driver launches it on [own.lo+1, own.hi-1)); nothing else is written."""
    rng = np.random.default_rng(13)
    W, H, D = dims
    h = (1.3, 0.9, 2.0)
    arrs = solver_inputs(rng, dims, cdims)
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, 0.001, 0.001)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, 7.5)
    phi_n, ksi_n = oracle.phi_ksi(*arrs[:5], *s1, dims, h, 0.001, 0.001)
    z_lo, z_hi = window
    z_base, top = max(0, z_lo - 2), min(D, z_hi + 2)
    sub = lambda a: np.ascontiguousarray(a[z_base:top])
    dev = Dev(f3d, (cdims[0], cdims[1], top - z_base))
    try:
        ptr = [dev.put(sub(a)) for a in arrs] + [dev.put(sub(phi_o)), dev.put(sub(ksi_o))]
        outs = [dev.out() for _ in range(5)]
        slab = f3d.Slab(z_base, z_lo, z_hi)
        f3d.check(f3d.hip().f3d_solve_sweep_phi_ksi_edges(*ptr, W, H, D, *h, 7.5, 0.001, 0.001, *outs, C.byref(slab), *keep))
        s_lo = z_lo - (1 if keep[0] and z_lo > 0 else 0)
        s_hi = z_hi + (1 if keep[1] and z_hi < D else 0)
        for g, e in zip(outs[:3], s1):
            got = dev.get(g)
            assert bit_same(got[s_lo - z_base:s_hi - z_base, :H, :W], e[s_lo:s_hi, :H, :W])
            assert np.isnan(got[:s_lo - z_base]).all() and np.isnan(got[s_hi - z_base:]).all()  # Dev.out() poisons with NaN
        for g, e in zip(outs[3:], (phi_n, ksi_n)):
            got = dev.get(g)
            assert bit_same(got[z_lo - z_base:z_hi - z_base, :H, :W], e[z_lo:z_hi, :H, :W])
            assert np.isnan(got[:z_lo - z_base]).all() and np.isnan(got[z_hi - z_base:]).all()
    finally:
        dev.close()


@pytest.mark.parametrize("value", [0, 0xFF, 0x3C])
def test_memset2d_sub_box(f3d, value):
    """f3d_memset2d (cuMemsetD2D8 of optical_flow_e.cpp:305-310): width_bytes of every row set, the rest of the pitch and
    the rows beyond untouched."""
    W, H, D = 70, 9, 5
    box = f3d.Containers(W, H, D)
    try:
        rng = np.random.default_rng(3)
        ref = rng.standard_normal((D, H, W)).astype(np.float32)
        p = box.new(ref)
        box.set_current()
        w_set, rows = 37, H * 3 + 2          # 37 floats of the first 29 rows
        f3d.check(f3d.hip().f3d_memset2d(p, box.pitch, value, w_set * 4, rows))
        f3d.sync()
        got = box.download(p, (W, H, D)).reshape(D * H, W)
        exp = ref.reshape(D * H, W).copy()
        exp[:rows, :w_set] = np.frombuffer(bytes([value]) * 4, np.float32)[0]
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    finally:
        box.free()


@pytest.mark.parametrize("dims,cdims", CASES[:4])
def test_flow_stats(f3d, oracle, dims, cdims):
    """f3d_flow_stats / the "stat" operator: min and max of |(u,v,w)| equal the reference's host loop bit for bit; the
    average is the double-precision sum rounded once (the reference's scan-order float sum is only approximated)."""
    rng = np.random.default_rng(5)
    W, H, D = dims
    u, v, w = (box_in_container(rng, dims, cdims, -4, 4) for _ in range(3))
    mn_o, mx_o, avg_o, sum_o = oracle.flow_stats(u, v, w, dims)
    dev = Dev(f3d, cdims)
    try:
        pu, pv, pw = dev.put(u), dev.put(v), dev.put(w)
        mn, mx, s = C.c_float(), C.c_float(), C.c_double()
        f3d.check(f3d.hip().f3d_flow_stats(pu, pv, pw, W, H, D, None, C.byref(mn), C.byref(mx), C.byref(s)))
        assert (mn.value, mx.value) == (mn_o, mx_o)
        assert abs(s.value - sum_o) <= 1e-12 * sum_o
        op = f3d.Operation("stat")
        assert op.name == "CUDA Stat" and op.initialize(dev.cont)
        st = f3d.Stat3()
        op.execute(dev_flow_u=pu, dev_flow_v=pv, dev_flow_w=pw, data_size=dims, stat=st)
        assert (st.min, st.max) == (mn_o, mx_o)
        # the reference adds up to 343 000 floats in scan order: its own rounding error is ~1e-5 here (and grows with the
        # volume); the device average is the double sum rounded once
        assert abs(st.avg - avg_o) <= 2e-4 * avg_o
        assert abs(st.avg - np.float32(sum_o / (W * H * D))) <= 1e-7 * avg_o
        # a slab window
        slab = f3d.Slab(0, 1, D - 1)
        f3d.check(f3d.hip().f3d_flow_stats(pu, pv, pw, W, H, D, C.byref(slab), C.byref(mn), C.byref(mx), C.byref(s)))
        g = oracle.geom(u, z_base=0, z_lo=1, z_hi=D - 1)
        mn2, mx2, _, sum2 = oracle.flow_stats(u, v, w, dims, g=g)
        assert (mn.value, mx.value) == (mn2, mx2) and abs(s.value - sum2) <= 1e-12 * sum2
        op.destroy()
    finally:
        dev.close()


@pytest.mark.parametrize("dims,cdims", CASES[:4])
def test_residual_stats(f3d, oracle, dims, cdims):
    """f3d_residual_stats (sum of squares, sum of absolute values, maximum of warped - frame_0) against the oracle's scan:
    the maximum exactly, the double sums to 1e-12 (the order of a parallel reduction differs from the scan's)."""
    rng = np.random.default_rng(6)
    W, H, D = dims
    f0 = box_in_container(rng, dims, cdims, 0, 255)
    fw = box_in_container(rng, dims, cdims, 0, 255)
    ssq_o, sab_o, mx_o = oracle.residual_stats(f0, fw, dims)
    dev = Dev(f3d, cdims)
    try:
        p0, pw = dev.put(f0), dev.put(fw)
        ssq, sab, mx = C.c_double(), C.c_double(), C.c_float()
        f3d.check(f3d.hip().f3d_residual_stats(p0, pw, W, H, D, None, C.byref(ssq), C.byref(sab), C.byref(mx)))
        assert mx.value == mx_o
        assert abs(ssq.value - ssq_o) <= 1e-12 * ssq_o and abs(sab.value - sab_o) <= 1e-12 * sab_o
        slab = f3d.Slab(0, 1, D - 1)
        f3d.check(f3d.hip().f3d_residual_stats(p0, pw, W, H, D, C.byref(slab), C.byref(ssq), C.byref(sab), C.byref(mx)))
        ssq2, sab2, mx2 = oracle.residual_stats(f0, fw, dims, g=oracle.geom(f0, z_base=0, z_lo=1, z_hi=D - 1))
        assert mx.value == mx2 and abs(ssq.value - ssq2) <= 1e-12 * ssq2 and abs(sab.value - sab2) <= 1e-12 * sab2
    finally:
        dev.close()


@pytest.mark.parametrize("scale", [1e-33, 1e-38, 3e-42])
def test_phi_ksi_and_fused_sweeps_tiny_numerators(f3d, oracle, scale):
    """Derivative numerators below 2^-100 (down to subnormal data): the exact-division shortcut by the uniform divisors 2h
    and 4h (f3d_solve.hip, UDiv) must hand such waves to the ordinary IEEE sequence -- quotients in the subnormal range
    can be ties.  Same bits as the oracle, which only ever divides the ordinary way."""
    dims, cdims, h = (70, 21, 9), (128, 24, 9), (7.1, 1.6, 1.25)
    W, H, D = dims
    rng = np.random.default_rng(12)
    arrs = [a * np.float32(scale) for a in solver_inputs(rng, dims, cdims)]
    arrs[2:5] = solver_inputs(rng, dims, cdims)[2:5]   # u, v, w stay O(1): their differences are normal, du's are tiny
    arrs[2][:D, :H, :W] = np.float32(0.75)              # ... and constant, so (u+ - u-) + (du+ - du-) is tiny as well
    arrs[3][:D, :H, :W] = np.float32(-1.5)
    arrs[4][:D, :H, :W] = np.float32(0.25)
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, 0.001, 0.001)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, 7.5)
    s2 = oracle.solve_sweep(*arrs[:5], *s1, phi_o, ksi_o, dims, h, 7.5)
    dev = Dev(f3d, cdims)
    try:
        ptr = [dev.put(a) for a in arrs]
        phi, ksi = dev.out(), dev.out()
        f3d.check(f3d.hip().f3d_phi_ksi(*ptr, W, H, D, *h, 0.001, 0.001, phi, ksi, None))
        assert bit_same(dev.get(phi)[:D, :H, :W], phi_o[:D, :H, :W])
        assert bit_same(dev.get(ksi)[:D, :H, :W], ksi_o[:D, :H, :W])
        outs = [dev.out() for _ in range(3)]
        f3d.check(f3d.hip().f3d_solve_sweep2(*ptr, phi, ksi, W, H, D, *h, 7.5, *outs, None))
        for g, e in zip(outs, s2):
            assert bit_same(dev.get(g)[:D, :H, :W], e[:D, :H, :W])
    finally:
        dev.close()


TINY_CASES = [((2, 2, 2), (64, 4, 4)), ((3, 2, 2), (3, 2, 2)), ((65, 2, 3), (128, 2, 3)), ((2, 10, 2), (2, 10, 2)),
              ((64, 9, 2), (64, 9, 2)), ((129, 19, 3), (192, 20, 3))]


@pytest.mark.parametrize("dims,cdims", TINY_CASES)
def test_solver_kernels_at_minimum_sizes(f3d, oracle, dims, cdims):
    """Every dimension down to 2 (the smallest the mirror rule m(-1) = 1, m(n) = n - 2 allows): phi/ksi, one sweep and the
    fused pair; single planes, single rows, one column past a tile."""
    rng = np.random.default_rng(21)
    W, H, D = dims
    h = (1.7, 0.8, 1.1)
    arrs = solver_inputs(rng, dims, cdims)
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, 0.001, 0.001)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, 7.5)
    s2 = oracle.solve_sweep(*arrs[:5], *s1, phi_o, ksi_o, dims, h, 7.5)
    dev = Dev(f3d, cdims)
    try:
        ptr = [dev.put(a) for a in arrs]
        phi, ksi = dev.out(), dev.out()
        f3d.check(f3d.hip().f3d_phi_ksi(*ptr, W, H, D, *h, 0.001, 0.001, phi, ksi, None))
        assert bit_same(dev.get(phi)[:D, :H, :W], phi_o[:D, :H, :W])
        assert bit_same(dev.get(ksi)[:D, :H, :W], ksi_o[:D, :H, :W])
        o1 = [dev.out() for _ in range(3)]
        o2 = [dev.out() for _ in range(3)]
        f3d.check(f3d.hip().f3d_solve_sweep(*ptr, phi, ksi, W, H, D, *h, 7.5, *o1, None))
        f3d.check(f3d.hip().f3d_solve_sweep2(*ptr, phi, ksi, W, H, D, *h, 7.5, *o2, None))
        for g, e in zip(o1, s1):
            assert bit_same(dev.get(g)[:D, :H, :W], e[:D, :H, :W])
        for g, e in zip(o2, s2):
            assert bit_same(dev.get(g)[:D, :H, :W], e[:D, :H, :W])
    finally:
        dev.close()


def _random_shapes(n, seed):
    """seeded random boxes inside random containers: widths around the 64-lane tile edges, a few rows, a few planes"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        W = int(rng.choice([rng.integers(2, 20), rng.integers(60, 70), rng.integers(120, 135), rng.integers(180, 200)]))
        H = int(rng.integers(2, 30))
        D = int(rng.integers(2, 14))
        cd = (W + int(rng.integers(0, 9)), H + int(rng.integers(0, 4)), D + int(rng.integers(0, 3)))
        h = tuple(float(x) for x in np.round(rng.uniform(0.6, 6.0, 3), 3))
        out.append(((W, H, D), cd, h))
    return out


@pytest.mark.parametrize("dims,cdims,h", _random_shapes(14, 20261004))
def test_solver_launchers_on_random_shapes(f3d, oracle, dims, cdims, h):
    """phi/ksi, one sweep, two fused sweeps and sweep + next phi/ksi on seeded random boxes, containers and grid spacings (the
    fixed CASES above were chosen by hand around known tile edges; these were not chosen at all)."""
    rng = np.random.default_rng(abs(hash((dims, cdims))) % 2**32)
    W, H, D = dims
    eps_s, eps_d, alpha = 0.001, 0.001, 7.5
    arrs = solver_inputs(rng, dims, cdims)
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, eps_s, eps_d)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, alpha)
    s2 = oracle.solve_sweep(*arrs[:5], *s1, phi_o, ksi_o, dims, h, alpha)
    phi_n, ksi_n = oracle.phi_ksi(*arrs[:5], *s1, dims, h, eps_s, eps_d)
    dev = Dev(f3d, cdims)
    hip = f3d.hip()
    box = lambda p: dev.get(p)[:D, :H, :W]
    cut = lambda a: a[:D, :H, :W]
    try:
        ptr = [dev.put(a) for a in arrs]
        phi, ksi = dev.out(), dev.out()
        f3d.check(hip.f3d_phi_ksi(*ptr, W, H, D, *h, eps_s, eps_d, phi, ksi, None))
        assert bit_same(box(phi), cut(phi_o)) and bit_same(box(ksi), cut(ksi_o))
        outs = [dev.out() for _ in range(3)]
        f3d.check(hip.f3d_solve_sweep(*ptr, phi, ksi, W, H, D, *h, alpha, *outs, None))
        for g, e in zip(outs, s1):
            assert bit_same(box(g), cut(e))
        outs = [dev.out() for _ in range(3)]
        f3d.check(hip.f3d_solve_sweep2(*ptr, phi, ksi, W, H, D, *h, alpha, *outs, None))
        for g, e in zip(outs, s2):
            assert bit_same(box(g), cut(e))
        outs = [dev.out() for _ in range(5)]
        f3d.check(hip.f3d_solve_sweep_phi_ksi(*ptr, phi, ksi, W, H, D, *h, alpha, eps_s, eps_d, *outs, None))
        for g, e in zip(outs, list(s1) + [phi_n, ksi_n]):
            assert bit_same(box(g), cut(e))
    finally:
        dev.close()


# thin volumes: the fused launches march along y with all z planes in the tile (k_pair8 with YM, csrc/f3d_solve_pair8.h)
THIN_CASES = [
    ((70, 40, 5), (128, 40, 5)),      # the shape class of BASELINE config 3: five planes, tile height 5
    ((129, 33, 4), (192, 36, 4)),     # four planes (the coarse levels of config 3), W = 64 k + 1, container higher than the level
    ((37, 21, 2), (64, 32, 8)),       # two planes in a deeper container: every z neighbour is a mirror
    ((64, 50, 3), (64, 50, 3)),
    ((66, 19, 6), (128, 19, 6)),      # six, seven, eight planes: tile height 8, rows beyond the volume idle
    ((31, 64, 7), (64, 64, 8)),
    ((200, 17, 8), (256, 20, 8)),
    ((5, 9, 4), (16, 9, 4)),
    ((129, 40, 5), (192, 40, 5)),     # five planes, W = 64 k + 1 (a halo column is the last column of the volume), several y chunks
    ((64, 70, 4), (64, 72, 6)),       # four planes in a deeper container, one tile column exactly
    ((200, 33, 5), (256, 36, 5)),
]


# (march switch, case): "halo-rows" = the y march through the build WITH halo rows, for the depths that have a build without
THIN_RUNS = [(m, c) for m in ("1", "0") for c in THIN_CASES] + [("halo-rows", c) for c in THIN_CASES if c[0][2] in (4, 5)]


@pytest.mark.parametrize("ymarch,case", THIN_RUNS)
@pytest.mark.parametrize("h", SPACINGS)
def test_fused_launches_on_thin_volumes_march_along_y(f3d, oracle, case, h, ymarch, monkeypatch):
    """Two fused sweeps and sweep + next phi/ksi on volumes of 2 ... 8 planes, forced through the y-marching build (F3D_PAIR8_YMARCH=1,
    whatever the H / D ratio) and through the ordinary z march (=0): both equal the oracle bit for bit.  Volumes of exactly four or five
    planes take the tile WITHOUT halo rows by default (k_pair8t: the mirrored neighbour of a face plane is read from the opposite row,
    two workgroups per CU); "halo-rows" runs them through the build with halo rows as well (F3D_PAIR8_TIGHT=0)."""
    dims, cdims = case
    if ymarch == "halo-rows":
        monkeypatch.setenv("F3D_PAIR8_TIGHT", "0")
        ymarch = "1"
    monkeypatch.setenv("F3D_PAIR8_YMARCH", ymarch)
    rng = np.random.default_rng(hash((dims, h, 9)) % 2**32)
    W, H, D = dims
    arrs = solver_inputs(rng, dims, cdims)
    alpha, eps_s, eps_d = 7.5, 0.001, 0.002
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, eps_s, eps_d)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, alpha)
    s2 = oracle.solve_sweep(*arrs[:5], *s1, phi_o, ksi_o, dims, h, alpha)
    phi_n, ksi_n = oracle.phi_ksi(*arrs[:5], *s1, dims, h, eps_s, eps_d)
    dev = Dev(f3d, cdims)
    try:
        ptr = [dev.put(a) for a in arrs]
        phi, ksi = dev.put(phi_o), dev.put(ksi_o)
        outs = [dev.out() for _ in range(5)]
        f3d.check(f3d.hip().f3d_solve_sweep2(*ptr, phi, ksi, W, H, D, *h, alpha, *outs[:3], None))
        for name, g, e in zip(("du", "dv", "dw"), outs, s2):
            got = dev.get(g)[:D, :H, :W]
            assert bit_same(got, e[:D, :H, :W]), \
                f"two sweeps, {name}: {np.count_nonzero(got.view(np.uint32) != np.ascontiguousarray(e[:D, :H, :W]).view(np.uint32))} voxels differ"
        f3d.check(f3d.hip().f3d_solve_sweep_phi_ksi(*ptr, phi, ksi, W, H, D, *h, alpha, eps_s, eps_d, *outs, None))
        for name, g, e in zip(("du", "dv", "dw", "phi", "ksi"), outs, list(s1) + [phi_n, ksi_n]):
            got = dev.get(g)[:D, :H, :W]
            assert bit_same(got, e[:D, :H, :W]), \
                f"sweep + phi/ksi, {name}: {np.count_nonzero(got.view(np.uint32) != np.ascontiguousarray(e[:D, :H, :W]).view(np.uint32))} voxels differ"
    finally:
        dev.close()


def test_the_short_road_to_the_weights_is_exact_for_every_float(f3d):
    """phi = 1 / (2 sqrt(a)) by v_rsq_f32 and four fused multiply-adds (csrc/f3d_solve_pair8.h, weight_fast) against the IEEE
    square root and division the reference's expression compiles to, for ALL 2^32 bit patterns: wherever the kernel's guard
    (weight_fast_ok) lets the short road through, the bits are those of the IEEE chain.  The guard passes the whole range
    2^-100 .. 2^100 except the two arguments per binade whose root has an all-ones significand."""
    hip = f3d.hip()
    checked, excluded, bad = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
    first = C.c_uint()
    f3d.check(hip.f3d_selftest_weights(0, 0xFFFFFFFF, C.byref(checked), C.byref(excluded), C.byref(bad), C.byref(first)))
    assert bad.value == 0, f"{bad.value} arguments differ from the IEEE chain, e.g. bits 0x{first.value:08x}"
    in_range = 0x71800000 - 0x0d800000 + 1
    assert checked.value + excluded.value == 2 ** 32
    assert checked.value == in_range - 200, (checked.value, in_range)    # 100 binades x 2 all-ones roots are left to the IEEE road


@pytest.mark.parametrize("ty", ["4", "8", "12"])
@pytest.mark.parametrize("dims,cdims", [((70, 29, 11), (128, 32, 12)), ((129, 13, 9), (192, 16, 9)), ((40, 50, 6), (64, 50, 6))])
def test_every_tile_height_of_the_fused_kernels(f3d, oracle, dims, cdims, ty, monkeypatch):
    """The launchers pick the tile height (4, 8 or 12 core rows: 8, 12 or 16 waves) per level from a cost model, and on shapes this
    small they never pick twelve.  F3D_PAIR8_TY pins it: every height, frames and frame derivatives (the 12-row frame-derivative build
    keeps its five centre-only inputs in a two-slot ring of their own), rows that do not divide the height, against the oracle."""
    monkeypatch.setenv("F3D_PAIR8_TY", ty)
    rng = np.random.default_rng(hash((dims, ty)) % 2**32)
    W, H, D = dims
    h = (1.3, 0.9, 2.0)
    arrs = solver_inputs(rng, dims, cdims)
    alpha, eps_s, eps_d = 7.5, 0.001, 0.002
    phi_o, ksi_o = oracle.phi_ksi(*arrs, dims, h, eps_s, eps_d)
    s1 = oracle.solve_sweep(*arrs, phi_o, ksi_o, dims, h, alpha)
    s2 = oracle.solve_sweep(*arrs[:5], *s1, phi_o, ksi_o, dims, h, alpha)
    phi_n, ksi_n = oracle.phi_ksi(*arrs[:5], *s1, dims, h, eps_s, eps_d)
    dev = Dev(f3d, cdims)
    hip = f3d.hip()
    box = lambda p: dev.get(p)[:D, :H, :W]
    cut = lambda a: a[:D, :H, :W]
    try:
        ptr = [dev.put(a) for a in arrs]
        phi, ksi = dev.put(phi_o), dev.put(ksi_o)
        fd = [dev.out() for _ in range(4)]
        f3d.check(hip.f3d_frame_derivatives(ptr[0], ptr[1], W, H, D, *h, *fd, None))
        for label, two, one in (("frames", lambda o: hip.f3d_solve_sweep2(*ptr, phi, ksi, W, H, D, *h, alpha, *o, None),
                                 lambda o: hip.f3d_solve_sweep_phi_ksi(*ptr, phi, ksi, W, H, D, *h, alpha, eps_s, eps_d, *o, None)),
                                ("derivatives", lambda o: hip.f3d_solve_sweep2_fd(*fd, *ptr[2:], phi, ksi, W, H, D, *h, alpha, *o, None),
                                 lambda o: hip.f3d_solve_sweep_phi_ksi_fd(*fd, *ptr[2:], phi, ksi, W, H, D, *h, alpha, eps_s, eps_d, *o, None))):
            outs = [dev.out() for _ in range(5)]
            f3d.check(two(outs[:3]))
            for name, g, e in zip("uvw", outs, s2):
                assert bit_same(box(g), cut(e)), f"{ty} rows, {label}, two sweeps: d{name}"
            f3d.check(one(outs))
            for name, g, e in zip(("du", "dv", "dw", "phi", "ksi"), outs, list(s1) + [phi_n, ksi_n]):
                assert bit_same(box(g), cut(e)), f"{ty} rows, {label}, sweep + phi/ksi: {name}"
    finally:
        dev.close()


def _dev_array(ptrs):
    return (C.c_uint64 * len(ptrs))(*ptrs)


@pytest.mark.parametrize("count", [1, 2, 3])
@pytest.mark.parametrize("dims,cdims,window", [((37, 21, 9), (64, 32, 16), None), ((70, 9, 5), (70, 9, 5), None),
                                               ((20, 12, 17), (24, 12, 17), (3, 11)), ((129, 7, 6), (132, 8, 6), None)])
def test_batched_entries_equal_the_single_volume_ones(f3d, oracle, dims, cdims, window, count, monkeypatch):
    """f3d_add_n, f3d_median_n (3, 5 -- every 5^3 kernel -- and 7), f3d_resample_{x,y,z}_n and f3d_clear_box_n on one, two and
    three volumes of a box: what the single-volume entries (pinned to the oracle above) leave, volume by volume, also on a slab
    window; nothing outside the box or the window is written."""
    hip = f3d.hip()
    rng = np.random.default_rng(hash((dims, count)) % 2**32)
    W, H, D = dims
    z_lo, z_hi = window or (0, D)
    slab = C.byref(f3d.Slab(0, z_lo, z_hi)) if window else None
    vols = [box_in_container(rng, dims, cdims, -2, 2) for _ in range(count)]
    incs = [box_in_container(rng, dims, cdims, -1, 1) for _ in range(count)]
    for v in vols:
        v[:D, :H, :W][rng.random((D, H, W)) < 0.2] = 0.5
    dev = Dev(f3d, cdims)
    try:
        # add
        a1 = [dev.put(v) for v in vols]
        a2 = [dev.put(v) for v in vols]
        b = [dev.put(v) for v in incs]
        for i in range(count):
            f3d.check(hip.f3d_add(a1[i], b[i], W, H, D, slab))
        f3d.check(hip.f3d_add_n(_dev_array(a2), _dev_array(b), count, W, H, D, slab))
        for i in range(count):
            one, many = dev.get(a1[i]), dev.get(a2[i])
            assert one.tobytes() == many.tobytes()
            exp = vols[i].copy()
            oracle.add(exp, incs[i], dims)
            assert bit_same(many[z_lo:z_hi, :H, :W], exp[z_lo:z_hi, :H, :W])
        # median: every window, every 5^3 kernel
        for r, variant in ((3, None), (5, "0"), (5, "1"), (5, "2"), (5, None), (7, None)):
            if min(W, H, D) <= r // 2:
                continue
            if variant is None:
                monkeypatch.delenv("F3D_MEDIAN_PAIR", raising=False)
            else:
                monkeypatch.setenv("F3D_MEDIAN_PAIR", variant)
            o1 = [dev.out() for _ in range(count)]
            o2 = [dev.out() for _ in range(count)]
            for i in range(count):
                f3d.check(hip.f3d_median(a1[i], W, H, D, r, o1[i], slab))
            f3d.check(hip.f3d_median_n(_dev_array(a1), count, W, H, D, r, _dev_array(o2), slab))
            for i in range(count):
                assert dev.get(o1[i]).tobytes() == dev.get(o2[i]).tobytes(), (r, variant, i)
        monkeypatch.delenv("F3D_MEDIAN_PAIR", raising=False)
        # clear: the box (window planes) becomes +0, everything else keeps its bits
        before = [dev.get(p) for p in a2]
        f3d.check(hip.f3d_clear_box_n(_dev_array(a2), count, W, H, D, slab))
        for i in range(count):
            got = dev.get(a2[i])
            assert (got[z_lo:z_hi, :H, :W].view(np.uint32) == 0).all()
            keep = before[i].copy()
            keep[z_lo:z_hi, :H, :W] = 0
            assert got.tobytes() == keep.tobytes()
        assert hip.f3d_add_n(_dev_array(a1), _dev_array(b), 0, W, H, D, slab) != 0
        assert hip.f3d_add_n(_dev_array(a1 * 4), _dev_array(b * 4), 4, W, H, D, slab) != 0
        if count > 1:
            o = [dev.out() for _ in range(count)]
            o[1] = a1[0]                                   # an input of the batch as another volume's output
            assert hip.f3d_median_n(_dev_array(a1), count, W, H, D, 3, _dev_array(o), slab) != 0
    finally:
        dev.close()


@pytest.mark.parametrize("count", [2, 3])
@pytest.mark.parametrize("src,dst", [((37, 21, 9), (36, 20, 9)), ((36, 20, 9), (37, 21, 9)), ((50, 33, 23), (7, 5, 4)),
                                     ((18, 18, 18), (128, 20, 40))])
def test_batched_resampling(f3d, oracle, src, dst, count):
    """The resample operator on a batch of bags (three launches for all volumes) against the oracle and against Execute bag by bag;
    a batch whose bags share a temp falls back to one bag after the other and gives the same."""
    rng = np.random.default_rng(hash((src, dst, count)) % 2**32)
    cdims = tuple(max(a, b) + 3 for a, b in zip(src, dst))
    inps = [box_in_container(rng, src, cdims, -5, 5) for _ in range(count)]
    W, H, D = dst
    dev = Dev(f3d, cdims)
    try:
        op = f3d.Operation("resample")
        assert op.initialize(dev.cont)
        pin = [dev.put(x) for x in inps]
        for shared_temp in (False, True):
            pout = [dev.out() for _ in range(count)]
            one = dev.out()
            ptmp = [one if shared_temp else dev.out() for _ in range(count)]
            op.execute_batch([dict(dev_input=pin[i], dev_output=pout[i], dev_temp=ptmp[i], data_size=src, resample_size=dst)
                              for i in range(count)])
            for i in range(count):
                exp = oracle.resample(inps[i], src, dst)
                assert bit_same(dev.get(pout[i])[:D, :H, :W], exp[:D, :H, :W]), (i, shared_temp)
        op.destroy()
    finally:
        dev.close()
