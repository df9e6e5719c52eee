"""Maximum size of the north-star configurations (C5: 1024^3): every container is exactly 4 GiB, so byte
offsets inside one array run up to 2^32 - 4 and anything that keeps them in a signed 32-bit integer breaks in the rear
half of the volume.  The launchers run over the WHOLE volume on the MI355X; the oracle restates three z-windows of it
(front, the 2 GiB crossing, rear) from the same data and the planes must agree bit for bit."""
import ctypes as C

import numpy as np
import pytest

from conftest import bit_same

pytestmark = pytest.mark.gpu

S = 1024
PERIOD = 16          # the inputs repeat along z with this period: 64 MiB of host data per field instead of 4 GiB
MARGIN = 5           # planes of context around a window (warp reach 4 for |w| <= 3, median 2, solver 2)
WINDOWS = [(0, 3), (509, 515), (S - 3, S)]


def test_full_volume_1024_cubed(f3d, oracle):
    free, _total = f3d.mem_info()
    if free < 64 * 2**30:
        pytest.skip("needs 64 GiB of free device memory")
    rng = np.random.default_rng(1024)
    mk = lambda lo, hi: rng.uniform(lo, hi, (PERIOD, S, S)).astype(np.float32)
    chunk = [mk(0, 255), mk(0, 255), mk(-3, 3), mk(-3, 3), mk(-3, 3), mk(-0.5, 0.5), mk(-0.5, 0.5), mk(-0.5, 0.5)]
    h, eps, alpha = (1.0, 1.0, 1.0), 0.001, 7.5
    hip = f3d.hip()
    box = f3d.Containers(S, S, S)
    try:
        ptr = [box.alloc() for _ in chunk]
        box.set_current()
        for p, c in zip(ptr, chunk):
            for z in range(0, S, PERIOD):
                box.upload(p, c, plane0=z)
        phi, ksi, o_du, o_dv, o_dw, med, wrp = (box.alloc(fill=0xFF) for _ in range(7))
        f3d.check(hip.f3d_phi_ksi(*ptr, S, S, S, *h, eps, eps, phi, ksi, None))
        f3d.check(hip.f3d_solve_sweep(*ptr, phi, ksi, S, S, S, *h, alpha, o_du, o_dv, o_dw, None))
        f3d.check(hip.f3d_median(ptr[2], S, S, S, 5, med, None))
        f3d.check(hip.f3d_warp(ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], S, S, S, *h, wrp, None))
        f3d.sync()

        for a, b in WINDOWS:
            zb, ze = max(0, a - MARGIN), min(S, b + MARGIN)
            sub = [np.ascontiguousarray(np.stack([c[z % PERIOD] for z in range(zb, ze)])) for c in chunk]
            pa, pb = max(0, a - 1), min(S, b + 1)       # phi/ksi one plane around the sweep window
            g_phi = oracle.geom(sub[0], z_base=zb, z_lo=pa, z_hi=pb)
            g_win = oracle.geom(sub[0], z_base=zb, z_lo=a, z_hi=b)
            phi_o, ksi_o = oracle.phi_ksi(*sub, (S, S, S), h, eps, eps, g=g_phi)
            sw_o = oracle.solve_sweep(*sub, phi_o, ksi_o, (S, S, S), h, alpha, g=g_win)
            med_o = oracle.median(sub[2], (S, S, S), 5, g=g_win)
            wrp_o = oracle.warp(*sub[:5], (S, S, S), h, g=g_win)
            rows = slice(a - zb, b - zb)
            get = lambda p: box.download(p, (S, S, b - a), plane0=a)
            assert bit_same(get(phi), phi_o[rows]), f"phi planes [{a},{b})"
            assert bit_same(get(ksi), ksi_o[rows]), f"ksi planes [{a},{b})"
            for name, p, e in zip(("du", "dv", "dw"), (o_du, o_dv, o_dw), sw_o):
                assert bit_same(get(p), e[rows]), f"{name} planes [{a},{b})"
            assert np.array_equal(get(med), med_o[rows]), f"median planes [{a},{b})"
            assert bit_same(get(wrp), wrp_o[rows]), f"warp planes [{a},{b})"
    finally:
        f3d.sync()
        box.free()


def test_level_in_a_container_with_planes_above_4_mib(f3d, oracle):
    """A 1000^3 level in the corner of a 1280 x 1280 container (6.25 MiB planes, the geometry of a 1280^3 run): the solver
    kernels address a z-chunk with 32-bit byte offsets, so no chunk may span 4 GiB of an array -- 655 of these planes -- while
    the chunk rules would otherwise give the fused pair ONE chunk of 1000 planes here (wrong addresses, NaN: found with
    tools/pbench.py --size 1280).  Whole-level launches of phi/ksi, one sweep and the fused pair; the oracle restates
    windows at the front, around plane 655 and at the rear."""
    free, _total = f3d.mem_info()
    if free < 120 * 2**30:
        pytest.skip("needs 120 GiB of free device memory")
    L, CS, period = 1000, 1280, 8
    windows = [(0, 3), (652, 660), (L - 3, L)]
    margin = 6   # the warp reaches ceil(3 / 1.28) + 1 = 4 planes, the median 2, the pair 2
    rng = np.random.default_rng(1280)
    mk = lambda lo, hi: rng.uniform(lo, hi, (period, L, L)).astype(np.float32)
    chunk = [mk(0, 255), mk(0, 255), mk(-3, 3), mk(-3, 3), mk(-3, 3), mk(-0.5, 0.5), mk(-0.5, 0.5), mk(-0.5, 0.5)]
    h, eps, alpha = (1.28, 1.28, 1.28), 0.001, 7.5
    hip = f3d.hip()
    box = f3d.Containers(CS, CS, L)
    try:
        ptr = [box.alloc() for _ in chunk]
        box.set_current()
        for p, c in zip(ptr, chunk):
            for z in range(0, L, period):
                box.upload(p, c, plane0=z)
        phi, ksi, s1u, s1v, s1w, s2u, s2v, s2w, med, wrp, blur, tmp = (box.alloc(fill=0xFF) for _ in range(12))
        f3d.check(hip.f3d_phi_ksi(*ptr, L, L, L, *h, eps, eps, phi, ksi, None))
        f3d.check(hip.f3d_solve_sweep(*ptr, phi, ksi, L, L, L, *h, alpha, s1u, s1v, s1w, None))
        f3d.check(hip.f3d_solve_sweep2(*ptr, phi, ksi, L, L, L, *h, alpha, s2u, s2v, s2w, None))
        f3d.check(hip.f3d_median(ptr[2], L, L, L, 5, med, None))
        f3d.check(hip.f3d_warp(ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], L, L, L, *h, wrp, None))
        radius, taps = f3d.gaussian_taps(2.0)
        f3d.check(hip.f3d_set_conv_taps(taps.ctypes.data_as(C.POINTER(C.c_float)), len(taps)))
        f3d.check(hip.f3d_conv_rows(blur, ptr[0], L, L, L, radius, None))
        f3d.check(hip.f3d_conv_cols(tmp, blur, L, L, L, radius, None))
        f3d.check(hip.f3d_conv_slices(blur, tmp, L, L, L, radius, None))
        f3d.sync()
        for a, b in windows:
            zb, ze = max(0, a - margin), min(L, b + margin)
            sub = [np.ascontiguousarray(np.stack([c[z % period] for z in range(zb, ze)])) for c in chunk]
            wide = (max(0, a - 2), min(L, b + 2))     # phi/ksi and the first sweep of the pair on two planes more
            g = lambda lo, hi: oracle.geom(sub[0], z_base=zb, z_lo=lo, z_hi=hi)
            phi_o, ksi_o = oracle.phi_ksi(*sub, (L, L, L), h, eps, eps, g=g(*wide))
            first = oracle.solve_sweep(*sub, phi_o, ksi_o, (L, L, L), h, alpha, g=g(max(0, a - 1), min(L, b + 1)))
            second = oracle.solve_sweep(*sub[:5], *first, phi_o, ksi_o, (L, L, L), h, alpha, g=g(a, b))
            rows = slice(a - zb, b - zb)
            get = lambda p: box.download(p, (L, L, b - a), plane0=a)
            assert bit_same(get(phi), phi_o[rows]), f"phi planes [{a},{b})"
            for name, p, e in zip(("du", "dv", "dw"), (s1u, s1v, s1w), first):
                assert bit_same(get(p), e[rows]), f"one sweep, {name} planes [{a},{b})"
            for name, p, e in zip(("du", "dv", "dw"), (s2u, s2v, s2w), second):
                assert bit_same(get(p), e[rows]), f"fused pair, {name} planes [{a},{b})"
            assert np.array_equal(get(med), oracle.median(sub[2], (L, L, L), 5, g=g(a, b))[rows]), f"median planes [{a},{b})"
            assert bit_same(get(wrp), oracle.warp(*sub[:5], (L, L, L), h, g=g(a, b))[rows]), f"warp planes [{a},{b})"
            ta, tb = np.full_like(sub[0], np.nan), np.full_like(sub[0], np.nan)
            oracle.conv_axis(ta, sub[0], (L, L, L), radius, taps, 0, g(zb, ze))
            oracle.conv_axis(tb, ta, (L, L, L), radius, taps, 1, g(zb, ze))
            oracle.conv_axis(ta, tb, (L, L, L), radius, taps, 2, g(a, b))
            assert bit_same(get(blur), ta[rows]), f"Gaussian planes [{a},{b})"
    finally:
        f3d.sync()
        box.free()
