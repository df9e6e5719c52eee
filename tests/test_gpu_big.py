"""Maximum size of the north-star configurations (C5: 1024^3): every container is exactly 4 GiB, so byte
offsets inside one array run up to 2^32 - 4 and anything that keeps them in a signed 32-bit integer breaks in the rear
half of the volume.  The launchers run over the WHOLE volume on the MI355X; the oracle restates three z-windows of it
(front, the 2 GiB crossing, rear) from the same data and the planes must agree bit for bit."""
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
