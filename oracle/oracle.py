"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/liboracle.so (the CPU restatement, f3d_oracle.c) and, when it was built,
oracle/_ref/libf3d_ref.so (the CUDA-free reference sources compiled in place).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product package
never does.  Volumes are numpy float32 arrays indexed [z, y, x] (x fastest, like the reference's Data3D).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

DEFAULT_PARAMS = dict(
    warp_levels_count=40, warp_scale_factor=0.95, outer_iterations_count=40, inner_iterations_count=5,
    equation_alpha=7.5, equation_smoothness=0.001, equation_data=0.001, median_radius=5, gaussian_sigma=2.0,
)  # src/main.cpp:77-85


class Geom(C.Structure):
    _fields_ = [("Hc", C.c_int), ("pitch_f", C.c_int), ("z_base", C.c_int), ("z_lo", C.c_int), ("z_hi", C.c_int)]


class Params(C.Structure):
    _fields_ = [
        ("warp_levels_count", C.c_size_t), ("warp_scale_factor", C.c_float),
        ("outer_iterations_count", C.c_size_t), ("inner_iterations_count", C.c_size_t),
        ("equation_alpha", C.c_float), ("equation_smoothness", C.c_float), ("equation_data", C.c_float),
        ("median_radius", C.c_size_t), ("gaussian_sigma", C.c_float),
    ]


def build(force=False):
    so = os.path.join(HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "f3d_oracle.c")):
        subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)
    return so


def cpu_share():
    """Threads to use: the process's CPU affinity, capped at 16 (the GPU box's share per GPU)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def lib():
    global _LIB
    if _LIB is None:
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # no spinning if the box is oversubscribed
        os.environ.setdefault("OMP_NUM_THREADS", str(cpu_share()))
        L = C.CDLL(build())
        fp = C.POINTER(C.c_float)
        gp = C.POINTER(Geom)
        L.orc_max_warp_level.restype = C.c_size_t
        L.orc_max_warp_level.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_float]
        L.orc_level_geometry.restype = None
        L.orc_level_geometry.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, C.c_int] + \
            [C.POINTER(C.c_size_t)] * 3 + [fp] * 3
        L.orc_gaussian_taps.restype = C.c_int
        L.orc_gaussian_taps.argtypes = [C.c_float, fp, C.c_int]
        L.orc_conv_axis.restype = None
        L.orc_conv_axis.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, gp]
        L.orc_resample_axis.restype = None
        L.orc_resample_axis.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, gp, gp]
        L.orc_warp.restype = None
        L.orc_warp.argtypes = [fp] * 5 + [C.c_int] * 3 + [C.c_float] * 3 + [fp, gp]
        L.orc_phi_ksi.restype = None
        L.orc_phi_ksi.argtypes = [fp] * 8 + [C.c_int] * 3 + [C.c_float] * 5 + [fp, fp, gp]
        L.orc_solve_sweep.restype = None
        L.orc_solve_sweep.argtypes = [fp] * 10 + [C.c_int] * 3 + [C.c_float] * 4 + [fp] * 3 + [gp]
        L.orc_flow_stats.restype = None
        L.orc_flow_stats.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, gp, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.orc_residual_stats.restype = None
        L.orc_residual_stats.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, gp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                         C.POINTER(C.c_float)]
        L.orc_add.restype = None
        L.orc_add.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, gp]
        L.orc_median.restype = None
        L.orc_median.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, gp]
        L.orc_compute_flow.restype = C.c_int
        L.orc_compute_flow.argtypes = [fp, fp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(Params), C.c_int, fp, fp, fp]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads.restype = None
        L.orc_set_threads(cpu_share())
        _LIB = L
    return _LIB


def ref():
    """The compiled-in-place reference bridge, or None when oracle/_ref was not built."""
    global _REF
    if _REF is None:
        so = os.path.join(HERE, "_ref", "libf3d_ref.so")
        if not os.path.exists(so):
            return None
        R = C.CDLL(so)
        R.ref_max_warp_level.restype = C.c_size_t
        R.ref_max_warp_level.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_float]
        R.ref_params_first_push_wins.restype = C.c_int
        R.ref_have_data3d.restype = C.c_int
        if R.ref_have_data3d():
            fp = C.POINTER(C.c_float)
            R.ref_read_raw_u8.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, fp]
            R.ref_read_raw_f32.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, fp]
            R.ref_write_raw.argtypes = [C.c_char_p, fp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
            R.ref_write_vtk.argtypes = [C.c_char_p, fp, fp, fp, C.c_size_t, C.c_size_t, C.c_size_t]
        if hasattr(R, "ref_have_host_ops"):
            fp = C.POINTER(C.c_float)
            sz = C.c_size_t
            R.ref_have_host_ops.restype = C.c_int
            R.ref_warp.argtypes = [fp, fp, fp, fp, fp, sz, sz, sz, C.c_float, C.c_float, C.c_float, fp]
            R.ref_flow_stats.argtypes = [fp, fp, fp, sz, sz, sz, fp, fp, fp]
        _REF = R
    return _REF


def ref_host_ops():
    """True when the reference's host-only operators (CPU warp, flow statistics) were built (oracle/_ref/libf3d_ref_ops.so)."""
    r = ref()
    return bool(r is not None and hasattr(r, "ref_have_host_ops") and r.ref_have_host_ops())


def ref_warp(f0, f1, u, v, w, h):
    """The REFERENCE's own CPU warp (cuda_operation_register_p.cpp:96-139) on dense [D, H, W] volumes."""
    D, H, W = f0.shape
    out = np.empty_like(f0)
    ok = ref().ref_warp(_p(f0), _p(f1), _p(u), _p(v), _p(w), W, H, D, h[0], h[1], h[2], _p(out))
    assert ok
    return out


def ref_flow_stats(u, v, w):
    """(min, max, avg) exactly as the reference's CudaOperationStatP computes them (cuda_operation_stat_p.cpp:85-104)."""
    D, H, W = u.shape
    mn, mx, avg = C.c_float(), C.c_float(), C.c_float()
    ok = ref().ref_flow_stats(_p(u), _p(v), _p(w), W, H, D, C.byref(mn), C.byref(mx), C.byref(avg))
    assert ok
    return mn.value, mx.value, avg.value


def _p(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_float))


def geom(container, z_base=0, z_lo=0, z_hi=None):
    """Geometry of a container array of shape [Dc, Hc, pitch_f]."""
    dc, hc, pf = container.shape
    return Geom(hc, pf, z_base, z_lo, dc if z_hi is None else z_hi)


def make_params(**kw):
    d = dict(DEFAULT_PARAMS)
    d.update(kw)
    return Params(**d)


def max_warp_level(w, h, d, sf):
    return int(lib().orc_max_warp_level(w, h, d, sf))


def level_geometry(w0, h0, d0, sf, level):
    w, h, d = C.c_size_t(), C.c_size_t(), C.c_size_t()
    hx, hy, hz = C.c_float(), C.c_float(), C.c_float()
    lib().orc_level_geometry(w0, h0, d0, sf, level, w, h, d, hx, hy, hz)
    return (w.value, h.value, d.value), (hx.value, hy.value, hz.value)


def gaussian_taps(sigma):
    taps = np.zeros(51, np.float32)
    r = lib().orc_gaussian_taps(sigma, _p(taps), 51)
    if r < 0:
        raise ValueError("sigma too large")
    return r, taps[: 2 * r + 1].copy()


def conv_axis(dst, src, dims, radius, taps, axis, g=None):
    W, H, D = dims
    g = g or geom(src, z_hi=D)
    t = np.ascontiguousarray(taps, np.float32)
    lib().orc_conv_axis(_p(dst), _p(src), W, H, D, radius, _p(t), axis, g)


def gaussian(src, dims, sigma, g=None):
    """rows -> columns -> slices like cuda_operation_convolution.cpp:172-181; returns the output container."""
    r, taps = gaussian_taps(sigma)
    a = np.full_like(src, np.nan)
    b = np.full_like(src, np.nan)
    conv_axis(a, src, dims, r, taps, 0, g)
    conv_axis(b, a, dims, r, taps, 1, g)
    conv_axis(a, b, dims, r, taps, 2, g)
    return a


def resample_axis(inp, out, out_dims, in_n, axis, g_in=None, g_out=None):
    ow, oh, od = out_dims
    g_in = g_in or geom(inp)
    g_out = g_out or geom(out, z_hi=od)
    lib().orc_resample_axis(_p(inp), _p(out), ow, oh, od, in_n, axis, g_in, g_out)


def resample(inp, in_dims, out_dims):
    """Full X->Y->Z resample inside same-shaped containers (cuda_operation_resample.cpp:95-105)."""
    iw, ih, idp = in_dims
    ow, oh, od = out_dims
    out = np.full_like(inp, np.nan)
    tmp = np.full_like(inp, np.nan)
    gi = geom(inp, z_hi=idp)
    resample_axis(inp, out, (ow, ih, idp), iw, 0, gi, geom(out, z_hi=idp))
    resample_axis(out, tmp, (ow, oh, idp), ih, 1, gi, geom(tmp, z_hi=idp))
    resample_axis(tmp, out, (ow, oh, od), idp, 2, gi, geom(out, z_hi=od))
    return out


def warp(f0, f1, u, v, w, dims, h, g=None):
    W, H, D = dims
    out = np.full_like(f0, np.nan)
    lib().orc_warp(_p(f0), _p(f1), _p(u), _p(v), _p(w), W, H, D, h[0], h[1], h[2], _p(out), g or geom(f0, z_hi=D))
    return out


def phi_ksi(f0, f1, u, v, w, du, dv, dw, dims, h, eps_s, eps_d, g=None):
    W, H, D = dims
    phi = np.full_like(f0, np.nan)
    ksi = np.full_like(f0, np.nan)
    lib().orc_phi_ksi(_p(f0), _p(f1), _p(u), _p(v), _p(w), _p(du), _p(dv), _p(dw), W, H, D,
                      h[0], h[1], h[2], eps_s, eps_d, _p(phi), _p(ksi), g or geom(f0, z_hi=D))
    return phi, ksi


def solve_sweep(f0, f1, u, v, w, du, dv, dw, phi, ksi, dims, h, alpha, g=None, out=None):
    W, H, D = dims
    if out is None:
        out = tuple(np.full_like(f0, np.nan) for _ in range(3))
    lib().orc_solve_sweep(_p(f0), _p(f1), _p(u), _p(v), _p(w), _p(du), _p(dv), _p(dw), _p(phi), _p(ksi), W, H, D,
                          h[0], h[1], h[2], alpha, _p(out[0]), _p(out[1]), _p(out[2]), g or geom(f0, z_hi=D))
    return out


def add(a, b, dims, g=None):
    W, H, D = dims
    lib().orc_add(_p(a), _p(b), W, H, D, g or geom(a, z_hi=D))


def flow_stats(u, v, w, dims, g=None):
    """(min, max, avg as the reference's float scan computes it, sum in double) of the flow magnitude"""
    W, H, D = dims
    mn, mx, avg, s = C.c_float(), C.c_float(), C.c_float(), C.c_double()
    lib().orc_flow_stats(_p(u), _p(v), _p(w), W, H, D, g or geom(u, z_hi=D), C.byref(mn), C.byref(mx), C.byref(avg), C.byref(s))
    return mn.value, mx.value, avg.value, s.value


def residual_stats(f0, fw, dims, g=None):
    """(sum of squares, sum of absolute values, max) of warped - frame_0"""
    W, H, D = dims
    ssq, sab, mx = C.c_double(), C.c_double(), C.c_float()
    lib().orc_residual_stats(_p(f0), _p(fw), W, H, D, g or geom(f0, z_hi=D), C.byref(ssq), C.byref(sab), C.byref(mx))
    return ssq.value, sab.value, mx.value


def median(inp, dims, r, g=None):
    W, H, D = dims
    out = np.full_like(inp, np.nan)
    lib().orc_median(_p(inp), _p(out), W, H, D, r, g or geom(inp, z_hi=D))
    return out


def compute_flow(frame0, frame1, pitch_f=0, **kw):
    """OpticalFlowE::ComputeFlow on dense [D, H, W] float32 volumes -> (u, v, w), levels."""
    f0 = np.ascontiguousarray(frame0, np.float32)
    f1 = np.ascontiguousarray(frame1, np.float32)
    D, H, W = f0.shape
    u, v, w = (np.empty_like(f0) for _ in range(3))
    prm = make_params(**kw)
    levels = lib().orc_compute_flow(_p(f0), _p(f1), W, H, D, C.byref(prm), pitch_f, _p(u), _p(v), _p(w))
    return (u, v, w), levels


def num_threads():
    return int(lib().orc_num_threads())
