"""cuda-flow3d_amd -- Python binding of the MI355X-native 3-D optical-flow solver.

The product is native: libf3d_hip.so (hand-written gfx950 kernels behind the C ABI of include/f3d.h) and
libf3d_host.so (the C++ driver / operator classes that mirror the reference's src/optical_flow and
src/cuda_operations, C ABI in include/f3d_host.h).  This module only binds those two libraries with ctypes
and moves numpy volumes ([z, y, x], float32, x fastest like the reference's Data3D) across the boundary.
There is no Python or CPU implementation of any kernel here: if the libraries are missing, importing the
native handles raises -- build them with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C cuda-flow3d_amd`.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.environ.get("F3D_LIBDIR") or os.path.join(_HERE, "lib")  # F3D_LIBDIR: A/B timing of two builds in one GPU call

DEFAULT_PARAMS = dict(
    warp_levels_count=40, warp_scale_factor=0.95, outer_iterations_count=40, inner_iterations_count=5,
    equation_alpha=7.5, equation_smoothness=0.001, equation_data=0.001, median_radius=5, gaussian_sigma=2.0,
)  # src/main.cpp:77-85


class F3dError(RuntimeError):
    pass


class Size4(C.Structure):  # DataSize4 / f3d_size4
    _fields_ = [("width", C.c_size_t), ("height", C.c_size_t), ("depth", C.c_size_t), ("pitch", C.c_size_t)]


class Slab(C.Structure):  # f3d_slab
    _fields_ = [("z_base", C.c_int), ("z_lo", C.c_int), ("z_hi", C.c_int)]


class LevelStat(C.Structure):  # f3d_level_stat
    _fields_ = [("level", C.c_int), ("width", C.c_size_t), ("height", C.c_size_t), ("depth", C.c_size_t),
                ("residual_rms", C.c_double), ("residual_mean_abs", C.c_double), ("residual_max_abs", C.c_float),
                ("flow_min", C.c_float), ("flow_max", C.c_float), ("flow_avg", C.c_float)]


class FlowParams(C.Structure):  # f3d_flow_params
    _fields_ = [
        ("warp_levels_count", C.c_size_t), ("warp_scale_factor", C.c_float),
        ("outer_iterations_count", C.c_size_t), ("inner_iterations_count", C.c_size_t),
        ("equation_alpha", C.c_float), ("equation_smoothness", C.c_float), ("equation_data", C.c_float),
        ("median_radius", C.c_size_t), ("gaussian_sigma", C.c_float),
    ]


_hip = None
_host = None
_fp = C.POINTER(C.c_float)
_dp = C.c_uint64
_dpp = C.POINTER(C.c_uint64)
_sz = C.c_size_t
_slabp = C.POINTER(Slab)


def _load(name):
    path = os.path.join(_LIBDIR, name)
    if not os.path.exists(path):
        raise F3dError(f"{path} is missing: the HIP extension has not been built (make -C {_HERE}); "
                       "there is no fallback path")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def hip():
    """Handle of libf3d_hip.so with argument types declared (include/f3d.h)."""
    global _hip
    if _hip is not None:
        return _hip
    L = _load("libf3d_hip.so")
    L.f3d_last_error.restype = C.c_char_p
    sig = {
        "f3d_init": [C.c_int], "f3d_shutdown": [], "f3d_is_initialized": [], "f3d_device_count": [C.POINTER(C.c_int)],
        "f3d_crash_maps_enable": [C.c_char_p],
        "f3d_lane_create": [C.POINTER(C.c_void_p)], "f3d_lane_make_current": [C.c_void_p], "f3d_lane_is_private": [], "f3d_lane_get_current": [C.POINTER(C.c_void_p)],
        "f3d_lane_destroy": [C.c_void_p],
        "f3d_selftest_weights": [C.c_uint, C.c_uint, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong),
                                 C.POINTER(C.c_uint)],
        "f3d_device_name": [C.c_char_p, _sz], "f3d_mem_info": [C.POINTER(_sz), C.POINTER(_sz)],
        "f3d_lds_per_workgroup": [C.POINTER(C.c_int)],
        "f3d_alloc_pitched": [C.POINTER(_dp), C.POINTER(_sz), _sz, _sz], "f3d_free": [_dp],
        "f3d_memset2d": [_dp, _sz, C.c_int, _sz, _sz],
        "f3d_copy3d_h2d": [_dp, _sz, _sz, _sz, _fp, _sz, _sz, _sz],
        "f3d_copy3d_d2h": [_fp, _sz, _sz, _sz, _dp, _sz, _sz, _sz],
        "f3d_copy_planes_h2d": [_dp, _sz, _sz, _sz, _fp, _sz, _sz, _sz, _sz, _sz],
        "f3d_copy_planes_d2h": [_fp, _sz, _sz, _sz, _sz, _sz, _dp, _sz, _sz, _sz],
        "f3d_queue_create": [C.POINTER(C.c_void_p)], "f3d_queue_destroy": [C.c_void_p], "f3d_queue_sync": [C.c_void_p],
        "f3d_event_record_on": [C.c_void_p, C.c_void_p], "f3d_queue_wait_event": [C.c_void_p, C.c_void_p],
        "f3d_copy_planes_h2d_on": [C.c_void_p, _dp, _sz, _sz, _sz, _fp, _sz, _sz, _sz, _sz, _sz],
        "f3d_copy_planes_d2h_on": [C.c_void_p, _fp, _sz, _sz, _sz, _sz, _sz, _dp, _sz, _sz, _sz],
        "f3d_copy_rect_d2d": [_dp, _sz, _sz, _sz, _dp, _sz, _sz, _sz, _sz, _sz, _sz],
        "f3d_host_register": [C.c_void_p, _sz], "f3d_host_unregister": [C.c_void_p],
        "f3d_host_is_pinned": [C.c_void_p, C.POINTER(C.c_int)],
        "f3d_copy_d2d": [_dp, _dp, _sz], "f3d_set_container": [C.POINTER(Size4)], "f3d_get_container": [C.POINTER(Size4)],
        "f3d_event_create": [C.POINTER(C.c_void_p)], "f3d_event_record": [C.c_void_p],
        "f3d_event_sync": [C.c_void_p], "f3d_event_elapsed_ms": [_fp, C.c_void_p, C.c_void_p],
        "f3d_event_destroy": [C.c_void_p], "f3d_stream_sync": [],
        "f3d_phi_ksi": [_dp] * 8 + [_sz] * 3 + [C.c_float] * 5 + [_dp, _dp, _slabp],
        "f3d_phi_ksi_zones": [_dp] * 8 + [_sz] * 3 + [C.c_float] * 5 + [_dp, _dp, _slabp, _slabp],
        "f3d_solve_sweep": [_dp] * 10 + [_sz] * 3 + [C.c_float] * 4 + [_dp] * 3 + [_slabp],
        "f3d_solve_sweep2": [_dp] * 10 + [_sz] * 3 + [C.c_float] * 4 + [_dp] * 3 + [_slabp],
        "f3d_solve_sweep_phi_ksi": [_dp] * 10 + [_sz] * 3 + [C.c_float] * 6 + [_dp] * 5 + [_slabp],
        "f3d_solve_sweep_phi_ksi_edges": [_dp] * 10 + [_sz] * 3 + [C.c_float] * 6 + [_dp] * 5 + [_slabp, C.c_int, C.c_int],
        "f3d_frame_derivatives": [_dp, _dp, _sz, _sz, _sz, C.c_float, C.c_float, C.c_float, _dp, _dp, _dp, _dp, _slabp],
        "f3d_solve_sweep_phi_ksi_edges_fd": [_dp] * 12 + [_sz] * 3 + [C.c_float] * 6 + [_dp] * 5 + [_slabp, C.c_int, C.c_int],
        "f3d_solve_sweep2_fd": [_dp] * 12 + [_sz] * 3 + [C.c_float] * 4 + [_dp] * 3 + [_slabp],
        "f3d_fused_launches_march_along_y": [_sz, _sz, _sz],
        "f3d_solve_sweep3": [_dp] * 10 + [_sz] * 3 + [C.c_float] * 4 + [_dp] * 3 + [_slabp],
        "f3d_solve_sweep2_phi_ksi": [_dp] * 10 + [_sz] * 3 + [C.c_float] * 6 + [_dp] * 5 + [_slabp],
        "f3d_solve_sweep_phi_ksi_fd": [_dp] * 12 + [_sz] * 3 + [C.c_float] * 6 + [_dp] * 5 + [_slabp],
        "f3d_warp": [_dp] * 5 + [_sz] * 3 + [C.c_float] * 3 + [_dp, _slabp],
        "f3d_resample_x": [_dp, _dp, _sz, _sz, _sz, _sz, _slabp],
        "f3d_resample_y": [_dp, _dp, _sz, _sz, _sz, _sz, _slabp],
        "f3d_resample_z": [_dp, _dp, _sz, _sz, _sz, _sz, _slabp, _slabp],
        "f3d_add": [_dp, _dp, _sz, _sz, _sz, _slabp],
        "f3d_median": [_dp, _sz, _sz, _sz, _sz, _dp, _slabp],
        # up to three volumes of one box per launch (arrays of device pointers)
        "f3d_resample_x_n": [_dpp, _dpp, _sz, _sz, _sz, _sz, _sz, _slabp],
        "f3d_resample_y_n": [_dpp, _dpp, _sz, _sz, _sz, _sz, _sz, _slabp],
        "f3d_resample_z_n": [_dpp, _dpp, _sz, _sz, _sz, _sz, _sz, _slabp, _slabp],
        "f3d_add_n": [_dpp, _dpp, _sz, _sz, _sz, _sz, _slabp],
        "f3d_median_n": [_dpp, _sz, _sz, _sz, _sz, _sz, _dpp, _slabp],
        "f3d_clear_box_n": [_dpp, _sz, _sz, _sz, _sz, _slabp],
        "f3d_set_conv_taps": [_fp, _sz],
        "f3d_conv_rows": [_dp, _dp, _sz, _sz, _sz, _sz, _slabp],
        "f3d_conv_cols": [_dp, _dp, _sz, _sz, _sz, _sz, _slabp],
        "f3d_conv_slices": [_dp, _dp, _sz, _sz, _sz, _sz, _slabp],
        "f3d_conv_rows_cols": [_dp, _dp, _sz, _sz, _sz, _sz, _slabp],
        "f3d_range_push": [C.c_char_p], "f3d_range_pop": [],
        "f3d_prof_enable": [C.c_int], "f3d_prof_reset": [], "f3d_prof_select": [C.c_uint],
        "f3d_prof_read": [C.c_int, _sz, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double)],
        "f3d_abs_max": [_dp, _sz, _sz, _sz, _slabp, _fp],
        "f3d_comm_unique_id": [C.c_void_p], "f3d_comm_init": [C.c_void_p, C.c_int, C.c_int], "f3d_comm_destroy": [],
        "f3d_comm_rank": [C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "f3d_comm_info": [C.POINTER(C.c_int)] * 4 + [C.POINTER(C.c_ulonglong)] * 2,
        "f3d_pack_planes": [_dp, C.c_int, C.c_int, _sz, _sz, _dp, _sz],
        "f3d_unpack_planes": [_dp, C.c_int, C.c_int, _sz, _sz, _dp, _sz],
        "f3d_copy_planes": [_dp, C.c_int, _dp, C.c_int, C.c_int, _sz, _sz],
        "f3d_copy_plane_segments": [C.POINTER(_dp), C.POINTER(C.c_int), C.POINTER(_dp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, _sz, _sz],
        "f3d_pack_segments": [C.POINTER(_dp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_sz), C.c_int, _sz, _sz, _dp],
        "f3d_unpack_segments": [C.POINTER(_dp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_sz), C.c_int, _sz, _sz, _dp],
        "f3d_comm_sendrecv": [_dp, C.POINTER(_sz), C.POINTER(_sz), _dp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(C.c_int), C.c_int],
        "f3d_flow_stats": [_dp, _dp, _dp, _sz, _sz, _sz, _slabp, _fp, _fp, C.POINTER(C.c_double)],
        "f3d_residual_stats": [_dp, _dp, _sz, _sz, _sz, _slabp, C.POINTER(C.c_double), C.POINTER(C.c_double), _fp],
        "f3d_comm_sendrecv_begin": [_dp, C.POINTER(_sz), C.POINTER(_sz), _dp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(C.c_int), C.c_int],
        "f3d_comm_sendrecv_end": [],
        "f3d_comm_allreduce_max_f32": [_fp],
        "f3d_comm_timing": [C.c_int],
        "f3d_comm_mark": [C.c_int, C.c_int],
        "f3d_comm_timing_read": [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.POINTER(C.c_ulonglong)],
    }
    for name, args in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            # an older build named by F3D_LIBDIR (A/B timing of two libraries in one GPU call) may lack the newest entry points;
            # the library of the package itself must export every one of them
            if os.environ.get("F3D_LIBDIR"):
                continue
            raise
        fn.argtypes = args
        fn.restype = C.c_int
    _hip = L
    if hasattr(L, "f3d_crash_maps_enable"):
        _arm_crash_maps(L)
    return L


def _arm_crash_maps(L):
    """F3D_CRASH_MAPS=<file> (or any run under a rocprofiler tool): a fatal signal leaves /proc/self/maps in <file> before the
    usual handlers run, so the anonymous frames of a native stack trace can be put into libraries (include/f3d.h,
    f3d_crash_maps_enable; the one crash on record happened under `rocprofv3 --pmc` and could only be resolved after the fact)."""
    path = os.environ.get("F3D_CRASH_MAPS")
    if not path:
        blob = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
        if "rocprof" not in blob:
            return
        base = os.environ.get("F3D_OUT") or os.getcwd()
        path = os.path.join(base, f"f3d_crash_maps.{os.getpid()}.txt")
    L.f3d_crash_maps_enable(path.encode())


def host():
    """Handle of libf3d_host.so with argument types declared (include/f3d_host.h)."""
    global _host
    if _host is not None:
        return _host
    hip()
    L = _load("libf3d_host.so")
    pp = C.POINTER(FlowParams)
    sig = {
        "f3d_flow_create": [C.POINTER(C.c_void_p)], "f3d_flow_initialize": [C.c_void_p, _sz, _sz, _sz],
        "f3d_flow_compute": [C.c_void_p, _fp, _fp, pp, C.c_int, _fp, _fp, _fp],
        "f3d_flow_upload": [C.c_void_p, _fp, _fp],
        "f3d_flow_compute_resident": [C.c_void_p, pp, C.c_int, _fp],
        "f3d_flow_download": [C.c_void_p, _fp, _fp, _fp],
        "f3d_flow_container": [C.c_void_p, C.POINTER(Size4)], "f3d_flow_destroy": [C.c_void_p],
        "f3d_flow_set_level_stats": [C.c_void_p, C.c_int], "f3d_flow_level_stat_count": [C.c_void_p, C.POINTER(_sz)],
        "f3d_flow_level_stat": [C.c_void_p, _sz, C.POINTER(LevelStat)],
        "f3d_flow_final_residual": [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)],
        "f3d_op_create": [C.POINTER(C.c_void_p), C.c_char_p], "f3d_op_initialize": [C.c_void_p, C.POINTER(Size4)],
        "f3d_op_execute": [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), _sz],
        "f3d_op_execute_batch": [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(_sz), _sz],
        "f3d_op_set_slab": [C.c_void_p, _slabp], "f3d_op_destroy": [C.c_void_p],
        "f3d_level_geometry": [_sz, _sz, _sz, C.c_float, C.c_int, C.POINTER(Size4), _fp, _fp, _fp],
        "f3d_gaussian_taps": [C.c_float, _fp, _sz, C.POINTER(_sz)],
        "f3d_raw_read_u8": [C.c_char_p, _sz, _sz, _sz, _fp], "f3d_raw_read_f32": [C.c_char_p, _sz, _sz, _sz, _fp],
        "f3d_raw_write_u8": [C.c_char_p, _fp, _sz, _sz, _sz], "f3d_raw_write_f32": [C.c_char_p, _fp, _sz, _sz, _sz],
        "f3d_vtk_write_flow": [C.c_char_p, _fp, _fp, _fp, _sz, _sz, _sz],
        "f3d_synth_pair": [_sz, _sz, _sz, _fp, _fp],
        "f3d_synth_planes": [_sz, _sz, _sz, _sz, _sz, _fp, _fp, _fp],
        "f3d_slabflow_create": [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int],
        "f3d_slabflow_initialize": [C.c_void_p, _sz, _sz, _sz],
        "f3d_slabflow_compute": [C.c_void_p, _fp, _fp, pp, _fp, _fp, _fp],
        "f3d_slabflow_upload": [C.c_void_p, _fp, _fp],
        "f3d_slabflow_compute_resident": [C.c_void_p, pp, _fp],
        "f3d_slabflow_download": [C.c_void_p, _fp, _fp, _fp],
        "f3d_slabflow_overlapped_iterations": [C.c_void_p, C.POINTER(_sz)],
        "f3d_slabflow_batched_exchanges": [C.c_void_p, C.POINTER(_sz)],
        "f3d_slabflow_gathered_warps": [C.c_void_p, C.POINTER(_sz)],
        "f3d_slabflow_stage_exchanges": [C.c_void_p, C.POINTER(_sz)],
        "f3d_slabflow_set_exchange_per_stage": [C.c_void_p, C.c_int],
        "f3d_slabflow_destroy": [C.c_void_p],
        "f3d_plan_owned": [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "f3d_plan_exchange": [C.c_int] * 5 + [C.POINTER(C.c_int)] * 5 + [C.c_int],
        "f3d_plan_resample_source": [C.c_int] * 4 + [C.POINTER(C.c_int)] * 2,
        "f3d_volume_wrap": [C.POINTER(C.c_void_p), _fp, _sz, _sz, _sz], "f3d_volume_destroy": [C.c_void_p],
        "f3d_op_solve_p_last": [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_sz), C.POINTER(C.c_int)],
        "f3d_op_solve_p_fused_weights": [C.c_void_p, C.POINTER(C.c_int)],
        "f3d_plan_solve_piecemeal": [_sz, _sz, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 5,
        "f3d_pflow_create": [C.POINTER(C.c_void_p)], "f3d_pflow_initialize": [C.c_void_p, _sz, _sz, _sz],
        "f3d_pflow_compute": [C.c_void_p, _fp, _fp, _sz, _sz, _sz, pp, C.c_int, _fp, _fp, _fp, _fp],
        "f3d_pflow_stats": [C.c_void_p, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)], "f3d_pflow_destroy": [C.c_void_p],
        "f3d_pflow_set_resident": [C.c_void_p, C.c_int], "f3d_pflow_set_full_pipeline": [C.c_void_p, C.c_int], "f3d_pflow_originals_on_device": [C.c_void_p, C.POINTER(C.c_int)],
        "f3d_pflow_operator_seconds": [C.c_void_p, C.POINTER(C.c_double)],
        "f3d_pflow_levels_registered_inside": [C.c_void_p, C.POINTER(_sz)],
        "f3d_pflow_levels_with_constants_on_device": [C.c_void_p, C.POINTER(_sz)],
        "f3d_host_shutdown": [],
    }
    for name, args in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            if os.environ.get("F3D_LIBDIR"):   # an older build under A/B timing (see hip())
                continue
            raise
        fn.argtypes = args
        fn.restype = C.c_int
    L.f3d_flow_default_params.argtypes = [pp]
    L.f3d_flow_default_params.restype = None
    L.f3d_op_name.argtypes = [C.c_void_p]
    L.f3d_op_name.restype = C.c_char_p
    L.f3d_max_warp_level.argtypes = [_sz, _sz, _sz, C.c_float]
    L.f3d_max_warp_level.restype = _sz
    L.f3d_volume_object.argtypes = [C.c_void_p]
    L.f3d_volume_object.restype = C.c_void_p
    L.f3d_volume_data.argtypes = [C.c_void_p]
    L.f3d_volume_data.restype = C.c_void_p
    L.f3d_piecemeal_budget_bytes.argtypes = []
    L.f3d_piecemeal_budget_bytes.restype = _sz
    _host = L
    return L


def shutdown():
    """Orderly end of device use: host volumes that are still page-locked are released, then f3d_host_shutdown() drops the
    out-of-core arena, the copy queues, the RCCL communicator, the timing events and the library stream.  Registered with
    atexit on import, so it runs while the interpreter, numpy's heap and the HIP runtime are all still alive -- nothing is
    left to the order in which the process unloads libraries.  Idempotent."""
    HostVolume.release_all()
    if _host is not None:
        _host.f3d_host_shutdown()
    elif _hip is not None and _hip.f3d_is_initialized():
        _hip.f3d_comm_destroy()
        _hip.f3d_shutdown()


atexit.register(shutdown)


def check(status, what="f3d call"):
    if status != 0:
        msg = hip().f3d_last_error()
        raise F3dError(f"{what} failed: {msg.decode() if msg else 'status %d' % status}")


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def make_params(**kw):
    d = dict(DEFAULT_PARAMS)
    unknown = set(kw) - set(d)
    if unknown:
        raise TypeError(f"unknown flow parameters: {sorted(unknown)}")
    d.update(kw)
    return FlowParams(**d)


# ---- host-only helpers --------------------------------------------------------------------------------------

def max_warp_level(width, height, depth, scale_factor):
    return int(host().f3d_max_warp_level(width, height, depth, scale_factor))


def level_geometry(width, height, depth, scale_factor, level):
    size = Size4()
    hx, hy, hz = C.c_float(), C.c_float(), C.c_float()
    check(host().f3d_level_geometry(width, height, depth, scale_factor, level, C.byref(size), hx, hy, hz))
    return (size.width, size.height, size.depth), (hx.value, hy.value, hz.value)


def gaussian_taps(sigma):
    taps = np.zeros(51, np.float32)
    radius = _sz()
    if host().f3d_gaussian_taps(sigma, taps.ctypes.data_as(_fp), 51, C.byref(radius)) != 0:
        raise F3dError("sigma too large for the 51-tap limit")
    return int(radius.value), taps[: 2 * radius.value + 1].copy()


def read_raw(path, dims, u8=True):
    w, h, d = dims
    out = np.empty((d, h, w), np.float32)
    fn = host().f3d_raw_read_u8 if u8 else host().f3d_raw_read_f32
    if fn(os.fsencode(path), w, h, d, out.ctypes.data_as(_fp)) != 0:
        raise F3dError(f"cannot read {path} as {w}x{h}x{d}")
    return out


def write_raw(path, vol, u8=False):
    vol, p = _f32(vol)
    d, h, w = vol.shape
    fn = host().f3d_raw_write_u8 if u8 else host().f3d_raw_write_f32
    if fn(os.fsencode(path), p, w, h, d) != 0:
        raise F3dError(f"cannot write {path}")


def write_vtk(path, u, v, w_):
    u, pu = _f32(u)
    v, pv = _f32(v)
    w_, pw = _f32(w_)
    d, h, w = u.shape
    if host().f3d_vtk_write_flow(os.fsencode(path), pu, pv, pw, w, h, d) != 0:
        raise F3dError(f"cannot write {path}")


def synth_pair(width, height, depth):
    f0 = np.empty((depth, height, width), np.float32)
    f1 = np.empty_like(f0)
    check(host().f3d_synth_pair(width, height, depth, f0.ctypes.data_as(_fp), f1.ctypes.data_as(_fp)))
    return f0, f1


def synth_planes(width, height, depth, z_lo, z_hi, frame_0, frame_1):
    """Render planes [z_lo, z_hi) of the synthetic pair, unscaled, into full-size arrays; returns max(frame_0 planes)."""
    m = C.c_float()
    check(host().f3d_synth_planes(width, height, depth, z_lo, z_hi, frame_0.ctypes.data_as(_fp), frame_1.ctypes.data_as(_fp),
                                  C.byref(m)), "f3d_synth_planes")
    return m.value


def flow_plane_digests(flow, z_lo=0, z_hi=None):
    """sha256 of every plane z_lo <= z < z_hi of each of (u, v, w) (float32, -0 normalised to +0): three lists of 32-byte
    digests.  Planes hash independently, so the ranks of a z-slab run can each hash what they own."""
    import hashlib
    out = []
    for vol in flow:
        hi = vol.shape[0] if z_hi is None else z_hi
        out.append([hashlib.sha256(np.ascontiguousarray(vol[z] + np.float32(0.0)).tobytes()).digest() for z in range(z_lo, hi)])
    return out


def combine_plane_digests(per_component):
    """One hex digest from the per-plane digests of (u, v, w), all planes of u first, in plane order."""
    import hashlib
    h = hashlib.sha256()
    for comp in per_component:
        for d in comp:
            h.update(d)
    return h.hexdigest()


# ---- device memory ---------------------------------------------------------------------------------------------

class Containers:
    """A set of equally sized pitched device containers (what OpticalFlowE::InitCudaMemory allocates)."""

    def __init__(self, width, height, depth, device=-1):
        check(hip().f3d_init(device), "f3d_init")
        self.width, self.height, self.depth = width, height, depth
        self.pitch = 0
        self._ptrs = []

    @property
    def size4(self):
        return Size4(self.width, self.height, self.depth, self.pitch)

    def set_current(self):
        s = self.size4
        check(hip().f3d_set_container(C.byref(s)), "f3d_set_container")

    def alloc(self, fill=None):
        ptr, pitch = _dp(), _sz()
        check(hip().f3d_alloc_pitched(C.byref(ptr), C.byref(pitch), self.width * 4, self.height * self.depth),
              "f3d_alloc_pitched")
        if self.pitch and pitch.value != self.pitch:
            raise F3dError("containers came back with different pitches")
        self.pitch = pitch.value
        self._ptrs.append(ptr.value)
        if fill is not None:
            # byte pattern over the whole pitched allocation (0xFF.. = NaN poison)
            check(hip().f3d_memset2d(ptr.value, self.pitch, fill, self.pitch, self.height * self.depth))
        return ptr.value

    def upload(self, ptr, vol, plane0=0):
        vol, p = _f32(vol)
        d, h, w = vol.shape
        check(hip().f3d_copy3d_h2d(ptr, self.pitch, self.height, plane0, p, w, h, d), "f3d_copy3d_h2d")

    def download(self, ptr, dims, plane0=0):
        w, h, d = dims
        out = np.empty((d, h, w), np.float32)
        check(hip().f3d_copy3d_d2h(out.ctypes.data_as(_fp), w, h, d, ptr, self.pitch, self.height, plane0),
              "f3d_copy3d_d2h")
        return out

    def new(self, vol=None, fill=0xFF):
        """Allocate a NaN-poisoned container and optionally upload a [d, h, w] sub-box into its corner."""
        p = self.alloc(fill=fill)
        if vol is not None:
            self.upload(p, vol)
        return p

    def free(self):
        for p in self._ptrs:
            hip().f3d_free(p)
        self._ptrs = []


def sync():
    check(hip().f3d_stream_sync(), "f3d_stream_sync")


class Lane:
    """A stream and a container geometry of one's own (include/f3d.h, f3d_lane_*): a driver created and used between
    make_current() and release() in ONE thread runs beside the drivers of other threads instead of in line with them.
        with f3d.Lane():            # in a worker thread
            flow = f3d.OpticalFlow(); flow.initialize(w, h, d); u, v, w = flow.compute(f0, f1); flow.destroy()"""

    def __init__(self):
        check(hip().f3d_init(-1), "f3d_init")
        self._h = C.c_void_p()
        check(hip().f3d_lane_create(C.byref(self._h)), "f3d_lane_create")

    def make_current(self):
        check(hip().f3d_lane_make_current(self._h), "f3d_lane_make_current")

    def release(self):
        hip().f3d_lane_make_current(None)

    def destroy(self):
        if self._h:
            hip().f3d_lane_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        self.make_current()
        return self

    def __exit__(self, *exc):
        self.release()
        self.destroy()
        return False


def mem_info():
    """(free, total) bytes of device memory."""
    check(hip().f3d_init(-1), "f3d_init")
    free, total = _sz(), _sz()
    check(hip().f3d_mem_info(C.byref(free), C.byref(total)), "f3d_mem_info")
    return free.value, total.value


# ---- operator layer (CudaOperation* through the string-keyed bag) -------------------------------------------------

_PTR_KEYS = {
    "dev_frame_0", "dev_frame_1", "dev_flow_u", "dev_flow_v", "dev_flow_w", "dev_phi", "dev_ksi", "dev_flow_du",
    "dev_flow_dv", "dev_flow_dw", "dev_temp_du", "dev_temp_dv", "dev_temp_dw", "dev_input", "dev_output", "dev_temp",
    "operand_0", "operand_1",
}
_SIZE_T_KEYS = {"outer_iterations_count", "inner_iterations_count", "radius", "warp_levels_count", "median_radius"}
_FLOAT_KEYS = {"equation_alpha", "equation_smoothness", "equation_data", "hx", "hy", "hz", "gaussian_sigma",
               "warp_scale_factor"}
_SIZE4_KEYS = {"data_size", "resample_size", "container_size"}


class Stat3(C.Structure):
    """src/data_types/data_structs.h:29-33"""
    _fields_ = [("min", C.c_float), ("max", C.c_float), ("avg", C.c_float)]


class HostVolume:
    """Dense host volume [z, y, x] float32 handed to the piecemeal operators as a Data3D* (f3d_volume_wrap).  Like the
    reference's Data3D::Swap, registration_p / solve_p exchange STORAGE between the volumes of one call, so a volume may
    end up holding the array another one was created with.  The class therefore keeps every wrapped array alive in a
    table keyed by its address, together with whether it is page-locked, and a volume that goes away releases the storage
    it holds AT THAT MOMENT (f3d_volume_data), never the one it started with: the survivor of a swapped pair keeps its
    memory, its page-lock and its .array.  A finalizer does the same for volumes that are dropped without destroy()."""
    _storage = {}   # address -> [array, page-locked]

    def __init__(self, array, pin=False):
        a = np.ascontiguousarray(array, dtype=np.float32)
        if a.ndim != 3:
            raise ValueError("volume must be [z, y, x]")
        if pin and a.nbytes < (32 << 20):
            # A small array lives in the allocator's shared heap: page-locked there it shares pages with its neighbours and sits under
            # a heap top that moves (a GPU memory access fault once in ~4 000 small runs, LABBOOK round 4).  The volume gets a mapping of
            # its own instead and the values are copied in; `.array` is that storage.
            import mmap
            own = np.frombuffer(mmap.mmap(-1, max(a.nbytes, mmap.PAGESIZE)), dtype=np.float32, count=a.size).reshape(a.shape)
            own[...] = a
            a = own
        if a.ctypes.data in HostVolume._storage:
            raise ValueError("this array is already wrapped by another HostVolume")
        self._h = C.c_void_p()
        d, h, w = a.shape
        check(host().f3d_volume_wrap(C.byref(self._h), a.ctypes.data_as(_fp), w, h, d), "f3d_volume_wrap")
        entry = [a, False]
        HostVolume._storage[a.ctypes.data] = entry
        if pin:  # page-locked: full link rate, and a precondition of the solver's overlapped schedule
            try:
                check(hip().f3d_init(-1), "f3d_init")
                check(hip().f3d_host_register(C.c_void_p(a.ctypes.data), a.nbytes), "f3d_host_register")
            except Exception:
                self.destroy()
                raise
            entry[1] = True
        self._finalizer = weakref.finalize(self, HostVolume._release, self._h.value)

    @staticmethod
    def _release(handle):
        """drop the Data3D and whatever storage it holds now"""
        if not handle or _host is None:
            return
        cur = _host.f3d_volume_data(C.c_void_p(handle))
        _host.f3d_volume_destroy(C.c_void_p(handle))
        entry = HostVolume._storage.pop(cur, None)
        if entry is not None and entry[1] and _hip is not None and _hip.f3d_is_initialized():
            _hip.f3d_host_unregister(C.c_void_p(cur))

    @classmethod
    def release_all(cls):
        """exit hook: remove every page-lock that is still in place (the arrays themselves are numpy's to free)"""
        for addr, entry in list(cls._storage.items()):
            if entry[1] and _hip is not None and _hip.f3d_is_initialized():
                _hip.f3d_host_unregister(C.c_void_p(addr))
            entry[1] = False

    @property
    def object(self):
        return host().f3d_volume_object(self._h)

    @property
    def array(self):
        return HostVolume._storage[host().f3d_volume_data(self._h)][0]

    def destroy(self):
        if self._h:
            fin = getattr(self, "_finalizer", None)
            if fin is not None:
                fin.detach()
            HostVolume._release(self._h.value)
            self._h = C.c_void_p()


def plan_solve_piecemeal(budget_bytes, width, height, depth, inner_iterations, outer_iterations, forced_outer_per_pass=0,
                         overlap_mode=0):
    """(chunk, outer_per_pass, halo, max_planes, overlapped) the piecemeal solver would use for a level (host arithmetic);
    overlap_mode 0 = serial schedule, 1 = copies beside the kernels, -1 = the cost model's choice."""
    out = [C.c_int() for _ in range(5)]
    check(host().f3d_plan_solve_piecemeal(budget_bytes, width, height, depth, inner_iterations, outer_iterations,
                                          forced_outer_per_pass, overlap_mode, *[C.byref(o) for o in out]))
    return tuple(o.value for o in out)


class Operation:
    """One of the operators (add, convolution, median, registration, resample, solve, stat), driven exactly like the reference drives them: Initialize({"container_size"}),
    Execute(bag of pointers to caller variables).  After execute() the (possibly swapped) pointer values are
    available in .values (the solver swaps dev_flow_d* / dev_temp_d* through the bag)."""

    def __init__(self, name):
        self._h = C.c_void_p()
        if host().f3d_op_create(C.byref(self._h), name.encode()) != 0:
            raise F3dError(f"unknown operation {name!r}")
        self.values = {}

    @property
    def name(self):
        return host().f3d_op_name(self._h).decode()

    def initialize(self, containers=None):
        if containers is None:
            return host().f3d_op_initialize(self._h, None) == 0
        s = containers.size4
        return host().f3d_op_initialize(self._h, C.byref(s)) == 0

    def set_slab(self, slab):
        self._slab = slab
        host().f3d_op_set_slab(self._h, C.byref(slab) if slab is not None else None)

    def execute(self, **params):
        store = {}
        volumes = {}
        for k, v in params.items():
            if isinstance(v, HostVolume):
                volumes[k] = v          # Data3D* keys of the piecemeal operators: the bag holds the object itself
            elif k in _PTR_KEYS:
                store[k] = _dp(v)
            elif k in _SIZE_T_KEYS:
                store[k] = _sz(v)
            elif k in _FLOAT_KEYS:
                store[k] = C.c_float(v)
            elif k in _SIZE4_KEYS:
                store[k] = Size4(v[0], v[1], v[2], 0) if not isinstance(v, Size4) else v
            elif k == "max_mag":
                store[k] = _sz(v)
            elif k == "stat":
                store[k] = v            # a Stat3 the operator fills in
            else:
                raise TypeError(f"unknown parameter key {k!r}")
        n = len(store) + len(volumes)
        keys = (C.c_char_p * n)(*[k.encode() for k in list(store) + list(volumes)])
        ptrs = (C.c_void_p * n)(*([C.cast(C.byref(v), C.c_void_p) for v in store.values()] +
                                  [C.c_void_p(v.object) for v in volumes.values()]))
        check(host().f3d_op_execute(self._h, keys, ptrs, n), "f3d_op_execute")
        self.values = {k: (v.value if hasattr(v, "value") else v) for k, v in store.items()}
        return self.values

    def execute_batch(self, bags):
        """ExecuteBatch of the add / median / resample operators: a list of parameter dicts, one per volume."""
        stores = []
        for params in bags:
            store = {}
            for k, v in params.items():
                if k in _PTR_KEYS:
                    store[k] = _dp(v)
                elif k in _SIZE_T_KEYS:
                    store[k] = _sz(v)
                elif k in _SIZE4_KEYS:
                    store[k] = Size4(v[0], v[1], v[2], 0) if not isinstance(v, Size4) else v
                else:
                    raise TypeError(f"unknown parameter key {k!r}")
            stores.append(store)
        n = sum(len(s) for s in stores)
        keys = (C.c_char_p * n)(*[k.encode() for s in stores for k in s])
        ptrs = (C.c_void_p * n)(*[C.cast(C.byref(v), C.c_void_p) for s in stores for v in s.values()])
        counts = (C.c_size_t * len(stores))(*[len(s) for s in stores])
        check(host().f3d_op_execute_batch(self._h, keys, ptrs, counts, len(stores)), "f3d_op_execute_batch")

    def solve_p_fused_weights(self):
        """whether the last solve_p execute fused the last sweep of an outer iteration with the next weights"""
        f = C.c_int()
        check(host().f3d_op_solve_p_fused_weights(self._h, C.byref(f)), "f3d_op_solve_p_fused_weights")
        return bool(f.value)

    def solve_p_last(self):
        """(chunk, outer_per_pass, halo, passes, overlapped) of the last solve_p execute"""
        c, n, h, p, o = C.c_int(), C.c_int(), C.c_int(), _sz(), C.c_int()
        check(host().f3d_op_solve_p_last(self._h, C.byref(c), C.byref(n), C.byref(h), C.byref(p), C.byref(o)), "f3d_op_solve_p_last")
        return c.value, n.value, h.value, p.value, bool(o.value)

    def destroy(self):
        if self._h:
            host().f3d_op_destroy(self._h)
            self._h = C.c_void_p()


# ---- driver (OpticalFlowE) ----------------------------------------------------------------------------------------

class OpticalFlow:
    """OpticalFlowE: Initialize(DataSize4) / ComputeFlow(frame_0, frame_1 -> u, v, w) / Destroy()."""

    def __init__(self):
        self._h = C.c_void_p()
        check(host().f3d_flow_create(C.byref(self._h)), "f3d_flow_create")
        self.dims = None

    def initialize(self, width, height, depth):
        if host().f3d_flow_initialize(self._h, width, height, depth) != 0:
            raise F3dError("OpticalFlowE::Initialize failed: " + (hip().f3d_last_error() or b"").decode())
        self.dims = (width, height, depth)
        return True

    def compute(self, frame_0, frame_1, silent=True, out=None, **kw):
        """OpticalFlowE::ComputeFlow: upload, solve, download.  `out` = three preallocated C-contiguous float32 [z,y,x] arrays
        to receive u, v, w (the reference's caller owns its flow volumes too, page-locked or not); fresh ones otherwise."""
        f0, p0 = _f32(frame_0)
        f1, p1 = _f32(frame_1)
        w, h, d = self.dims
        if f0.shape != (d, h, w) or f1.shape != (d, h, w):
            raise ValueError(f"frames must be [z,y,x] = {(d, h, w)}")
        if out is None:
            u, v, ww = (np.empty((d, h, w), np.float32) for _ in range(3))
        else:
            u, v, ww = out
            for a in (u, v, ww):
                if a.dtype != np.float32 or a.shape != (d, h, w) or not a.flags["C_CONTIGUOUS"]:
                    raise ValueError(f"out arrays must be C-contiguous float32 [z,y,x] = {(d, h, w)}")
        prm = make_params(**kw)
        check(host().f3d_flow_compute(self._h, p0, p1, C.byref(prm), int(silent), u.ctypes.data_as(_fp),
                                      v.ctypes.data_as(_fp), ww.ctypes.data_as(_fp)), "f3d_flow_compute")
        return u, v, ww

    def upload(self, frame_0, frame_1):
        f0, p0 = _f32(frame_0)
        f1, p1 = _f32(frame_1)
        check(host().f3d_flow_upload(self._h, p0, p1), "f3d_flow_upload")

    def compute_resident(self, silent=True, **kw):
        prm = make_params(**kw)
        secs = C.c_float()
        check(host().f3d_flow_compute_resident(self._h, C.byref(prm), int(silent), C.byref(secs)),
              "f3d_flow_compute_resident")
        return secs.value

    def download(self):
        w, h, d = self.dims
        u, v, ww = (np.empty((d, h, w), np.float32) for _ in range(3))
        check(host().f3d_flow_download(self._h, u.ctypes.data_as(_fp), v.ctypes.data_as(_fp), ww.ctypes.data_as(_fp)))
        return u, v, ww

    def set_level_stats(self, enable=True):
        """record, per pyramid level of every later compute, the residual before the solve and the flow statistics after it"""
        check(host().f3d_flow_set_level_stats(self._h, int(bool(enable))))

    def level_stats(self):
        """list of dicts, coarsest level first"""
        n = _sz()
        check(host().f3d_flow_level_stat_count(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            st = LevelStat()
            check(host().f3d_flow_level_stat(self._h, i, C.byref(st)))
            out.append({k: getattr(st, k) for k, _ in LevelStat._fields_})
        return out

    def final_residual(self):
        """((rms, mean |.|, max |.|) of frame_1 registered with the flow on the device against frame_0, the same unregistered)"""
        a, b = (C.c_double * 3)(), (C.c_double * 3)()
        check(host().f3d_flow_final_residual(self._h, a, b), "f3d_flow_final_residual")
        return tuple(a), tuple(b)

    def destroy(self):
        if self._h:
            host().f3d_flow_destroy(self._h)
            self._h = C.c_void_p()


class PiecemealOpticalFlow:
    """OpticalFlowP: every volume stays in host memory, z-chunks stream through the device (no pre-blur, no median, like
    the reference's piecemeal driver).  F3D_P_BUDGET_MB bounds the device memory it uses."""

    def __init__(self):
        self._h = C.c_void_p()
        check(host().f3d_pflow_create(C.byref(self._h)), "f3d_pflow_create")
        self.device_seconds = 0.0

    def initialize(self, width, height, depth):
        if host().f3d_pflow_initialize(self._h, width, height, depth) != 0:
            raise F3dError("OpticalFlowP::Initialize failed: " + (hip().f3d_last_error() or b"").decode())
        self.dims = (width, height, depth)
        return True

    def compute(self, frame_0, frame_1, silent=True, **kw):
        f0, p0 = _f32(frame_0)
        f1, p1 = _f32(frame_1)
        w, h, d = self.dims
        if f0.shape != (d, h, w) or f1.shape != (d, h, w):
            raise ValueError(f"frames must be [z,y,x] = {(d, h, w)}")
        u, v, ww = (np.empty((d, h, w), np.float32) for _ in range(3))
        prm = make_params(**kw)
        secs = C.c_float()
        check(host().f3d_pflow_compute(self._h, p0, p1, w, h, d, C.byref(prm), int(silent), u.ctypes.data_as(_fp),
                                       v.ctypes.data_as(_fp), ww.ctypes.data_as(_fp), C.byref(secs)), "f3d_pflow_compute")
        self.device_seconds = secs.value
        return u, v, ww

    def operator_seconds(self):
        """wall seconds of the last compute per operator"""
        t = (C.c_double * 6)()
        check(host().f3d_pflow_operator_seconds(self._h, t), "f3d_pflow_operator_seconds")
        return dict(zip(("frames", "flow_resample", "registration", "solve", "add", "resident_levels"), t))

    def originals_on_device(self):
        """True when the resident levels of the last compute read the original frames from device copies"""
        y = C.c_int()
        check(host().f3d_pflow_originals_on_device(self._h, C.byref(y)), "f3d_pflow_originals_on_device")
        return bool(y.value)

    def set_full_pipeline(self, enabled):
        """also run the Gaussian pre-blur and the per-level median: OpticalFlowE's whole pipeline on host volumes"""
        check(host().f3d_pflow_set_full_pipeline(self._h, int(bool(enabled))), "f3d_pflow_set_full_pipeline")

    def set_resident(self, enabled):
        """coarse levels that fit the budget stay on the device (default) or every level goes through the host"""
        check(host().f3d_pflow_set_resident(self._h, int(bool(enabled))), "f3d_pflow_set_resident")

    def stats(self):
        """(solver residencies, levels cut into chunks, coarse levels run wholly on the device) of the last compute"""
        a, b, c = _sz(), _sz(), _sz()
        check(host().f3d_pflow_stats(self._h, C.byref(a), C.byref(b), C.byref(c)), "f3d_pflow_stats")
        return a.value, b.value, c.value

    def levels_registered_inside(self):
        """host levels of the last compute whose frame 1 was registered inside the solver's first residency"""
        a = _sz()
        check(host().f3d_pflow_levels_registered_inside(self._h, C.byref(a)), "f3d_pflow_levels_registered_inside")
        return a.value

    def levels_with_constants_on_device(self):
        """host levels of the last compute whose solver held the frames and u, v, w on the device for the whole level"""
        a = _sz()
        check(host().f3d_pflow_levels_with_constants_on_device(self._h, C.byref(a)), "f3d_pflow_levels_with_constants_on_device")
        return a.value

    def destroy(self):
        if self._h:
            host().f3d_pflow_destroy(self._h)
            self._h = C.c_void_p()


# ---- multi-GPU z-slab driver (OpticalFlowSlab) -----------------------------------------------------------------------

def plan_owned(depth, rank, n_ranks):
    lo, hi = C.c_int(), C.c_int()
    check(host().f3d_plan_owned(depth, rank, n_ranks, C.byref(lo), C.byref(hi)), "f3d_plan_owned")
    return lo.value, hi.value


def plan_exchange(depth, rank, n_ranks, need_lo, need_hi):
    """[(peer, (send_lo, send_hi), (recv_lo, recv_hi)), ...] in global planes."""
    cap = max(1, n_ranks)
    arr = [(C.c_int * cap)() for _ in range(5)]
    n = host().f3d_plan_exchange(depth, rank, n_ranks, need_lo, need_hi, *arr, cap)
    if n < 0:
        raise F3dError("f3d_plan_exchange failed")
    return [(arr[0][i], (arr[1][i], arr[2][i]), (arr[3][i], arr[4][i])) for i in range(n)]


def plan_resample_source(in_depth, out_depth, out_lo, out_hi):
    lo, hi = C.c_int(), C.c_int()
    check(host().f3d_plan_resample_source(in_depth, out_depth, out_lo, out_hi, C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def comm_unique_id():
    buf = C.create_string_buffer(128)
    check(hip().f3d_comm_unique_id(buf), "f3d_comm_unique_id")
    return buf.raw


def comm_init(unique_id, rank, n_ranks, device=-1):
    check(hip().f3d_init(device), "f3d_init")
    buf = C.create_string_buffer(bytes(unique_id), 128)
    check(hip().f3d_comm_init(buf, rank, n_ranks), "f3d_comm_init")


def comm_destroy():
    hip().f3d_comm_destroy()


def comm_info():
    """what the transport says about itself: RCCL's own rank count / rank / device for the live communicator, bytes and
    exchanges this rank has handed to it"""
    be, n, r, dev = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    sent, ex = C.c_ulonglong(), C.c_ulonglong()
    check(hip().f3d_comm_info(C.byref(be), C.byref(n), C.byref(r), C.byref(dev), C.byref(sent), C.byref(ex)), "f3d_comm_info")
    return {"backend": {0: "none", 1: "rccl", 2: "shm"}.get(be.value, "?"), "ranks": n.value, "rank": r.value,
            "device": dev.value, "sent_bytes": sent.value, "exchanges": ex.value}


def comm_timing(enable):
    """HIP-event timing of the halo exchanges (f3d_comm_timing): on also clears the sums"""
    check(hip().f3d_comm_timing(1 if enable else 0), "f3d_comm_timing")


def comm_timing_read():
    """{class: {count, mean_us, min_us, max_us, mean_bytes_sent}} for the three classes of interval of include/f3d.h"""
    out = {}
    for cls, name in ((0, "blocking_exchange"), (1, "overlapped_exchange_incl_interior"), (2, "grouped_send_recv_alone")):
        us, n, mn, mx, by = C.c_double(), C.c_ulonglong(), C.c_double(), C.c_double(), C.c_ulonglong()
        check(hip().f3d_comm_timing_read(cls, C.byref(us), C.byref(n), C.byref(mn), C.byref(mx), C.byref(by)), "f3d_comm_timing_read")
        out[name] = {"count": n.value, "mean_us": round(us.value / n.value, 2) if n.value else None,
                     "min_us": round(mn.value, 2) if n.value else None, "max_us": round(mx.value, 2) if n.value else None,
                     "mean_bytes_sent": int(by.value / n.value) if n.value else None}
    return out


class SlabOpticalFlow:
    """OpticalFlowSlab: the same solve on n_ranks z-slabs.  local_ranks = [rank] with RCCL (call comm_init first), or
    list(range(n_ranks)) for the one-GPU rehearsal."""

    def __init__(self, n_ranks, local_ranks, halo_capacity=16):
        self._h = C.c_void_p()
        lr = (C.c_int * len(local_ranks))(*local_ranks)
        check(host().f3d_slabflow_create(C.byref(self._h), n_ranks, lr, len(local_ranks), halo_capacity), "f3d_slabflow_create")
        self.dims = None

    def initialize(self, width, height, depth):
        if host().f3d_slabflow_initialize(self._h, width, height, depth) != 0:
            raise F3dError("OpticalFlowSlab::Initialize failed: " + (hip().f3d_last_error() or b"").decode())
        self.dims = (width, height, depth)

    def compute(self, frame_0, frame_1, **kw):
        f0, p0 = _f32(frame_0)
        f1, p1 = _f32(frame_1)
        w, h, d = self.dims
        u, v, ww = (np.zeros((d, h, w), np.float32) for _ in range(3))
        prm = make_params(**kw)
        check(host().f3d_slabflow_compute(self._h, p0, p1, C.byref(prm), u.ctypes.data_as(_fp), v.ctypes.data_as(_fp),
                                          ww.ctypes.data_as(_fp)), "f3d_slabflow_compute")
        return u, v, ww

    def upload(self, frame_0, frame_1):
        f0, p0 = _f32(frame_0)
        f1, p1 = _f32(frame_1)
        check(host().f3d_slabflow_upload(self._h, p0, p1), "f3d_slabflow_upload")

    def compute_resident(self, **kw):
        prm = make_params(**kw)
        secs = C.c_float()
        check(host().f3d_slabflow_compute_resident(self._h, C.byref(prm), C.byref(secs)), "f3d_slabflow_compute_resident")
        return secs.value

    def download(self):
        w, h, d = self.dims
        u, v, ww = (np.zeros((d, h, w), np.float32) for _ in range(3))
        check(host().f3d_slabflow_download(self._h, u.ctypes.data_as(_fp), v.ctypes.data_as(_fp), ww.ctypes.data_as(_fp)))
        return u, v, ww

    def overlapped_iterations(self):
        n = _sz()
        check(host().f3d_slabflow_overlapped_iterations(self._h, C.byref(n)))
        return n.value

    def stage_exchanges(self):
        """exchanges of the last compute made after a solver stage (F3D_SLAB_EXCHANGE=stage)"""
        n = _sz()
        check(host().f3d_slabflow_stage_exchanges(self._h, C.byref(n)))
        return n.value

    def set_exchange_per_stage(self, per_stage):
        """exchange order of the solves that follow: False = once per outer iteration (default), True = after every solver stage"""
        check(host().f3d_slabflow_set_exchange_per_stage(self._h, 1 if per_stage else 0), "f3d_slabflow_set_exchange_per_stage")

    def gathered_warps(self):
        """pyramid levels of the last compute whose warp needed frame 1 gathered beyond the halo room"""
        n = _sz()
        check(host().f3d_slabflow_gathered_warps(self._h, C.byref(n)))
        return n.value

    def batched_exchanges(self):
        """groups of several outer iterations the last compute ran between two exchanges"""
        n = _sz()
        check(host().f3d_slabflow_batched_exchanges(self._h, C.byref(n)))
        return n.value

    def destroy(self):
        if self._h:
            host().f3d_slabflow_destroy(self._h)
            self._h = C.c_void_p()
