"""Test infrastructure: the REFERENCE'S OWN KERNELS on the MI355X.

oracle/Makefile compiles the reference's entire_data .cu files for gfx950 where they lie under /root/reference (hipcc takes them as
HIP source, unmodified; see the recipe's comment for the one include-guard definition it needs) into oracle/_ref/*.hsaco.  This module
loads those code objects with hipModuleLoad and launches their kernels exactly as the reference's operators do -- same block sizes,
same dynamic shared-memory sizes, same argument lists, the `container_size` / `c_Kernel` constants uploaded through
hipModuleGetGlobal the way the operators use cuModuleGetGlobal:

    compute_phi_ksi_3d, solve_3d   cuda_operation_solve.cpp:141-155, 189-254     block 16 x 8 x 4, 8 / 10 extended blocks of shared floats
    median_3d                      cuda_operation_median.cpp:110-146             block 16 x 8 x 4, (16+2h)(8+2h)(4+2h) shared floats
    registration_3d                cuda_operation_registration.cpp:105-131       block 16 x 8 x 4
    resample_{x,y,z}_3d            cuda_operation_resample.cpp:108-175           block 16 x 8 x 8
    convolution{Rows,Columns,Slices}Kernel   cuda_operation_convolution.cpp:190-343    blocks 16x4x4 / 4x16x4 / 4x4x16, 4 result + 2 halo steps
    add_3d                         cuda_operation_add.cpp:81-100                 block 16 x 8 x 4

Only tests import this (the files travel to the GPU box prebuilt; /root/reference does not exist there).  Nothing of the product
depends on it."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
MODULES = ("solve_3d", "median_3d", "registration_3d", "resample_3d", "convolution_3d", "add_3d")


def available():
    return all(os.path.exists(os.path.join(REF_DIR, m + ".hsaco")) for m in MODULES)


class DataSize4(C.Structure):  # src/data_types/data_structs.h:20-25
    _fields_ = [("width", C.c_size_t), ("height", C.c_size_t), ("depth", C.c_size_t), ("pitch", C.c_size_t)]


class RefKernels:
    """kernels of the reference on the containers of one `f3d.Containers` (width, height, depth, pitch in bytes)"""

    def __init__(self, containers):
        self.rt = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        self.size = DataSize4(containers.width, containers.height, containers.depth, containers.pitch)
        self.mods = {}
        self.funcs = {}

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: HIP error {rc}")

    def module(self, name):
        if name not in self.mods:
            mod = C.c_void_p()
            self._check(self.rt.hipModuleLoad(C.byref(mod), os.path.join(REF_DIR, name + ".hsaco").encode()), f"hipModuleLoad {name}")
            self.mods[name] = mod
            self.set_global(name, "container_size", self.size)   # every Initialize of the reference: cuModuleGetGlobal + cuMemcpyHtoD
        return self.mods[name]

    def set_global(self, module, symbol, value):
        mod = self.module(module) if module not in self.mods else self.mods[module]
        dptr, nbytes = C.c_void_p(), C.c_size_t()
        self._check(self.rt.hipModuleGetGlobal(C.byref(dptr), C.byref(nbytes), mod, symbol.encode()), f"hipModuleGetGlobal {symbol}")
        if C.sizeof(value) > nbytes.value:
            raise RuntimeError(f"{symbol}: {C.sizeof(value)} bytes for a symbol of {nbytes.value}")
        self._check(self.rt.hipMemcpyHtoD(dptr, C.byref(value), C.c_size_t(C.sizeof(value))), f"hipMemcpyHtoD {symbol}")

    def launch(self, module, kernel, grid, block, shared_bytes, args):
        """args: ctypes objects in the order of the reference's void* args[]; NULL stream, then wait (the library's launches run
        on a stream of their own: the caller syncs it before handing buffers over)"""
        mod = self.module(module)
        key = (module, kernel)
        if key not in self.funcs:
            fn = C.c_void_p()
            self._check(self.rt.hipModuleGetFunction(C.byref(fn), mod, kernel.encode()), f"hipModuleGetFunction {kernel}")
            self.funcs[key] = fn
        params = (C.c_void_p * len(args))(*[C.cast(C.pointer(a), C.c_void_p) for a in args])
        self._check(self.rt.hipModuleLaunchKernel(self.funcs[key], *[C.c_uint(g) for g in grid], *[C.c_uint(b) for b in block],
                                                  C.c_uint(shared_bytes), None, params, None), f"hipModuleLaunchKernel {kernel}")
        self._check(self.rt.hipDeviceSynchronize(), f"{kernel}: hipDeviceSynchronize")

    @staticmethod
    def _grid(dims, block, steps=(1, 1, 1)):
        return tuple((d + b * s - 1) // (b * s) for d, b, s in zip(dims, block, steps))

    # ---- cuda_operation_solve.cpp ---------------------------------------------------------------------------------------------
    def phi_ksi(self, f0, f1, u, v, w, du, dv, dw, dims, h, eps_s, eps_d, phi, ksi):
        block = (16, 8, 4)
        shared = (block[0] + 2) * (block[1] + 2) * (block[2] + 2) * 4 * 8
        args = [C.c_uint64(p) for p in (f0, f1, u, v, w, du, dv, dw)] + [C.c_size_t(d) for d in dims] + [C.c_float(x) for x in h] + \
               [C.c_float(eps_s), C.c_float(eps_d), C.c_uint64(phi), C.c_uint64(ksi)]
        self.launch("solve_3d", "compute_phi_ksi_3d", self._grid(dims, block), block, shared, args)

    def solve_sweep(self, f0, f1, u, v, w, du, dv, dw, phi, ksi, dims, h, alpha, out_du, out_dv, out_dw):
        block = (16, 8, 4)
        shared = (block[0] + 2) * (block[1] + 2) * (block[2] + 2) * 4 * 10
        args = [C.c_uint64(p) for p in (f0, f1, u, v, w, du, dv, dw, phi, ksi)] + [C.c_size_t(d) for d in dims] + \
               [C.c_float(x) for x in h] + [C.c_float(alpha)] + [C.c_uint64(p) for p in (out_du, out_dv, out_dw)]
        self.launch("solve_3d", "solve_3d", self._grid(dims, block), block, shared, args)

    # ---- cuda_operation_median.cpp --------------------------------------------------------------------------------------------
    def median(self, src, dims, radius, dst):
        block = (16, 8, 4)
        half = radius // 2
        shared = (block[0] + 2 * half) * (block[1] + 2 * half) * (block[2] + 2 * half) * 4
        args = [C.c_uint64(src)] + [C.c_size_t(d) for d in dims] + [C.c_size_t(radius), C.c_uint64(dst)]
        self.launch("median_3d", "median_3d", self._grid(dims, block), block, shared, args)

    # ---- cuda_operation_registration.cpp --------------------------------------------------------------------------------------
    def warp(self, f0, f1, u, v, w, dims, h, dst):
        block = (16, 8, 4)
        args = [C.c_uint64(p) for p in (f0, f1, u, v, w)] + [C.c_size_t(d) for d in dims] + [C.c_float(x) for x in h] + [C.c_uint64(dst)]
        self.launch("registration_3d", "registration_3d", self._grid(dims, block), block, 0, args)

    # ---- cuda_operation_resample.cpp ------------------------------------------------------------------------------------------
    def resample(self, src, dst, tmp, src_dims, dst_dims):
        """the three passes of CudaOperationResample::Execute (:95-105): x into dst, y into tmp, z into dst"""
        block = (16, 8, 8)
        (sw, sh, sd), (W, H, D) = src_dims, dst_dims
        for kernel, a, b, out, n_in in (("resample_x_3d", src, dst, (W, sh, sd), sw), ("resample_y_3d", dst, tmp, (W, H, sd), sh),
                                        ("resample_z_3d", tmp, dst, (W, H, D), sd)):
            args = [C.c_uint64(a), C.c_uint64(b)] + [C.c_size_t(d) for d in out] + [C.c_size_t(n_in)]
            self.launch("resample_3d", kernel, self._grid(out, block), block, 0, args)

    # ---- cuda_operation_convolution.cpp ---------------------------------------------------------------------------------------
    def gaussian(self, src, dst, tmp, dims, taps, radius):
        """c_Kernel upload (:160-161), then rows into dst, columns into tmp, slices into dst (:163-175)"""
        if len(taps) > 51:
            raise ValueError("MAX_KERNEL_LENGTH is 51")
        self.set_global("convolution_3d", "c_Kernel", (C.c_float * len(taps))(*taps))
        pitch = self.size.pitch // 4
        for kernel, block, steps, a, b in (("convolutionRowsKernel", (16, 4, 4), (4, 1, 1), src, dst),
                                           ("convolutionColumnsKernel", (4, 16, 4), (1, 4, 1), dst, tmp),
                                           ("convolutionSlicesKernel", (4, 4, 16), (1, 1, 4), tmp, dst)):
            shared = block[0] * block[1] * block[2] * (4 + 2) * 4
            args = [C.c_uint64(b), C.c_uint64(a)] + [C.c_int(d) for d in dims] + [C.c_int(pitch), C.c_int(radius)]
            self.launch("convolution_3d", kernel, self._grid(dims, block, steps), block, shared, args)

    # ---- cuda_operation_add.cpp -----------------------------------------------------------------------------------------------
    def add(self, a, b, dims):
        block = (16, 8, 4)
        args = [C.c_uint64(a), C.c_uint64(b)] + [C.c_size_t(d) for d in dims]
        self.launch("add_3d", "add_3d", self._grid(dims, block), block, 0, args)

    def close(self):
        for mod in self.mods.values():
            self.rt.hipModuleUnload(mod)
        self.mods, self.funcs = {}, {}
