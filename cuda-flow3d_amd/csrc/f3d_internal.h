// Internal glue shared by the translation units of libf3d_hip.so (not installed, not part of the C ABI).
#ifndef F3D_INTERNAL_H_
#define F3D_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>

#include "f3d.h"

// Geometry handed to every kernel by value: level dims (global depth), container strides, slab window.
struct F3dGeo {
  int W, H, D;
  int Hc, pitch;  // container height (rows per plane) and row pitch in floats
  int z_base, z_lo, z_hi;
};

namespace f3d {

int fail(const char* fmt, ...);          // records the thread-local error text, returns 1
int hip_fail(hipError_t e, const char* what, const char* file, int line);
hipStream_t stream();                    // library stream (valid after f3d_init)
bool ready();
const f3d_size4& container();
bool make_geo(F3dGeo* g, size_t w, size_t h, size_t d, const f3d_slab* slab, const char* who);
// conv taps live in host memory and are passed to the kernels by value
struct ConvTaps { float k[51]; int count; };
const ConvTaps& conv_taps();

// per-kernel event timing
void prof_begin(int kernel, size_t voxels);
void prof_end(int kernel);

}  // namespace f3d

#define F3D_HIP(call)                                                        \
  do {                                                                       \
    hipError_t f3d_e_ = (call);                                              \
    if (f3d_e_ != hipSuccess) return f3d::hip_fail(f3d_e_, #call, __FILE__, __LINE__); \
  } while (0)

#define F3D_REQUIRE_READY(who)                                               \
  do {                                                                       \
    if (!f3d::ready()) return f3d::fail("%s: f3d_init() has not been called", who); \
  } while (0)

template <typename T>
static inline T* f3d_ptr(f3d_devptr p) { return reinterpret_cast<T*>(static_cast<uintptr_t>(p)); }

// ---- device helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ size_t f3d_row(const F3dGeo& g, int y, int z)
{
  return (static_cast<size_t>(z - g.z_base) * static_cast<size_t>(g.Hc) + static_cast<size_t>(y)) *
         static_cast<size_t>(g.pitch);
}
// mirror index of the reference halo loads (solve_3d.cu:73-75,89-90; median_3d.cu:70-72)
__device__ __forceinline__ int f3d_mir(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - i - 2 : i); }
__device__ __forceinline__ int f3d_clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

#endif
