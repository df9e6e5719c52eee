/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin C bridge onto the CUDA-free parts of the reference, compiled IN PLACE from /root/reference by
 * oracle/Makefile into oracle/_ref/libf3d_ref.so (git-ignored; never copied into this repo):
 *   src/optical_flow/optical_flow_base.cpp   -> GetMaxWarpLevel (level schedule)
 *   src/data_types/operation_parameters.cpp  -> the string-keyed parameter bag (first push wins)
 *   src/data_types/data3d.cpp                -> RAW U8/F32 + VTK volume I/O   (needs <cuda.h> types only;
 *                                               the image carries that header under triton/backends/nvidia)
 * The reference's kernels (src/kernels/*.cu) and operator hosts call into nvcc/the CUDA driver and are
 * NOT buildable here; nothing in this file stands in for them.
 * Two operators whose Execute() is plain host C++ live in a second library (ref_ops_bridge.cpp ->
 * oracle/_ref/libf3d_ref_ops.so).  Their translation units reference three CUDA driver symbols that are never reached;
 * the library is therefore opened HERE with lazy binding (Python's ctypes binds eagerly and could not load it).
 */
#include <dlfcn.h>

#include <cstddef>
#include <cstdio>
#include <cstring>
#include <string>

#include "src/optical_flow/optical_flow_base.h"
#include "src/data_types/operation_parameters.h"
#ifdef F3D_REF_HAVE_DATA3D
#include "src/data_types/data3d.h"
#endif

namespace {
class LevelProbe : public OpticalFlowBase {
 public:
  LevelProbe() : OpticalFlowBase("level probe") {}
  bool Initialize(const DataSize4&) override { return true; }
  size_t MaxLevel(size_t w, size_t h, size_t d, float sf) const { return GetMaxWarpLevel(w, h, d, sf); }
};
}  // namespace

extern "C" {

size_t ref_max_warp_level(size_t w, size_t h, size_t d, float sf)
{
  LevelProbe probe;
  return probe.MaxLevel(w, h, d, sf);
}

/* returns 1 when a second push under the same key was refused and the first value is kept */
int ref_params_first_push_wins(void)
{
  OperationParameters bag;
  int a = 1, b = 2;
  bool first = bag.PushValuePtr("key", &a);
  bool second = bag.PushValuePtr("key", &b);
  int kept = *static_cast<int*>(bag.GetValuePtr("key"));
  bool missing_is_null = bag.GetValuePtr("absent") == nullptr;
  bag.Clear();
  bool cleared = bag.GetValuePtr("key") == nullptr;
  return first && !second && kept == 1 && missing_is_null && cleared;
}

int ref_have_data3d(void)
{
#ifdef F3D_REF_HAVE_DATA3D
  return 1;
#else
  return 0;
#endif
}

#ifdef F3D_REF_HAVE_DATA3D
int ref_read_raw_u8(const char* path, size_t w, size_t h, size_t d, float* out)
{
  Data3D vol;
  if (!vol.ReadRAWFromFileU8(path, w, h, d)) return 0;
  std::memcpy(out, vol.DataPtr(), w * h * d * sizeof(float));
  return 1;
}

int ref_read_raw_f32(const char* path, size_t w, size_t h, size_t d, float* out)
{
  Data3D vol;
  if (!vol.ReadRAWFromFileF32(path, w, h, d)) return 0;
  std::memcpy(out, vol.DataPtr(), w * h * d * sizeof(float));
  return 1;
}

int ref_write_raw(const char* path, const float* in, size_t w, size_t h, size_t d, int as_u8)
{
  Data3D vol(w, h, d);
  std::memcpy(vol.DataPtr(), in, w * h * d * sizeof(float));
  return as_u8 ? vol.WriteRAWToFileU8(path) : vol.WriteRAWToFileF32(path);
}

int ref_write_vtk(const char* path, const float* u, const float* v, const float* w_, size_t w, size_t h, size_t d)
{
  Data3D a(w, h, d), b(w, h, d), c(w, h, d);
  std::memcpy(a.DataPtr(), u, w * h * d * sizeof(float));
  std::memcpy(b.DataPtr(), v, w * h * d * sizeof(float));
  std::memcpy(c.DataPtr(), w_, w * h * d * sizeof(float));
  return Data3D::WriteFlowToFileVTK(path, a, b, c);
}
#endif

/* ---- the host-only operators of libf3d_ref_ops.so, opened lazily ---- */
static void* ops_lib(void)
{
  static void* handle = nullptr;
  static bool tried = false;
  if (!tried) {
    tried = true;
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&ops_lib), &info) && info.dli_fname) {
      std::string path(info.dli_fname);
      const size_t slash = path.rfind('/');
      path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/libf3d_ref_ops.so";
      handle = dlopen(path.c_str(), RTLD_LAZY | RTLD_LOCAL);
    }
  }
  return handle;
}

int ref_have_host_ops(void) { return ops_lib() != nullptr; }

int ref_warp(const float* f0, const float* f1, const float* u, const float* v, const float* w, size_t W, size_t H, size_t D, float hx,
             float hy, float hz, float* out)
{
  typedef int (*fn_t)(const float*, const float*, const float*, const float*, const float*, size_t, size_t, size_t, float, float,
                      float, float*);
  void* lib = ops_lib();
  fn_t fn = lib ? reinterpret_cast<fn_t>(dlsym(lib, "refops_warp")) : nullptr;
  return fn ? fn(f0, f1, u, v, w, W, H, D, hx, hy, hz, out) : 0;
}

int ref_flow_stats(const float* u, const float* v, const float* w, size_t W, size_t H, size_t D, float* mn, float* mx, float* avg)
{
  typedef int (*fn_t)(const float*, const float*, const float*, size_t, size_t, size_t, float*, float*, float*);
  void* lib = ops_lib();
  fn_t fn = lib ? reinterpret_cast<fn_t>(dlsym(lib, "refops_flow_stats")) : nullptr;
  return fn ? fn(u, v, w, W, H, D, mn, mx, avg) : 0;
}

}  // extern "C"
