/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * C bridge onto the two operators of the reference whose Execute() is plain host C++ (no kernel, no driver call on the
 * path taken here), compiled IN PLACE from /root/reference by oracle/Makefile into oracle/_ref/libf3d_ref_ops.so:
 *   src/cuda_operations/partial_data/cuda_operation_register_p.cpp:96-139   the trilinear backward warp ("CPU Version")
 *   src/cuda_operations/partial_data/cuda_operation_stat_p.cpp:85-104       min / max / avg flow magnitude
 * Their translation units also contain calls into the CUDA driver (cuModuleLoad in Initialize, cuModuleUnload in
 * Destroy).  Nothing here stands in for those: the symbols stay undefined in this library, it is opened with lazy binding
 * by ref_bridge.cpp, and the code below never reaches them -- the warp operator is marked initialised through its
 * protected flag instead of Initialize() (which would load a PTX module), and with a null module handle Destroy() makes
 * no driver call.
 */
#include <cstddef>
#include <cstring>

#include "src/cuda_operations/partial_data/cuda_operation_register_p.h"
#include "src/cuda_operations/partial_data/cuda_operation_stat_p.h"
#include "src/data_types/data3d.h"
#include "src/data_types/data_structs.h"
#include "src/data_types/operation_parameters.h"

namespace {
struct HostWarp : CudaOperationRegistrationP {
  HostWarp() { initialized_ = true; }
};
void fill(Data3D& d, const float* src, size_t n) { std::memcpy(d.DataPtr(), src, n * sizeof(float)); }
}  // namespace

extern "C" {

/* out = frame_1 warped by (u, v, w) towards frame_0, exactly as CudaOperationRegistrationP::Execute leaves it in frame_1 */
int refops_warp(const float* f0, const float* f1, const float* u, const float* v, const float* w, size_t W, size_t H, size_t D,
                float hx, float hy, float hz, float* out)
{
  const size_t n = W * H * D;
  Data3D frame_0(W, H, D), frame_1(W, H, D), flow_u(W, H, D), flow_v(W, H, D), flow_w(W, H, D), temp(W, H, D);
  fill(frame_0, f0, n);
  fill(frame_1, f1, n);
  fill(flow_u, u, n);
  fill(flow_v, v, n);
  fill(flow_w, w, n);
  DataSize4 data_size = {W, H, D, 0};
  size_t max_mag = 0;
  OperationParameters bag;
  bag.PushValuePtr("frame_0", &frame_0);
  bag.PushValuePtr("frame_1", &frame_1);
  bag.PushValuePtr("flow_u", &flow_u);
  bag.PushValuePtr("flow_v", &flow_v);
  bag.PushValuePtr("flow_w", &flow_w);
  bag.PushValuePtr("temp", &temp);
  bag.PushValuePtr("hx", &hx);
  bag.PushValuePtr("hy", &hy);
  bag.PushValuePtr("hz", &hz);
  bag.PushValuePtr("data_size", &data_size);
  bag.PushValuePtr("max_mag", &max_mag);
  HostWarp op;
  op.Execute(bag);
  std::memcpy(out, frame_1.DataPtr(), n * sizeof(float));
  return 1;
}

int refops_flow_stats(const float* u, const float* v, const float* w, size_t W, size_t H, size_t D, float* min_mag, float* max_mag,
                      float* avg)
{
  const size_t n = W * H * D;
  Data3D flow_u(W, H, D), flow_v(W, H, D), flow_w(W, H, D);
  fill(flow_u, u, n);
  fill(flow_v, v, n);
  fill(flow_w, w, n);
  DataSize4 data_size = {W, H, D, 0};
  Stat3 stat = {0.f, 0.f, 0.f};
  OperationParameters bag;
  bag.PushValuePtr("flow_u", &flow_u);
  bag.PushValuePtr("flow_v", &flow_v);
  bag.PushValuePtr("flow_w", &flow_w);
  bag.PushValuePtr("data_size", &data_size);
  bag.PushValuePtr("stat", &stat);
  CudaOperationStatP op;
  if (!op.Initialize(nullptr)) return 0;   /* makes no driver call: cuda_operation_stat_p.cpp:33-55 */
  op.Execute(bag);
  *min_mag = stat.min;
  *max_mag = stat.max;
  *avg = stat.avg;
  return 1;
}

}  // extern "C"
