#include "hip_utils.h"

#include <cmath>

#include <cstdlib>

bool InitDeviceContextWithFirstAvailableDevice()
{
  int count = 0;
  if (CheckDeviceError(f3d_device_count(&count))) return false;
  if (count == 0) {
    std::printf("There are no HIP capable devices.");
    return false;
  }
  if (CheckDeviceError(f3d_init(-1))) return false;
  char name[128];
  if (CheckDeviceError(f3d_device_name(name, sizeof(name)))) return false;
  std::printf("HIP Device: %s. Launch timeout: %s\n", name, "No");
  return true;
}

void CopyData3DtoDevice(Data3D& data3d, DevicePtr device_ptr, size_t device_height, size_t device_pitch)
{
  CheckDeviceError(f3d_copy3d_h2d(device_ptr, device_pitch, device_height, 0, data3d.DataPtr(), data3d.Width(),
                                  data3d.Height(), data3d.Depth()));
}

void CopyData3DFromDevice(DevicePtr device_ptr, Data3D& data3d, size_t device_height, size_t device_pitch)
{
  CheckDeviceError(f3d_copy3d_d2h(data3d.DataPtr(), data3d.Width(), data3d.Height(), data3d.Depth(), device_ptr,
                                  device_pitch, device_height, 0));
}

namespace {
// The fused launches apply the face weights alpha / h^2 by selection, which equals the reference's (float)(flag) * w only for a
// finite w that is not negative (include/f3d.h); a solve with other parameters -- the reference would propagate NaN or -0 --
// takes the one-sweep launches, whose kernels multiply as the reference does.  Per thread: drivers on lanes solve side by side.
thread_local bool g_solve_weights_plain = true;
}  // namespace

void NoteSolveWeights(float equation_alpha, float hx, float hy, float hz)
{
  g_solve_weights_plain = true;
  for (float h : {hx, hy, hz}) {
    const float w = equation_alpha / (h * h);   // the kernels' own expression (solve_3d.cu:437-439)
    if (!(w - w == 0.f) || std::signbit(w)) g_solve_weights_plain = false;
  }
}

bool FusedSweepsEnabled()
{
  static const bool on = [] {
    const char* e = std::getenv("F3D_FUSED_SWEEPS");
    return !(e && e[0] == '0');
  }();
  return on && g_solve_weights_plain;
}

bool FusedPhiKsiEnabled()
{
  static const bool on = [] {
    const char* e = std::getenv("F3D_FUSED_PHI_KSI");
    return !(e && e[0] == '0');
  }();
  return on;
}

bool FrameDerivativesEnabled()
{
  static const bool on = [] {
    // on since the end of round 3 (the centre-only inputs got a ring of their own and the 12-row tile fits): two sweeps -3.5 % at 512^3,
    // a 128^3 solve -3.4 %, a 512^3 solve +0.8 %; the operator falls back to the frame builds where its four extra volumes do not fit
    const char* e = std::getenv("F3D_FRAME_DERIVATIVES");
    return !(e && e[0] == '0');
  }();
  return on;
}

bool ThreeStageLaunchesPay(size_t width, size_t height, size_t depth)
{
  // The three-stage launches (f3d_solve_sweep3, f3d_solve_sweep2_phi_ksi: k_tri) are bit-identical to the two-stage schedule and were
  // built to halve the launches of the small levels.  Measured (profiles/r04_three_stage_kbench.txt, r04_three_stage_solves.txt), they
  // do NOT pay: a z-chunk of a three-stage march runs its planes + 4 steps where a two-stage march runs planes + 2, and on the levels
  // where launches dominate a chunk is one plane -- five steps against three -- so two three-stage launches cost what three two-stage
  // launches cost (24^3: 14.0 + 15.6 us against 2 x 9.7 + 10.3 us), and from ~96^3 up the 56-of-64 tiling and the four halo rows lose
  // outright (BASELINE config 2: 73.5 -> 82.9 ms).  OFF by default; F3D_TRI=1 takes them on levels of up to F3D_TRI_MAX_VOXELS voxels
  // (default 3e6) -- the tests run both schedules and demand the same bits.  Read once.
  static const bool on = [] {
    const char* e = std::getenv("F3D_TRI");
    return e && e[0] == '1';
  }();
  static const double max_voxels = [] {
    const char* e = std::getenv("F3D_TRI_MAX_VOXELS");
    return e ? std::atof(e) : 3.0e6;
  }();
  return on && static_cast<double>(width) * static_cast<double>(height) * static_cast<double>(depth) <= max_voxels;
}
