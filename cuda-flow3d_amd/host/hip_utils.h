// Device helpers of the host side: error reporting convention, context bring-up and the dense-host <->
// pitched-device volume copies (behaviour of src/utils/cuda_utils.{h,cpp} on top of the f3d C ABI).
#ifndef F3D_HOST_HIP_UTILS_H_
#define F3D_HOST_HIP_UTILS_H_

#include <cstdio>

#include "data_types.h"
#include "f3d.h"

// Same convention as the reference's CheckCudaError (src/utils/cuda_utils.h:26-46): prints to stderr and
// returns TRUE ON ERROR, so callers write `if (!CheckDeviceError(call)) { ...ok... }`.
#define CheckDeviceError(status) CheckDeviceErrorAt((status), __FILE__, __LINE__)
inline bool CheckDeviceErrorAt(int status, const char* file, int line)
{
  if (status != 0) {
    std::fprintf(stderr, "Device API error = %04d\n%s\n from file <%s>, line %i.\n", status, f3d_last_error(), file, line);
    return true;
  }
  return false;
}

// A profiler range for the lifetime of the object (f3d_range_push / f3d_range_pop: roctx, only when a profiler is attached).
class ProfilerRange {
 public:
  explicit ProfilerRange(const char* name) { f3d_range_push(name); }
  ~ProfilerRange() { f3d_range_pop(); }
  ProfilerRange(const ProfilerRange&) = delete;
  ProfilerRange& operator=(const ProfilerRange&) = delete;
};

// cuda_utils.cpp:21-57: pick the first device, print its name, create the context.
bool InitDeviceContextWithFirstAvailableDevice();
void CopyData3DtoDevice(Data3D& data3d, DevicePtr device_ptr, size_t device_height, size_t device_pitch);
void CopyData3DFromDevice(DevicePtr device_ptr, Data3D& data3d, size_t device_height, size_t device_pitch);
// Inner sweeps are launched in fused pairs unless F3D_FUSED_SWEEPS=0 (A/B timing; the results are bit-identical).
bool FusedSweepsEnabled();
// called by every solver driver with the parameters of the solve it is about to launch: alpha / h^2 that is not finite or negative
// switches the fused launches off for this thread until the next call (they select where the reference multiplies)
void NoteSolveWeights(float equation_alpha, float hx, float hy, float hz);
// The last sweep of an outer iteration and the phi/ksi of the next one are one launch unless F3D_FUSED_PHI_KSI=0.
bool FusedPhiKsiEnabled();
// F3D_FRAME_DERIVATIVES=1: the fused launches read frame derivatives computed once per level instead of the frames.  Off by
// default: measured on the MI355X it trades a seventh of the stage-1 arithmetic for a fifth more bytes through the loader and
// comes out even; the launchers stay tested because they need 36 registers fewer.
bool FrameDerivativesEnabled();
// small and mid-size levels: an outer iteration as (sweep, sweep, sweep) + (sweep, sweep, next phi/ksi) -- two launches instead of three
bool ThreeStageLaunchesPay(size_t width, size_t height, size_t depth);

#endif
