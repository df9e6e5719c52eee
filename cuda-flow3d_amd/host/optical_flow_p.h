// Out-of-core flow driver: every volume stays in host memory and the piecemeal operators (operations_p.h) stream
// z-chunks of it through the GPU, so the device only ever holds what its free memory allows.  Interface, name string,
// operator sequence and parameter keys of src/optical_flow/optical_flow_p.{h,cpp}: like the reference's piecemeal driver
// it runs NO Gaussian pre-blur and NO median (optical_flow_p.cpp:268-302 is commented out there), so its result equals
// OpticalFlowE's with gaussian_sigma <= 0 and median_radius = 1, bit for bit.
#ifndef F3D_HOST_OPTICAL_FLOW_P_H_
#define F3D_HOST_OPTICAL_FLOW_P_H_

#include <vector>

#include "operations_p.h"
#include "optical_flow.h"

class OpticalFlowP : public OpticalFlowBase {
 public:
  OpticalFlowP();
  ~OpticalFlowP() override;

  bool Initialize(const DataSize4& data_size) override;
  void ComputeFlow(Data3D& frame_0, Data3D& frame_1, Data3D& flow_u, Data3D& flow_v, Data3D& flow_w,
                   OperationParameters& params) override;
  void Destroy() override;

  bool silent = false;
  // Page-lock the host volumes for the duration of ComputeFlow (full link rate, asynchronous copies); F3D_P_PIN=0 or
  // this flag turns it off.
  bool pin_host_memory = true;

  float LastDeviceSeconds() const { return last_device_seconds_; }
  // solver residencies of the last ComputeFlow: levels that fitted the budget count one pass each
  size_t LastSolvePasses() const { return solve_passes_; }
  size_t LastStreamedLevels() const { return streamed_levels_; }
  // wall seconds the last ComputeFlow spent in {frame resample, flow resample, registration, solve, add}
  const double* LastOperatorSeconds() const { return op_seconds_; }

 private:
  DataSize4 data_size_ = {0, 0, 0, 0};
  float last_device_seconds_ = 0.f;
  size_t solve_passes_ = 0, streamed_levels_ = 0;
  double op_seconds_[5] = {0, 0, 0, 0, 0};

  CudaOperationRegistrationP cuop_register_p_;
  CudaOperationResampleP cuop_resample_p_;
  CudaOperationSolveP cuop_solve_p_;
  CudaOperationStatP cuop_stat_p_;
  CudaOperationAddP cuop_add_p_;
  std::vector<CudaOperationBase*> cuda_operations_;
};

#endif
