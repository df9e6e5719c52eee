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
  // host levels whose frame 1 was registered inside the solver's first residency (CudaOperationSolveP::register_frame_1; F3D_P_FUSED_WARP=0: none)
  size_t LastLevelsRegisteredInside() const { return levels_registered_inside_; }
  // host levels whose solver kept the two frames and u, v, w on the device for the whole level (SolvePiecemealPlan::constants_on_device)
  size_t LastLevelsWithConstantsOnDevice() const { return levels_with_constants_; }
  // coarse levels of the last ComputeFlow that ran entirely on the device (see resident_coarse_levels)
  size_t LastResidentLevels() const { return resident_levels_; }
  // whether the coarsest of those levels resampled their frames from device copies of the two originals (phase A)
  bool LastOriginalsOnDevice() const { return originals_on_device_; }
  // wall seconds the last ComputeFlow spent in {frame resample, flow resample, registration, solve, add} of the levels that
  // went through the host, and in the resident coarse levels as a whole
  const double* LastOperatorSeconds() const { return op_seconds_; }
  // Run the coarse levels whose whole working set (14 fields) fits the device budget (97 %, and room for the frame streaming arena) with the "entire data"
  // operators, flow and increments staying on the device from level to level; only the two original frames stream through
  // (they are resampled straight into device containers).  The flow goes to the host once, when the first level that does
  // not fit is reached.  Same kernels, same result; F3D_P_RESIDENT=0 or this flag sends every level through the host.
  bool resident_coarse_levels = true;
  // Beyond the reference's piecemeal driver: also apply the Gaussian pre-blur and the per-level median of the flow, i.e.
  // reproduce OpticalFlowE's whole pipeline (bit for bit) on volumes that do not fit the device.  Off by default, which is
  // the reference's behaviour; F3D_P_FULL=1 turns it on as well.  Costs two more host volumes (the blurred frames).
  bool full_pipeline = false;

 private:
  DataSize4 data_size_ = {0, 0, 0, 0};
  float last_device_seconds_ = 0.f;
  // levels first_level .. last_level (descending) on the device; false on a device error
  bool RunResidentLevels(Data3D& frame_0, Data3D& frame_1, Data3D& flow_u, Data3D& flow_v, Data3D& flow_w, OperationParameters& params,
                         int first_level, int last_level, size_t container_bytes, bool originals_on_device,
                         const DataSize4& carried_flow_size, size_t median_radius);

  size_t solve_passes_ = 0, streamed_levels_ = 0, resident_levels_ = 0, levels_registered_inside_ = 0, levels_with_constants_ = 0;
  bool originals_on_device_ = false;
  double op_seconds_[6] = {0, 0, 0, 0, 0, 0};

  // "entire data" operators for the resident coarse levels (not in the list Initialize prints: the reference's piecemeal
  // driver has five operators)
  CudaOperationResample cuop_resample_e_;
  CudaOperationRegistration cuop_register_e_;
  CudaOperationSolve cuop_solve_e_;
  CudaOperationAdd cuop_add_e_;
  CudaOperationMedian cuop_median_e_;

  CudaOperationRegistrationP cuop_register_p_;
  CudaOperationResampleP cuop_resample_p_;
  CudaOperationSolveP cuop_solve_p_;
  CudaOperationStatP cuop_stat_p_;
  CudaOperationAddP cuop_add_p_;
  CudaOperationConvolution3DP cuop_convolution_p_;  // full_pipeline only
  CudaOperationMedianP cuop_median_p_;
  std::vector<CudaOperationBase*> cuda_operations_;
};

#endif
