// Flow drivers of the drop-in surface (interfaces of src/optical_flow/optical_flow_base.h:24-45 and
// src/optical_flow/optical_flow_e.h:38-66), re-implemented on the operator layer in operations.h.
#ifndef F3D_HOST_OPTICAL_FLOW_H_
#define F3D_HOST_OPTICAL_FLOW_H_

#include <vector>

#include "data_types.h"
#include "operations.h"

// One level of the coarse-to-fine pyramid: size and grid spacing in original-voxel units.
struct PyramidLevel {
  DataSize4 size;
  float hx, hy, hz;
};

class OpticalFlowBase {
 public:
  const char* GetName() const { return name_; }

  virtual bool Initialize(const DataSize4& data_size) = 0;
  virtual void ComputeFlow(Data3D& frame_0, Data3D& frame_1, Data3D& flow_u, Data3D& flow_v, Data3D& flow_w,
                           OperationParameters& params);
  virtual void Destroy();
  virtual ~OpticalFlowBase();

  // Pyramid depth rule and per-level geometry (optical_flow_base.cpp:31-56, optical_flow_e.cpp:262-268);
  // public and static so the slab planner and the tests can use them without a device.
  static size_t GetMaxWarpLevel(size_t width, size_t height, size_t depth, float scale_factor);
  static PyramidLevel GetLevel(const DataSize4& original, float scale_factor, int level);

 protected:
  explicit OpticalFlowBase(const char* name) : name_(name) {}
  bool IsInitialized() const;

  bool initialized_ = false;

 private:
  const char* name_ = nullptr;
};

// Everything resident on one GPU: 15 pitched containers of the original size, pre-blur, then per level
// {resample frames, upsample flow, warp, solve, add, median}.
class OpticalFlowE : public OpticalFlowBase {
 public:
  OpticalFlowE();
  ~OpticalFlowE() override;

  bool Initialize(const DataSize4& data_size) override;
  void ComputeFlow(Data3D& frame_0, Data3D& frame_1, Data3D& flow_u, Data3D& flow_v, Data3D& flow_w,
                   OperationParameters& params) override;
  void Destroy() override;

  bool silent = false;

  // Device-resident variant (benchmarks, frame sequences): two extra containers hold the raw frames in HBM,
  // ComputeFlowResident() runs the same pyramid from them and leaves (u, v, w) on the device until
  // DownloadFlow().  The raw frames are never overwritten, so the call can be repeated.
  bool AllocateResidentFrames();
  void UploadResidentFrames(Data3D& frame_0, Data3D& frame_1);
  DevicePtr ResidentFrame(int which) const { return resident_frame_[which ? 1 : 0]; }
  const DataSize4& ContainerSize() const { return dev_container_size_; }
  void ComputeFlowResident(OperationParameters& params);
  void DownloadFlow(Data3D& flow_u, Data3D& flow_v, Data3D& flow_w);
  // min / max / average magnitude of the flow ComputeFlowResident() left on the device (CudaOperationStat); false when there
  // is none
  bool ResultStatistics(Stat3& stat);
  float LastDeviceSeconds() const { return last_device_seconds_; }

 private:
  static constexpr size_t kContainers = 15;  // optical_flow_e.h:40

  bool InitCudaMemory();
  bool InitCudaOperations();
  DevicePtr Borrow();
  void GiveBack(DevicePtr p);
  bool RunPyramid(OperationParameters& params, DevicePtr raw_0, DevicePtr raw_1, bool raw_is_pooled);
  void ReleaseResult();

  DataSize4 dev_container_size_ = {0, 0, 0, 0};
  std::vector<DevicePtr> free_containers_;  // LIFO like the reference's std::stack
  DevicePtr resident_frame_[2] = {0, 0};
  DevicePtr result_flow_[3] = {0, 0, 0};
  float last_device_seconds_ = 0.f;

  CudaOperationAdd cuop_add_;
  CudaOperationMedian cuop_median_;
  CudaOperationConvolution3D cuop_convolution_;
  CudaOperationRegistration cuop_register_;
  CudaOperationResample cuop_resample_;
  CudaOperationSolve cuop_solve_;
  CudaOperationStat cuop_stat_;  // not in the list below: the reference's single-GPU driver has six operators
  std::vector<CudaOperationBase*> cuda_operations_;
};

#endif
