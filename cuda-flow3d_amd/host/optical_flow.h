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

  // Frame sequences (bin/flow3d --frames f0 f1 f2 ...).  The reference sets the driver up, uploads both
  // frames, solves, downloads and tears everything down for every pair (src/main.cpp:132-185).  Here the frames live in THREE
  // rotating device containers, so a frame is uploaded once although it serves two pairs, and a solve is split into
  // BeginComputeFlowResident (enqueues the whole pyramid on the library stream and returns) and EndComputeFlowResident (waits):
  // in between the caller reads the next file and uploads it on a copy queue, and downloads / writes the previous pair's flow,
  // which TakeResult() has moved out of the driver's way (three spare containers join the pool the first time).
  bool AllocateSequenceFrames();                                    // the third frame container
  DevicePtr SequenceFrame(int slot) const { return sequence_frame_[slot % 3]; }
  void SelectResidentPair(int slot_0, int slot_1);                  // which two of the three are frame_0 / frame_1 of the next solve
  void BeginComputeFlowResident(OperationParameters& params);
  void EndComputeFlowResident();
  bool TakeResult(DevicePtr (&flow)[3]);                            // the caller owns the three containers until GiveResultBack
  void GiveResultBack(DevicePtr (&flow)[3]);
  // min / max / average magnitude of the flow ComputeFlowResident() left on the device (CudaOperationStat); false when there
  // is none
  bool ResultStatistics(Stat3& stat);
  float LastDeviceSeconds() const { return last_device_seconds_; }

  // Diagnostics per pyramid level and for the result (the reference's counterpart is the disabled
  // "apply the flow and dump the registered volume" block, optical_flow_e.cpp:536-571).  Collected only when asked for: every
  // level costs two small reductions and a read-back.
  struct Residual {
    double rms = 0.0, mean_abs = 0.0;
    float max_abs = 0.f;
  };
  struct LevelStatistics {
    int level = 0;
    DataSize4 size = {0, 0, 0, 0};
    Residual before;  // frame_1 warped with the flow handed down from the coarser level, against frame_0, at this level's size
    Stat3 flow = {0.f, 0.f, 0.f};  // min / max / average flow magnitude after this level's median
  };
  bool collect_level_statistics = false;
  const std::vector<LevelStatistics>& LevelStats() const { return level_stats_; }
  // The reference's debug block made measurable: the ORIGINAL frame_1 registered with the final flow (h = 1) against the
  // original frame_0, and the same difference without any flow.  Needs the resident raw frames and a computed flow.
  bool FinalResidual(Residual& registered, Residual& unregistered);

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
  DevicePtr sequence_frame_[3] = {0, 0, 0};  // sequence mode: resident_frame_ aliases two of these
  size_t extra_containers_ = 0;               // spares allocated for TakeResult
  f3d_event ev_begin_ = nullptr, ev_end_ = nullptr;
  DevicePtr result_flow_[3] = {0, 0, 0};
  float last_device_seconds_ = 0.f;
  std::vector<LevelStatistics> level_stats_;
  bool ResidualOf(DevicePtr frame_0, DevicePtr warped, const DataSize4& size, Residual& out);

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
