// Operator layer of the drop-in surface: the abstract operator and the six "entire data" operators with
// the reference's class names, method signatures, name strings and parameter keys
// (src/cuda_operations/cuda_operation_base.h:24-48 and src/cuda_operations/entire_data/*.h), re-implemented on
// the f3d C ABI (include/f3d.h) instead of cuModuleLoad/cuLaunchKernel.  The class names keep the reference's
// "Cuda" prefix on purpose: callers written against the reference compile unchanged.
#ifndef F3D_HOST_OPERATIONS_H_
#define F3D_HOST_OPERATIONS_H_

#include "data_types.h"

class CudaOperationBase {
 public:
  const char* GetName() const { return name_; }

  virtual bool Initialize(const OperationParameters* params = nullptr) = 0;
  virtual void Execute(OperationParameters& params);
  virtual void Destroy();
  virtual ~CudaOperationBase();

  // optional z-slab window applied to every launch of this operator (multi-GPU driver); nullptr = whole volume
  void SetSlab(const f3d_slab* slab) { slab_ = slab; }

 protected:
  explicit CudaOperationBase(const char* name) : name_(name) {}
  bool IsInitialized() const;
  // shared Initialize body: read "container_size" and hand it to the device library
  bool InitializeContainer(const OperationParameters* params);

  DataSize4 dev_container_size_ = {0, 0, 0, 0};
  bool initialized_ = false;
  const f3d_slab* slab_ = nullptr;

 private:
  const char* name_ = nullptr;
};

// operand_0 += operand_1                      keys: operand_0, operand_1, data_size
class CudaOperationAdd : public CudaOperationBase {
 public:
  CudaOperationAdd() : CudaOperationBase("CUDA Add") {}
  bool Initialize(const OperationParameters* params = nullptr) override { return InitializeContainer(params); }
  void Execute(OperationParameters& params) override;
  // `count` bags (one per flow component, keys as for Execute) in ONE launch where they describe the same box (f3d_add_n);
  // anything else -- more than three bags, differing sizes -- runs them one after the other through Execute.  Same results.
  void ExecuteBatch(OperationParameters* params, size_t count);
};

// min / max / average flow magnitude        keys: dev_flow_u, dev_flow_v, dev_flow_w, data_size, stat (Stat3*)
// (the reference's CudaOperationStatP, cuda_operation_stat_p.cpp:44-107, works on downloaded host volumes)
class CudaOperationStat : public CudaOperationBase {
 public:
  CudaOperationStat() : CudaOperationBase("CUDA Stat") {}
  bool Initialize(const OperationParameters* params = nullptr) override { return InitializeContainer(params); }
  void Execute(OperationParameters& params) override;
  bool silent = true;
};

// separable Gaussian, rows -> columns -> slices   keys: dev_input, dev_output, dev_temp, data_size, gaussian_sigma
class CudaOperationConvolution3D : public CudaOperationBase {
 public:
  CudaOperationConvolution3D() : CudaOperationBase("CUDA Convolution 3D") {}
  bool Initialize(const OperationParameters* params = nullptr) override { return InitializeContainer(params); }
  void Execute(OperationParameters& params) override;

  void ComputeGaussianKernel(float sigma, size_t precision, float pixel_size);
  void PrintConvolutionKernel() const;
  size_t KernelRadius() const { return kernel_radius_; }
  const float* Kernel() const { return kernel_; }

 private:
  static constexpr size_t kMaxKernelLength = 51;  // MAX_KERNEL_LENGTH, convolution_3d.cu:49
  float kernel_[kMaxKernelLength] = {0};
  size_t kernel_radius_ = 0;
  size_t kernel_length_ = 0;
};

// 3-D median, "radius" is the window diameter    keys: dev_input, dev_output, data_size, radius
class CudaOperationMedian : public CudaOperationBase {
 public:
  CudaOperationMedian() : CudaOperationBase("CUDA Median") {}
  bool Initialize(const OperationParameters* params = nullptr) override { return InitializeContainer(params); }
  void Execute(OperationParameters& params) override;
  // `count` bags in one launch (f3d_median_n) where box and window agree and no input is another bag's output; else one by one
  void ExecuteBatch(OperationParameters* params, size_t count);
};

// backward trilinear warp    keys: dev_frame_0, dev_frame_1, dev_flow_u/v/w, dev_output, data_size, hx, hy, hz
class CudaOperationRegistration : public CudaOperationBase {
 public:
  CudaOperationRegistration() : CudaOperationBase("CUDA Registration") {}
  bool Initialize(const OperationParameters* params = nullptr) override { return InitializeContainer(params); }
  void Execute(OperationParameters& params) override;
};

// separable area resample X (in->out), Y (out->temp), Z (temp->out)
// keys: dev_input, dev_output, dev_temp, data_size, resample_size
class CudaOperationResample : public CudaOperationBase {
 public:
  CudaOperationResample() : CudaOperationBase("CUDA Resample") {}
  bool Initialize(const OperationParameters* params = nullptr) override { return InitializeContainer(params); }
  void Execute(OperationParameters& params) override;
  // `count` bags with the same data_size / resample_size in three launches instead of 3 x count (f3d_resample_{x,y,z}_n): every
  // bag needs a dev_temp of its own, and no dev_input / dev_temp / dev_output may appear in two roles of the batch; else one by one
  void ExecuteBatch(OperationParameters* params, size_t count);

 private:
  void ResampleX(DevicePtr input, DevicePtr output, DataSize4& input_size, DataSize4& output_size) const;
  void ResampleY(DevicePtr input, DevicePtr output, DataSize4& input_size, DataSize4& output_size) const;
  void ResampleZ(DevicePtr input, DevicePtr output, DataSize4& input_size, DataSize4& output_size) const;
};

// lagged-nonlinearity solver: outer x (phi/ksi + inner x sweep)
// keys: dev_frame_0/1, dev_flow_u/v/w, dev_phi, dev_ksi, dev_flow_du/dv/dw (by pointer), dev_temp_du/dv/dw (by
// pointer), outer_iterations_count, inner_iterations_count, equation_alpha/smoothness/data, hx, hy, hz, data_size
class CudaOperationSolve : public CudaOperationBase {
 public:
  CudaOperationSolve() : CudaOperationBase("CUDA Solve") {}
  ~CudaOperationSolve() override { FreeScratch(); }   // a Solve operator dropped without Destroy() still returns its volumes
  // (Re-)initialisation with another container size or pitch releases the scratch volumes of the old one: they are allocated
  // again, at the new size, by the first Execute that wants them.
  bool Initialize(const OperationParameters* params = nullptr) override;
  // After Execute dev_phi / dev_ksi hold the weights of the LAST outer iteration, as in the reference
  // (cuda_operation_solve.cpp:215-221 writes them in place): the ping-pong with the operator's second pair starts on the side
  // that makes it end in the caller's buffers.
  void Execute(OperationParameters& params) override;
  void Destroy() override;

  bool silent = false;
  // container-sized device volumes the operator owns beyond what the caller lends it (0, 2, 4 or 6): the driver adds them to
  // its memory estimate
  static size_t ScratchVolumes();

 private:
  // second phi / ksi pair for the fused "last sweep + next phi/ksi" launch (f3d_solve_sweep_phi_ksi writes the weights of the
  // next outer iteration while other tiles still read the current ones); container sized, allocated on first use
  bool EnsureWeightScratch();
  DevicePtr phi_alt_ = 0, ksi_alt_ = 0;
  // frame derivatives of the level (f3d_frame_derivatives once per Execute, then the _fd launchers): four more container-sized
  // volumes owned by the operator
  bool EnsureDerivativeScratch();
  DevicePtr fder_[4] = {0, 0, 0, 0};
  // the container the scratch above was allocated for; anything else makes Ensure*Scratch start over
  DataSize4 scratch_size_ = {0, 0, 0, 0};
  bool ScratchFits() const;
  void FreeScratch();
};

#endif
