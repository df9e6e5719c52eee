// "Piecemeal" operators: the host-volume flavour of the operator layer (class names, name strings and parameter keys of
// src/cuda_operations/partial_data/cuda_operation_{add,resample,register,solve,stat}_p.{h,cpp}).  Every parameter that is
// a device pointer in operations.h is a Data3D* here: full-size dense host volumes whose corner sub-box holds the current
// pyramid level, streamed through the GPU in z-chunks sized to the free device memory, so a volume larger than HBM can be
// processed.
//
// What is new relative to the reference (which uploads ten fields and downloads three for EVERY sweep,
// cuda_operation_solve_p.cpp:188-207, 586-745, and warps on the CPU, cuda_operation_register_p.cpp:96-139):
//   * the solver keeps a chunk on the device for whole outer iterations: phi/ksi and the K inner sweeps run on windows
//     that shrink by one plane per sweep inside a halo of n (K + 1) planes, n outer iterations per residency, so the
//     PCIe traffic per outer iteration is 11 / n field transfers instead of 13 K + 10; phi and ksi never leave the device;
//   * the warp runs on the device with a per-chunk halo taken from max |w| of the chunk;
//   * chunks live in one reusable device arena, in a compact container of the level's own size.
// The kernels are the resident path's, launched on slab windows, so every result is bit-identical to the corresponding
// "entire data" operator.
#ifndef F3D_HOST_OPERATIONS_P_H_
#define F3D_HOST_OPERATIONS_P_H_

#include "operations.h"

// Device memory the piecemeal operators may use for chunks: F3D_P_BUDGET_MB when set (tests force small chunks with it),
// else 85 % of the free device memory plus what the shared chunk arena already holds.
size_t PiecemealBudgetBytes();
// Give the arena's device memory back (the driver's Destroy does).
void PiecemealReleaseArena();
// Device memory the driver holds outside the arena (resident coarse levels); counted against a forced budget.
void PiecemealSetReservedBytes(size_t bytes);
// Smallest arena the frame resample needs for a W0 x H0 original (one output plane per chunk).
size_t PiecemealMinResampleBytes(size_t width, size_t height);

// How the solver cuts a level: `chunk` owned planes per residency, `outer_per_pass` outer iterations computed before the
// increments go back to the host, `halo` = outer_per_pass * (K + 1) planes uploaded on either side (0 when the level fits).
struct SolvePiecemealPlan {
  int chunk = 0;
  int outer_per_pass = 0;
  int halo = 0;
  int max_planes = 0;       // planes of one field the budget allows (per chunk set when overlapped)
  bool overlapped = false;  // two chunk sets: upload of the next and download of the previous chunk beside the kernels
  double cost = 0.0;        // the model's seconds per owned voxel
  // the fields that do not change during a level's solve (the two frames, u, v, w) hold the WHOLE level on the device beside the chunk
  // sets and go up once instead of once per residency (three fields up per residency instead of eight)
  bool constants_on_device = false;
};
// Pure host arithmetic (CPU-testable).  forced_outer_per_pass > 0 pins n (F3D_P_OUTER_PER_PASS); overlap_mode 0 / 1 pins
// the serial / overlapped schedule, anything else lets the cost model choose (F3D_P_OVERLAP); chunk == 0 means the budget
// cannot hold even one plane with its halo.
// `constant_fields` > 0 prices the layout with that many whole-level fields held on the device (`fields` then counts what a chunk set
// still holds: 8, or 10 with the second weight pair): their bytes come off the budget, their upload is paid once.
SolvePiecemealPlan PlanSolvePiecemeal(size_t budget_bytes, size_t width, size_t height, int depth, int inner_iterations,
                                      int outer_iterations, int forced_outer_per_pass, int overlap_mode = -1, int fields = 13,
                                      int constant_fields = 0);

class CudaOperationPiecemealBase : public CudaOperationBase {
 public:
  // the reference's piecemeal operators take no container at Initialize (cuda_operation_solve_p.cpp:34-58)
  bool Initialize(const OperationParameters* params = nullptr) override;
  // the shared chunk arena goes back to the device when the last initialised piecemeal operator is destroyed
  void Destroy() override;
  ~CudaOperationPiecemealBase() override
  {
    if (initialized_) Destroy();
  }

 protected:
  explicit CudaOperationPiecemealBase(const char* name) : CudaOperationBase(name) {}
};

// operand_0 += operand_1                      keys: operand_0, operand_1 (Data3D*), data_size
class CudaOperationAddP : public CudaOperationPiecemealBase {
 public:
  CudaOperationAddP() : CudaOperationPiecemealBase("CUDA Add Piecemeal") {}
  void Execute(OperationParameters& params) override;
};

// area resample of the data_size sub-box of `input` into the resample_size sub-box of `output`; input == output is allowed
// (the driver resamples the flow in place)    keys: input, output (Data3D*), data_size, resample_size
class CudaOperationResampleP : public CudaOperationPiecemealBase {
 public:
  CudaOperationResampleP() : CudaOperationPiecemealBase("CUDA Resample Piecemeal") {}
  void Execute(OperationParameters& params) override;
  // Driver-internal variant: the same chunked passes, but the resampled sub-box lands in a device container (planes of
  // dst_rows rows of dst_pitch bytes) instead of a host volume.
  bool ExecuteToDevice(Data3D& input, const DataSize4& data_size, const DataSize4& resample_size, DevicePtr dst, size_t dst_pitch,
                       size_t dst_rows);
  // ... and with the source on the device as well (a container of src_rows rows of src_pitch bytes per plane): the chunked
  // passes keep the working set small, nothing crosses the link.
  bool ExecuteDeviceToDevice(DevicePtr src, size_t src_pitch, size_t src_rows, const DataSize4& data_size, const DataSize4& resample_size,
                             DevicePtr dst, size_t dst_pitch, size_t dst_rows);

 private:
  bool Run(Data3D* input, DevicePtr src, size_t src_pitch, size_t src_rows, const DataSize4& data_size, const DataSize4& resample_size,
           Data3D* output, DevicePtr dst, size_t dst_pitch, size_t dst_rows);
};

// backward trilinear warp of frame_1; the result is written to `temp` and the two volumes are swapped, like the reference
// keys: frame_0, frame_1, flow_u, flow_v, flow_w, temp (Data3D*), hx, hy, hz, data_size, max_mag (size_t, a hint only)
class CudaOperationRegistrationP : public CudaOperationPiecemealBase {
 public:
  CudaOperationRegistrationP() : CudaOperationPiecemealBase("CUDA Registration") {}
  void Execute(OperationParameters& params) override;
};

// lagged-nonlinearity solver on host volumes.  keys: frame_0, frame_1, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw,
// temp_du, temp_dv, temp_dw (Data3D*), outer_iterations_count, inner_iterations_count, equation_alpha,
// equation_smoothness, equation_data, hx, hy, hz, data_size.  "phi" and "ksi" are accepted and ignored: the
// nonlinearities stay on the device.  The name string keeps the reference's spelling.  The overlapped schedule (copies beside
// the kernels) is only considered when all eleven volumes are page-locked (f3d_host_register or pinned by the caller).
class CudaOperationSolveP : public CudaOperationPiecemealBase {
 public:
  CudaOperationSolveP() : CudaOperationPiecemealBase("CUDA Sove Piecemeal") {}
  void Execute(OperationParameters& params) override;

  bool silent = false;
  // The driver's next step after the solve is "flow += increment" (cuda_operation_add_p: two uploads and a download per component and
  // chunk).  With this set, the LAST residency of a chunk -- which holds u, v, w and the final du, dv, dw on the device anyway -- adds
  // them there and hands back the SUMS in the increments' volumes (IEEE addition commutes: du + u is the u + du the add operator
  // forms, bit for bit); LastAddedToFlow() says whether it did.  The flow volumes themselves are only read.
  bool add_increments_to_flow = false;
  bool LastAddedToFlow() const { return last_added_; }
  // The driver's step BEFORE the solve is the registration of frame 1 (cuda_operation_register_p: frame 0, u, v, w and frame 1 go up
  // per chunk, the registered frame comes down -- and the solver's first residency sends frame 0, u, v, w and the registered frame up
  // again).  With this set, "frame_1" names the UNREGISTERED frame: the first residency of a chunk brings its planes (widened by the
  // reach of the flow's w) into the three buffers the first sweep has not written yet, warps on the device into the chunk's frame-1
  // buffer and sends the registered planes the chunk owns down beside the increments; later residencies read that registered frame.
  // Five field uploads per level less, same bits (f3d_warp on the same operands).  Where the registered frame goes on the host:
  // `registered_frame_1` if given ("frame_1" is then only read); otherwise the "frame_1" volume itself holds the registered frame
  // afterwards, as after the registration operator -- during the first pass it collects in the storage of "flow_du", which idles then,
  // and the unregistered frame's storage becomes the download target that storage would have been (Data3D::Swap, no copy).
  // LastRegistered() says whether it happened: a flow whose reach does not leave room in those buffers (or a call that ran no sweep)
  // returns with nothing done and the caller registers the classical way.
  bool register_frame_1 = false;
  Data3D* registered_frame_1 = nullptr;
  bool LastRegistered() const { return last_registered_; }
  // what the last Execute did (tests and the driver's log)
  const SolvePiecemealPlan& LastPlan() const { return last_plan_; }
  size_t LastPasses() const { return last_passes_; }
  // whether the last Execute ran the last sweep of an outer iteration and the next weights as one launch (15 fields per chunk set)
  bool LastFusedWeights() const { return last_fused_weights_; }

 private:
  SolvePiecemealPlan last_plan_;
  size_t last_passes_ = 0;
  bool last_fused_weights_ = false;
  bool last_added_ = false;
  bool last_registered_ = false;
};

// The two filters the reference's piecemeal driver leaves out (its median is commented out, optical_flow_p.cpp:268-302, and
// it never blurs), as host-volume operators in the same style, so that OpticalFlowP can reproduce OpticalFlowE's full
// pipeline on volumes that do not fit the device (OpticalFlowP::full_pipeline).  No reference counterpart: names and keys
// follow the "entire data" operators with Data3D* values.

// separable Gaussian, rows -> columns -> slices     keys: input, output (Data3D*, different volumes), data_size, gaussian_sigma
class CudaOperationConvolution3DP : public CudaOperationPiecemealBase {
 public:
  CudaOperationConvolution3DP() : CudaOperationPiecemealBase("CUDA Convolution 3D Piecemeal") {}
  void Execute(OperationParameters& params) override;

 private:
  CudaOperationConvolution3D taps_;  // host arithmetic of the tap generator only
};

// 3-D median, "radius" is the window diameter (1 = copy, even = one less, 3 / 5 / 7).  input == output is allowed: the planes
// a later chunk still needs of what an earlier chunk overwrote stay on the device.   keys: input, output, data_size, radius
class CudaOperationMedianP : public CudaOperationPiecemealBase {
 public:
  CudaOperationMedianP() : CudaOperationPiecemealBase("CUDA Median Piecemeal") {}
  void Execute(OperationParameters& params) override;
};

// min / max / average flow magnitude            keys: flow_u, flow_v, flow_w (Data3D*), data_size, stat (Stat3*)
class CudaOperationStatP : public CudaOperationPiecemealBase {
 public:
  CudaOperationStatP() : CudaOperationPiecemealBase("CUDA Stat Piecemeal") {}
  void Execute(OperationParameters& params) override;
  bool silent = false;
};

#endif
