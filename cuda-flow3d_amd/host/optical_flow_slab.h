// Multi-GPU driver: the volume is cut into z-slabs, one per rank, and every rank runs the same coarse-to-fine loop on
// its slab with halo planes exchanged between neighbours.  New design (the reference is single-GPU);
// it reuses the reference-shaped pieces (level schedule, Gaussian taps, parameter keys) and the same kernels, so the
// result is bit-identical to OpticalFlowE on one GPU.
//
//   * Partition: rank r owns planes [floor(r D_l / P), floor((r+1) D_l / P)) of every level l.
//   * Halos are communication-avoiding: increments du, dv, dw are exchanged once per OUTER iteration, K + 1 planes
//     deep (K = inner sweeps); sweep j then runs on the slab widened by K-1-j planes, phi/ksi on the slab widened
//     by K.  The sweep is a Jacobi update, so the redundant planes reproduce the neighbour's values exactly.
//     Thin slabs of small (latency-bound) levels go further: n (K + 1) planes once per n outer iterations, nested windows.
//   * Per level: level-size halos of the two-pass-resampled frames and flows (z pass sources), K+1 planes of
//     u, v, w, f0; K+1+reach planes of f1 for the warp (reach = ceil(max|w| / hz) + 1, one all-reduce(max));
//     2 planes of u, v, w for the 5^3 median.
//   * Transport: RCCL send/recv between processes (one rank per GPU), or plane copies between the virtual ranks of
//     one process (a one-GPU rehearsal that exercises exactly the same plan).
#ifndef F3D_HOST_OPTICAL_FLOW_SLAB_H_
#define F3D_HOST_OPTICAL_FLOW_SLAB_H_

#include <vector>

#include "optical_flow.h"
#include "slab_plan.h"

class OpticalFlowSlab : public OpticalFlowBase {
 public:
  // n_ranks slabs in total; local_ranks are the ones this process computes: {rank} with RCCL, {0..n-1} for the
  // one-GPU rehearsal.  halo_capacity = planes kept free below and above the slab in every local container.
  OpticalFlowSlab(int n_ranks, std::vector<int> local_ranks, int halo_capacity = 16);
  ~OpticalFlowSlab() override;

  bool Initialize(const DataSize4& data_size) override;
  // frame_0 / frame_1 are the FULL host volumes; each local rank uploads its planes.  flow_* are full-size too and
  // receive the planes of the local ranks only.
  void ComputeFlow(Data3D& frame_0, Data3D& frame_1, Data3D& flow_u, Data3D& flow_v, Data3D& flow_w,
                   OperationParameters& params) override;
  void Destroy() override;

  // device-resident variant for benchmarks
  void UploadFrames(Data3D& frame_0, Data3D& frame_1);
  bool ComputeResident(OperationParameters& params);
  void DownloadFlow(Data3D& flow_u, Data3D& flow_v, Data3D& flow_w);
  float LastDeviceSeconds() const { return last_device_seconds_; }
  bool failed() const { return failed_; }
  // outer iterations of the last solve that ran in the overlapped order (exchange beside the interior)
  size_t OverlappedIterations() const { return overlapped_iterations_; }
  // groups of more than one outer iteration of the last solve that ran between two exchanges (thin slabs of small levels)
  size_t BatchedExchanges() const { return batched_exchanges_; }
  DataSize4 FullSize() const { return full_size_; }

  bool silent = true;

 private:
  // FDX .. FDT: the frame derivatives fx, fy, fz, ft of the level on the slab (optional: a rank without room for them runs the
  // fused launches on the frames, as rounds 1-3 did)
  enum Role { RAW0, RAW1, F0, F1, F0R, F1R, FU, FV, FW, DU, DV, DW, PHI, KSI, TDU, TDV, TDW, TMP, EDU, EDV, EDW, PHI2, KSI2,
              kRequiredRoles, FDX = kRequiredRoles, FDY, FDZ, FDT, kRoles };
  struct Local {
    int rank;
    DevicePtr buf[kRoles];
    // planes [weights_lo, weights_hi) of PHI / KSI already hold the weights of the coming outer iteration (written by the
    // fused last sweep of the previous one); empty = none
    int weights_lo = 0, weights_hi = 0;
    bool derivatives = false;   // FDX .. FDT hold the derivatives of the CURRENT level's frames on every plane a fused launch computes on
  };
  // the solver launches of one window: two sweeps (fused) or one; sweep + next weights with the edge planes kept.  The fused ones read
  // the frame derivatives when `l.derivatives` says they are there.
  bool Sweeps(Local& l, bool pair, const Role* in, const Role* out, size_t W, size_t H, int D, float hx, float hy, float hz,
              float equation_alpha, const f3d_slab& win);
  bool FrameDerivatives(Local& l, int D, size_t W, size_t H, float hx, float hy, float hz, int valid_halo);

  bool Pyramid(OperationParameters& params);
  int ZBase(int depth, int rank) const { return OwnedPlanes(depth, rank, n_ranks_).lo - halo_; }
  f3d_slab Window(int depth, int rank, int grow_lo, int grow_hi) const;
  // make `need` planes below/above every slab valid for the given roles (level of `depth`, sub-box width x height)
  bool Exchange(int depth, size_t width, size_t height, const std::vector<Role>& roles, int need_lo, int need_hi);
  // The same exchange in two halves for the one-rank-per-process case: Begin packs the planes the neighbours need out of
  // `send_roles` and starts the transfer beside the kernels issued afterwards; End waits for it and unpacks into `recv_roles`.
  bool ExchangeBegin(int depth, size_t width, size_t height, const std::vector<Role>& send_roles,
                     const std::vector<Role>& recv_roles, int need_lo, int need_hi);
  bool ExchangeEnd(size_t width, size_t height);
  // one outer iteration's sweeps with the halo exchange hidden behind the interior of the slab (see the .cpp)
  bool SweepsOverlapped(Local& l, int D, size_t W, size_t H, int K, float hx, float hy, float hz, float equation_alpha,
                        float equation_smoothness, float equation_data);
  // phi / ksi on planes [lo, hi) of the level (clipped to it), skipping the planes that already hold them
  bool CompleteWeights(Local& l, int lo, int hi, int D, size_t W, size_t H, float hx, float hy, float hz, float equation_smoothness,
                       float equation_data);
  // The last sweep of an outer iteration fused with the weights of the next one (f3d_solve_sweep_phi_ksi_edges): the sweep on
  // [sweep_lo, sweep_hi) from `in` into `out`, the weights on the planes of that range whose z neighbours' new increments
  // are at hand -- all but the first and the last one, unless that is a face of the volume.  `launched` stays false (and
  // the caller runs a plain sweep) when the range is too thin for that.
  bool SweepAndNextWeights(Local& l, const Role (&in)[3], const Role (&out)[3], int sweep_lo, int sweep_hi, int D, size_t W, size_t H,
                           float hx, float hy, float hz, float equation_alpha, float equation_smoothness, float equation_data,
                           bool& launched);
  // Warp whose z reach does not fit the halo room of the local containers ("fallback: gather f1_res"; the reference's
  // dead GPU warp asked for max_mag planes the same way, optical_flow_p.cpp:206, cuda_operation_register_p.cpp:165-179): frame 1 of
  // the level is gathered into a container of its own that holds the slab plus `need` planes on either side -- from as many ranks
  // as that takes (PlanHaloExchange is multi-hop) --, the warp runs on that container with the other operands' pointers rebased to
  // its first plane, and the result lands in TMP as usual.
  bool WarpWithGatheredFrame(int D, size_t W, size_t H, float hx, float hy, float hz, int wide, int need);
  bool GatherPlanes(int depth, size_t width, size_t height, Role src_role, int need);
  DevicePtr f1_wide_ = 0;          // per process: one buffer per local rank, each wide_planes_ deep
  size_t wide_planes_ = 0;
  size_t wide_warps_ = 0;          // levels of the last solve that took this road
 public:
  size_t GatheredWarps() const { return wide_warps_; }
  // exchanges of the last solve made after a solver STAGE (a fused pair or a single sweep) rather than once per outer iteration
  size_t StageExchanges() const { return stage_exchanges_; }
  // which of the two bit-identical exchange orders the next solve takes: false = K + 1 planes once per outer iteration with the
  // sweeps on widened windows (default), true = one message per solver stage (what F3D_SLAB_EXCHANGE=stage selects at construction);
  // bench.py times both in one invocation
  void SetExchangePerStage(bool per_stage) { exchange_per_stage_ = per_stage; }
  bool ExchangePerStage() const { return exchange_per_stage_; }
 private:
  // F3D_SLAB_EXCHANGE=stage: exchange the increments after every solver stage, as deep as the next stage reads (2 planes before a
  // fused pair, 1 before a single sweep, 3 before the next weights), instead of K + 1 = 6 planes once per outer iteration with the
  // sweeps on windows widened by up to K - 1 planes.  The same bytes travel in three messages instead of one; the redundant
  // planes drop from 20 sweep-planes + 10 weight-planes per outer iteration and rank to 4 + 4.  Which order wins depends on what an
  // exchange costs on the machine, hence a switch ("outer" = default) for the first run on a multi-GPU node to settle.
  bool exchange_per_stage_ = false;
  size_t stage_exchanges_ = 0;
  bool fused_weights_ = true;  // F3D_SLAB_FUSED_PHI_KSI=0 turns the fused last sweep off
  bool Check(int status);

  int n_ranks_;
  std::vector<int> local_ranks_;
  int halo_;
  DataSize4 full_size_ = {0, 0, 0, 0};
  DataSize4 local_container_ = {0, 0, 0, 0};
  std::vector<Local> locals_;
  DevicePtr stage_send_ = 0, stage_recv_ = 0;
  size_t stage_floats_ = 0;
  struct Unpack {  // what ExchangeEnd has to put where
    std::vector<f3d_devptr> field;
    std::vector<int> plane, count;
    std::vector<size_t> offset;
  } unpack_;
  int overlap_min_planes_ = 0;
  size_t overlapped_iterations_ = 0;
  size_t batched_exchanges_ = 0;
  int forced_outer_per_exchange_ = 0;
  int max_outer_per_exchange_ = 4;
  double small_level_voxels_ = 1.5e6;  // slab + halos below this many voxels: launches are latency-bound
  bool failed_ = false;
  float last_device_seconds_ = 0.f;
  CudaOperationConvolution3D taps_;  // only for ComputeGaussianKernel (host arithmetic)
};

#endif
