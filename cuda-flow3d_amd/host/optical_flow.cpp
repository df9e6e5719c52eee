// Coarse-to-fine driver.  Sequence of operations, buffer counts, parameter keys and console lines follow
// OpticalFlowE::ComputeFlow (src/optical_flow/optical_flow_e.cpp:132-601) and OpticalFlowBase
// (src/optical_flow/optical_flow_base.cpp); the code is organised around a container pool and one
// RunPyramid() shared by the host-volume entry point and the device-resident one.
#include "optical_flow.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

#include "common_utils.h"
#include "operator_calls.h"
#include "hip_utils.h"

// ---- base --------------------------------------------------------------------------------------------------

size_t OpticalFlowBase::GetMaxWarpLevel(size_t width, size_t height, size_t depth, float scale_factor)
{
  // count levels while every axis of ceil(dim * sf^level) keeps at least 4 voxels (optical_flow_base.cpp:31-56)
  size_t rw = 1, rh = 1, rd = 1;
  size_t level_counter = 1;
  while (scale_factor < 1.f) {
    const float scale = std::pow(scale_factor, static_cast<float>(level_counter));
    rw = static_cast<size_t>(std::ceil(width * scale));
    rh = static_cast<size_t>(std::ceil(height * scale));
    rd = static_cast<size_t>(std::ceil(depth * scale));
    if (rw < 4 || rh < 4 || rd < 4) break;
    ++level_counter;
  }
  if (rw == 1 || rh == 1 || rd == 1) --level_counter;
  return level_counter;
}

PyramidLevel OpticalFlowBase::GetLevel(const DataSize4& original, float scale_factor, int level)
{
  // optical_flow_e.cpp:262-268: float product, std::ceil, spacing = original / current
  PyramidLevel out;
  const float scale = std::pow(scale_factor, static_cast<float>(level));
  out.size.width = static_cast<size_t>(std::ceil(original.width * scale));
  out.size.height = static_cast<size_t>(std::ceil(original.height * scale));
  out.size.depth = static_cast<size_t>(std::ceil(original.depth * scale));
  out.size.pitch = 0;
  out.hx = original.width / static_cast<float>(out.size.width);
  out.hy = original.height / static_cast<float>(out.size.height);
  out.hz = original.depth / static_cast<float>(out.size.depth);
  return out;
}

bool OpticalFlowBase::IsInitialized() const
{
  if (!initialized_) std::printf("Error: '%s' was not initialized.\n", name_);
  return initialized_;
}

void OpticalFlowBase::ComputeFlow(Data3D&, Data3D&, Data3D&, Data3D&, Data3D&, OperationParameters&)
{
  std::printf("Warning: '%s' ComputeFlow() was not defined.\n", name_);
}

void OpticalFlowBase::Destroy() { initialized_ = false; }

OpticalFlowBase::~OpticalFlowBase() {}

// ---- single-GPU driver ---------------------------------------------------------------------------------------

OpticalFlowE::OpticalFlowE() : OpticalFlowBase("Optical Flow Single GPU")
{
  // same initialisation order as the reference's forward_list built with push_front (optical_flow_e.cpp:34-39)
  cuda_operations_ = {&cuop_solve_, &cuop_resample_, &cuop_register_, &cuop_median_, &cuop_convolution_, &cuop_add_};
}

OpticalFlowE::~OpticalFlowE() { Destroy(); }

bool OpticalFlowE::Initialize(const DataSize4& data_size)
{
  dev_container_size_ = data_size;
  dev_container_size_.pitch = 0;
  initialized_ = InitCudaMemory() && InitCudaOperations();
  return initialized_;
}

bool OpticalFlowE::InitCudaMemory()
{
  std::printf("Allocating memory on the device...\n");
  size_t free_memory = 0, total_memory = 0;
  CheckDeviceError(f3d_mem_info(&free_memory, &total_memory));
  const float mb = 1024.f * 1024.f;
  std::printf("Available\t:\t%.0fMB / %.0fMB\n", free_memory / mb, total_memory / mb);

  const size_t rows = dev_container_size_.height * dev_container_size_.depth;
  const size_t row_bytes = dev_container_size_.width * sizeof(float);
  const size_t pitch_guess = (row_bytes + 255) / 256 * 256;
  const size_t needed_memory = pitch_guess * rows * kContainers;
  // the solve operator allocates up to six more container-sized volumes on first use (second weight pair of the fused last
  // sweep, frame derivatives); it runs the unfused schedule when they do not fit, so they are reported, not required
  const size_t optional_memory = pitch_guess * rows * CudaOperationSolve::ScratchVolumes();
  std::printf("Needed (approx.):\t%.0fMB (+ %.0fMB optional solver scratch)\n", needed_memory / mb, optional_memory / mb);
  if (needed_memory >= free_memory) return false;
  // the fit decision counts the 15 containers only; what the optional scratch will meet is said here, so that a run that ends up on
  // the schedules without it (frame builds, separate phi/ksi launches) is not a surprise (advisor, round 3)
  if (optional_memory > 0 && needed_memory + optional_memory >= free_memory)
    std::printf("Solver scratch	:	does not fit beside the containers: the solver will take the launches that need none\n");

  size_t allocated_memory = 0;
  for (size_t i = 0; i < kContainers; ++i) {
    DevicePtr container = 0;
    size_t pitch = 0;
    const bool error = CheckDeviceError(f3d_alloc_pitched(&container, &pitch, row_bytes, rows));
    if (!error) free_containers_.push_back(container);
    // every container must come back with the same pitch
    if (error || (i != 0 && pitch != dev_container_size_.pitch)) {
      std::printf("Error during device memory allocation.");
      Destroy();
      return false;
    }
    dev_container_size_.pitch = pitch;
    allocated_memory += pitch * rows;
  }
  std::printf("Allocated\t:\t%.0fMB\n", allocated_memory / mb);
  return true;
}

bool OpticalFlowE::InitCudaOperations()
{
  if (dev_container_size_.pitch == 0) {
    std::printf("Initialization failed. Device pitch is 0.\n");
    return false;
  }
  std::printf("Initialization of cuda operations...\n");
  OperationParameters op;
  op.PushValuePtr("container_size", &dev_container_size_);
  for (CudaOperationBase* cuop : cuda_operations_) {
    std::printf("%-18s: ", cuop->GetName());
    if (!cuop->Initialize(&op)) {
      Destroy();
      return false;
    }
    std::printf("OK\n");
  }
  return true;
}

DevicePtr OpticalFlowE::Borrow()
{
  // ComputeFlow holds at most 13 of the 15 containers at a time (ten roles + three temps of the batched flow resampling); a caller
  // that keeps more -- results not yet released, a driver extended in place -- gets a fresh container instead of an empty stack
  if (free_containers_.empty()) {
    DevicePtr extra = 0;
    size_t pitch = 0;
    const size_t rows = dev_container_size_.height * dev_container_size_.depth;
    if (CheckDeviceError(f3d_alloc_pitched(&extra, &pitch, dev_container_size_.width * sizeof(float), rows)) ||
        pitch != dev_container_size_.pitch) {
      std::printf("'%s': the container pool is empty and another container could not be allocated.\n", GetName());
      if (extra) f3d_free(extra);
      return 0;
    }
    return extra;   // joins the pool when it is given back
  }
  DevicePtr p = free_containers_.back();
  free_containers_.pop_back();
  return p;
}

void OpticalFlowE::GiveBack(DevicePtr p) { free_containers_.push_back(p); }

void OpticalFlowE::ReleaseResult()
{
  for (DevicePtr& p : result_flow_) {
    if (p) GiveBack(p);
    p = 0;
  }
}

bool OpticalFlowE::AllocateResidentFrames()
{
  if (!IsInitialized()) return false;
  if (resident_frame_[0]) return true;
  const size_t rows = dev_container_size_.height * dev_container_size_.depth;
  for (int i = 0; i < 2; ++i) {
    size_t pitch = 0;
    if (CheckDeviceError(f3d_alloc_pitched(&resident_frame_[i], &pitch, dev_container_size_.width * sizeof(float), rows)) ||
        pitch != dev_container_size_.pitch)
      return false;
  }
  return true;
}

void OpticalFlowE::UploadResidentFrames(Data3D& frame_0, Data3D& frame_1)
{
  if (!AllocateResidentFrames()) return;
  CopyData3DtoDevice(frame_0, resident_frame_[0], dev_container_size_.height, dev_container_size_.pitch);
  CopyData3DtoDevice(frame_1, resident_frame_[1], dev_container_size_.height, dev_container_size_.pitch);
}

void OpticalFlowE::ComputeFlow(Data3D& frame_0, Data3D& frame_1, Data3D& flow_u, Data3D& flow_v, Data3D& flow_w,
                               OperationParameters& params)
{
  if (!IsInitialized()) return;
  if (frame_0.Width() != dev_container_size_.width || frame_0.Height() != dev_container_size_.height ||
      frame_0.Depth() != dev_container_size_.depth || frame_1.Width() != frame_0.Width() ||
      frame_1.Height() != frame_0.Height() || frame_1.Depth() != frame_0.Depth()) {
    std::printf("Error: '%s'. Frame dimensions differ from the initialised container.\n", GetName());
    return;
  }
  ReleaseResult();

  // the reference's timer spans H2D .. D2H (optical_flow_e.cpp:169,579)
  f3d_event ev_start = nullptr, ev_stop = nullptr;
  CheckDeviceError(f3d_event_create(&ev_start));
  CheckDeviceError(f3d_event_create(&ev_stop));
  CheckDeviceError(f3d_event_record(ev_start));

  DevicePtr raw_0 = Borrow(), raw_1 = Borrow();
  CopyData3DtoDevice(frame_0, raw_0, dev_container_size_.height, dev_container_size_.pitch);
  CopyData3DtoDevice(frame_1, raw_1, dev_container_size_.height, dev_container_size_.pitch);

  if (RunPyramid(params, raw_0, raw_1, true)) {
    DownloadFlow(flow_u, flow_v, flow_w);
    float elapsed_ms = 0.f;
    CheckDeviceError(f3d_event_record(ev_stop));
    CheckDeviceError(f3d_event_sync(ev_stop));
    CheckDeviceError(f3d_event_elapsed_ms(&elapsed_ms, ev_start, ev_stop));
    std::printf("Total GPU computation time: % 4.4fs\n", elapsed_ms / 1000.);
  } else {
    GiveBack(raw_0);
    GiveBack(raw_1);
  }
  ReleaseResult();
  f3d_event_destroy(ev_start);
  f3d_event_destroy(ev_stop);
}

void OpticalFlowE::ComputeFlowResident(OperationParameters& params)
{
  if (!IsInitialized()) return;
  if (!resident_frame_[0]) {
    std::printf("Error: '%s'. Resident frames were not allocated.\n", GetName());
    return;
  }
  ReleaseResult();
  f3d_event ev_start = nullptr, ev_stop = nullptr;
  CheckDeviceError(f3d_event_create(&ev_start));
  CheckDeviceError(f3d_event_create(&ev_stop));
  CheckDeviceError(f3d_event_record(ev_start));
  if (RunPyramid(params, resident_frame_[0], resident_frame_[1], false)) {
    float elapsed_ms = 0.f;
    CheckDeviceError(f3d_event_record(ev_stop));
    CheckDeviceError(f3d_event_sync(ev_stop));
    CheckDeviceError(f3d_event_elapsed_ms(&elapsed_ms, ev_start, ev_stop));
    last_device_seconds_ = elapsed_ms / 1000.f;
  }
  f3d_event_destroy(ev_start);
  f3d_event_destroy(ev_stop);
}

bool OpticalFlowE::AllocateSequenceFrames()
{
  if (!AllocateResidentFrames()) return false;
  if (sequence_frame_[2]) return true;
  sequence_frame_[0] = resident_frame_[0];
  sequence_frame_[1] = resident_frame_[1];
  size_t pitch = 0;
  const size_t rows = dev_container_size_.height * dev_container_size_.depth;
  if (CheckDeviceError(f3d_alloc_pitched(&sequence_frame_[2], &pitch, dev_container_size_.width * sizeof(float), rows)) ||
      pitch != dev_container_size_.pitch) {
    sequence_frame_[2] = 0;
    return false;
  }
  return true;
}

void OpticalFlowE::SelectResidentPair(int slot_0, int slot_1)
{
  if (!sequence_frame_[2]) return;
  resident_frame_[0] = sequence_frame_[slot_0 % 3];
  resident_frame_[1] = sequence_frame_[slot_1 % 3];
}

void OpticalFlowE::BeginComputeFlowResident(OperationParameters& params)
{
  if (!IsInitialized() || !resident_frame_[0] || ev_begin_) return;
  ReleaseResult();
  CheckDeviceError(f3d_event_create(&ev_begin_));
  CheckDeviceError(f3d_event_create(&ev_end_));
  CheckDeviceError(f3d_event_record(ev_begin_));
  RunPyramid(params, resident_frame_[0], resident_frame_[1], false);  // enqueues; nothing in it waits for the device when silent
  CheckDeviceError(f3d_event_record(ev_end_));
}

void OpticalFlowE::EndComputeFlowResident()
{
  if (!ev_begin_) return;
  float elapsed_ms = 0.f;
  CheckDeviceError(f3d_event_sync(ev_end_));
  CheckDeviceError(f3d_event_elapsed_ms(&elapsed_ms, ev_begin_, ev_end_));
  last_device_seconds_ = elapsed_ms / 1000.f;
  f3d_event_destroy(ev_begin_);
  f3d_event_destroy(ev_end_);
  ev_begin_ = ev_end_ = nullptr;
}

bool OpticalFlowE::TakeResult(DevicePtr (&flow)[3])
{
  if (!result_flow_[0]) return false;
  // the pool keeps its fifteen: three spare containers replace the ones that leave (allocated once, recycled afterwards)
  const size_t rows = dev_container_size_.height * dev_container_size_.depth;
  while (free_containers_.size() < kContainers) {
    DevicePtr p = 0;
    size_t pitch = 0;
    if (CheckDeviceError(f3d_alloc_pitched(&p, &pitch, dev_container_size_.width * sizeof(float), rows)) || pitch != dev_container_size_.pitch)
      return false;
    free_containers_.push_back(p);
    ++extra_containers_;
  }
  for (int i = 0; i < 3; ++i) {
    flow[i] = result_flow_[i];
    result_flow_[i] = 0;
  }
  return true;
}

void OpticalFlowE::GiveResultBack(DevicePtr (&flow)[3])
{
  for (DevicePtr& p : flow) {
    if (p) free_containers_.push_back(p);
    p = 0;
  }
}

void OpticalFlowE::DownloadFlow(Data3D& flow_u, Data3D& flow_v, Data3D& flow_w)
{
  if (!result_flow_[0]) return;
  CopyData3DFromDevice(result_flow_[0], flow_u, dev_container_size_.height, dev_container_size_.pitch);
  CopyData3DFromDevice(result_flow_[1], flow_v, dev_container_size_.height, dev_container_size_.pitch);
  CopyData3DFromDevice(result_flow_[2], flow_w, dev_container_size_.height, dev_container_size_.pitch);
}

bool OpticalFlowE::ResultStatistics(Stat3& stat)
{
  if (!IsInitialized() || !result_flow_[0]) return false;
  OperationParameters init;
  init.PushValuePtr("container_size", &dev_container_size_);
  if (!cuop_stat_.Initialize(&init)) return false;
  DataSize4 data_size = {dev_container_size_.width, dev_container_size_.height, dev_container_size_.depth, 0};
  OperationParameters bag;
  calls::Statistics(bag, {&result_flow_[0], &result_flow_[1], &result_flow_[2]}, &data_size, &stat);
  cuop_stat_.silent = true;
  cuop_stat_.Execute(bag);
  return true;
}

bool OpticalFlowE::ResidualOf(DevicePtr frame_0, DevicePtr warped, const DataSize4& size, Residual& out)
{
  double ssq = 0.0, sab = 0.0;
  float mx = 0.f;
  if (CheckDeviceError(f3d_residual_stats(frame_0, warped, size.width, size.height, size.depth, nullptr, &ssq, &sab, &mx))) return false;
  const double n = static_cast<double>(size.width) * static_cast<double>(size.height) * static_cast<double>(size.depth);
  out.rms = std::sqrt(ssq / n);
  out.mean_abs = sab / n;
  out.max_abs = mx;
  return true;
}

bool OpticalFlowE::FinalResidual(Residual& registered, Residual& unregistered)
{
  if (!IsInitialized() || !result_flow_[0] || !resident_frame_[0]) return false;
  DataSize4 size = {dev_container_size_.width, dev_container_size_.height, dev_container_size_.depth, 0};
  DevicePtr dev_temp = Borrow();
  float h = 1.f;
  OperationParameters op;
  calls::Registration(op, &resident_frame_[0], &resident_frame_[1], {&result_flow_[0], &result_flow_[1], &result_flow_[2]}, &dev_temp, &size,
                      {&h, &h, &h});
  cuop_register_.Execute(op);
  const bool ok = ResidualOf(resident_frame_[0], dev_temp, size, registered) &&
                  ResidualOf(resident_frame_[0], resident_frame_[1], size, unregistered);
  GiveBack(dev_temp);
  return ok;
}

// The coarse-to-fine solve on two frames that are already on the device (optical_flow_e.cpp:208-533 is the sequence of operator
// calls this reproduces: pre-blur; per level frames from the originals, flow from the level before, registration, solve, update,
// median).  Containers are named by what they hold:
//   blurred[2]   the two full-size frames the pyramid reads (pre-blurred copies, or the raw frames when sigma <= 0)
//   level[2]     the two frames at the current level; level[1] is replaced by its registered version
//   flow[3]      u, v, w so far        step[3]   the level's increments du, dv, dw (and, between levels, ping-pong room)
bool OpticalFlowE::RunPyramid(OperationParameters& params, DevicePtr raw_0, DevicePtr raw_1, bool raw_is_pooled)
{
  size_t warp_levels_count, outer_iterations_count, inner_iterations_count, median_radius;
  float warp_scale_factor, equation_alpha, equation_smoothness, equation_data, gaussian_sigma;
  GET_PARAM_OR_RETURN_VALUE(params, size_t, warp_levels_count, "warp_levels_count", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, warp_scale_factor, "warp_scale_factor", false);
  GET_PARAM_OR_RETURN_VALUE(params, size_t, outer_iterations_count, "outer_iterations_count", false);
  GET_PARAM_OR_RETURN_VALUE(params, size_t, inner_iterations_count, "inner_iterations_count", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, equation_alpha, "equation_alpha", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, equation_smoothness, "equation_smoothness", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, equation_data, "equation_data", false);
  GET_PARAM_OR_RETURN_VALUE(params, size_t, median_radius, "median_radius", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, gaussian_sigma, "gaussian_sigma", false);
  calls::SolverSettings settings = {&outer_iterations_count, &inner_iterations_count, &equation_alpha, &equation_smoothness, &equation_data};

  if (!silent) std::printf("\nStarting optical flow computation...\n");

  DataSize4 whole = {dev_container_size_.width, dev_container_size_.height, dev_container_size_.depth, 0};
  const size_t deepest = GetMaxWarpLevel(whole.width, whole.height, whole.depth, warp_scale_factor);
  const int first_level = static_cast<int>(std::min(warp_levels_count, deepest)) - 1;
  const size_t container_rows = dev_container_size_.height * dev_container_size_.depth;
  const size_t container_row_bytes = dev_container_size_.width * sizeof(float);
  auto zero_container = [&](DevicePtr p) { CheckDeviceError(f3d_memset2d(p, dev_container_size_.pitch, 0, container_row_bytes, container_rows)); };
  OperationParameters bag;

  // ---- the frames the pyramid reads ------------------------------------------------------------------------------------------
  DevicePtr raw[2] = {raw_0, raw_1};
  DevicePtr blurred[2];
  if (gaussian_sigma > 0.0) {
    DevicePtr scratch = Borrow();
    for (int f = 0; f < 2; ++f) {
      blurred[f] = Borrow();
      cuop_convolution_.Execute(calls::Convolution(bag, &raw[f], &blurred[f], &scratch, &whole, &gaussian_sigma));
    }
    GiveBack(scratch);
    if (raw_is_pooled)
      for (DevicePtr p : raw) GiveBack(p);
  } else if (raw_is_pooled) {
    blurred[0] = raw[0];   // nothing to blur: the uploaded containers themselves
    blurred[1] = raw[1];
  } else {
    // resident frames must survive the solve (level 0 takes the containers of `blurred` over): work on copies
    for (int f = 0; f < 2; ++f) {
      blurred[f] = Borrow();
      CheckDeviceError(f3d_copy_d2d(blurred[f], raw[f], dev_container_size_.pitch * container_rows));
    }
  }

  DevicePtr level_frame[2] = {Borrow(), Borrow()};
  DevicePtr flow[3] = {Borrow(), Borrow(), Borrow()};
  DevicePtr step[3] = {Borrow(), Borrow(), Borrow()};
  const calls::Flow flow_roles = {&flow[0], &flow[1], &flow[2]}, step_roles = {&step[0], &step[1], &step[2]};

  level_stats_.clear();
  if (first_level < 0)  // no level requested: the flow is identically zero
    for (DevicePtr p : flow) zero_container(p);

  // `count` (<= 3) volumes of one box through the three resampling passes together: three launches, a scratch container each
  auto resample_together = [&](DevicePtr* in, DevicePtr* out, size_t count, DataSize4& from, DataSize4& to) {
    OperationParameters bags[3];
    DevicePtr scratch[3] = {0, 0, 0};
    for (size_t i = 0; i < count; ++i) {
      scratch[i] = Borrow();
      calls::Resample(bags[i], &in[i], &out[i], &scratch[i], &from, &to);
    }
    cuop_resample_.ExecuteBatch(bags, count);
    for (size_t i = 0; i < count; ++i) GiveBack(scratch[i]);
  };

  DataSize4 coarser = {0, 0, 0, 0};   // the box of the level before (none yet)
  for (int level = first_level; level >= 0; --level) {
    PyramidLevel geometry = GetLevel(whole, warp_scale_factor, level);
    DataSize4 box = geometry.size;
    const calls::Spacing spacing = {&geometry.hx, &geometry.hy, &geometry.hz};
    char range_name[64];
    std::snprintf(range_name, sizeof(range_name), "level %d (%zu x %zu x %zu)", level, box.width, box.height, box.depth);
    ProfilerRange level_range(range_name);
    if (!silent) std::printf("Solve level %2d (%4zu x%4zu x%4zu) \n", level, box.width, box.height, box.depth);

    // 1. the two frames at this size: always from the ORIGINAL-size frames; at the finest level those are the level's frames
    if (level == 0)
      for (int f = 0; f < 2; ++f) std::swap(blurred[f], level_frame[f]);
    else
      resample_together(blurred, level_frame, 2, whole, box);

    // 2. the flow so far at this size (zero before the first level): resampled into the increments' containers, roles swapped;
    //    the values are not rescaled -- the flow is kept in original-voxel units
    if (coarser.width == 0) {
      for (DevicePtr p : flow) zero_container(p);
    } else {
      resample_together(flow, step, 3, coarser, box);
      for (int c = 0; c < 3; ++c) std::swap(flow[c], step[c]);
    }

    // 3. frame 1 registered with that flow
    {
      DevicePtr registered = Borrow();
      cuop_register_.Execute(calls::Registration(bag, &level_frame[0], &level_frame[1], flow_roles, &registered, &box, spacing));
      std::swap(level_frame[1], registered);
      GiveBack(registered);
    }
    if (collect_level_statistics) {
      LevelStatistics st;
      st.level = level;
      st.size = box;
      ResidualOf(level_frame[0], level_frame[1], box, st.before);
      level_stats_.push_back(st);
    }

    // 4. the increments of this level: weights + sweeps; five containers on loan for the duration
    {
      DevicePtr phi = Borrow(), ksi = Borrow();
      DevicePtr partner[3] = {Borrow(), Borrow(), Borrow()};
      const calls::Flow partner_roles = {&partner[0], &partner[1], &partner[2]};
      cuop_solve_.silent = silent;
      cuop_solve_.Execute(calls::Solve(bag, &level_frame[0], &level_frame[1], flow_roles, step_roles, partner_roles, &phi, &ksi, settings,
                                       &box, spacing));
      for (DevicePtr p : {phi, ksi, partner[0], partner[1], partner[2]}) GiveBack(p);
    }

    // 5. flow += increments, then the median of every component.  The components are independent through both, so the three go out
    //    in one launch each; the increments are consumed by "+=", so their containers receive the filtered flow and the roles swap
    //    (the reference filters through one scratch container, a component at a time)
    {
      OperationParameters bags[3];
      for (int c = 0; c < 3; ++c) calls::Add(bags[c], &flow[c], &step[c], &box);
      cuop_add_.ExecuteBatch(bags, 3);
      for (int c = 0; c < 3; ++c) calls::Median(bags[c], &flow[c], &step[c], &box, &median_radius);
      cuop_median_.ExecuteBatch(bags, 3);
      for (int c = 0; c < 3; ++c) std::swap(flow[c], step[c]);
    }

    if (collect_level_statistics) {
      float mn = 0.f, mx = 0.f;
      double sum = 0.0;
      if (!CheckDeviceError(f3d_flow_stats(flow[0], flow[1], flow[2], box.width, box.height, box.depth, nullptr, &mn, &mx, &sum))) {
        const double n = static_cast<double>(box.width) * static_cast<double>(box.height) * static_cast<double>(box.depth);
        level_stats_.back().flow = {mn, mx, static_cast<float>(sum / n)};
      }
      if (!silent) {
        const LevelStatistics& st = level_stats_.back();
        std::printf("  residual before the solve: rms %.5f  mean |.| %.5f  max %.4f;  flow after it: min %.4f  max %.4f  avg %.4f\n",
                    st.before.rms, st.before.mean_abs, st.before.max_abs, st.flow.min, st.flow.max, st.flow.avg);
      }
    }
    coarser = box;
  }

  // (u, v, w) stay out until they are downloaded or taken; everything else returns to the pool
  for (int c = 0; c < 3; ++c) result_flow_[c] = flow[c];
  for (DevicePtr p : {blurred[0], blurred[1], level_frame[0], level_frame[1], step[0], step[1], step[2]}) GiveBack(p);
  return true;
}

void OpticalFlowE::Destroy()
{
  for (CudaOperationBase* cuop : cuda_operations_) cuop->Destroy();
  ReleaseResult();
  size_t freed = 0;
  while (!free_containers_.empty()) {
    CheckDeviceError(f3d_free(free_containers_.back()));
    free_containers_.pop_back();
    ++freed;
  }
  if (sequence_frame_[2]) {  // sequence mode: the three frame containers, two of which resident_frame_ points at
    for (DevicePtr& p : sequence_frame_) {
      if (p) CheckDeviceError(f3d_free(p));
      p = 0;
    }
    resident_frame_[0] = resident_frame_[1] = 0;
  }
  for (DevicePtr& p : resident_frame_) {
    if (p) CheckDeviceError(f3d_free(p));
    p = 0;
  }
  if (freed && freed != kContainers + extra_containers_) std::printf("Warning. Not all device memory allocations were freed.\n");
  extra_containers_ = 0;
  initialized_ = false;
}
