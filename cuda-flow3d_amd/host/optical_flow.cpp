// Coarse-to-fine driver.  Sequence of operations, buffer counts, parameter keys and console lines follow
// OpticalFlowE::ComputeFlow (src/optical_flow/optical_flow_e.cpp:132-601) and OpticalFlowBase
// (src/optical_flow/optical_flow_base.cpp); the code is organised around a container pool and one
// RunPyramid() shared by the host-volume entry point and the device-resident one.
#include "optical_flow.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

#include "common_utils.h"
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
  bag.PushValuePtr("dev_flow_u", &result_flow_[0]);
  bag.PushValuePtr("dev_flow_v", &result_flow_[1]);
  bag.PushValuePtr("dev_flow_w", &result_flow_[2]);
  bag.PushValuePtr("data_size", &data_size);
  bag.PushValuePtr("stat", &stat);
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
  op.PushValuePtr("dev_frame_0", &resident_frame_[0]);
  op.PushValuePtr("dev_frame_1", &resident_frame_[1]);
  op.PushValuePtr("dev_flow_u", &result_flow_[0]);
  op.PushValuePtr("dev_flow_v", &result_flow_[1]);
  op.PushValuePtr("dev_flow_w", &result_flow_[2]);
  op.PushValuePtr("dev_output", &dev_temp);
  op.PushValuePtr("data_size", &size);
  op.PushValuePtr("hx", &h);
  op.PushValuePtr("hy", &h);
  op.PushValuePtr("hz", &h);
  cuop_register_.Execute(op);
  const bool ok = ResidualOf(resident_frame_[0], dev_temp, size, registered) &&
                  ResidualOf(resident_frame_[0], resident_frame_[1], size, unregistered);
  GiveBack(dev_temp);
  return ok;
}

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

  if (!silent) std::printf("\nStarting optical flow computation...\n");

  DataSize4 original = {dev_container_size_.width, dev_container_size_.height, dev_container_size_.depth, 0};
  const size_t max_warp_level = GetMaxWarpLevel(original.width, original.height, original.depth, warp_scale_factor);
  int level = static_cast<int>(std::min(warp_levels_count, max_warp_level)) - 1;

  OperationParameters op;

  // ---- pre-blur (optical_flow_e.cpp:213-242): full-size frames -> dev_frame_0/1 ------------------------------
  DevicePtr dev_frame_0, dev_frame_1;
  if (gaussian_sigma > 0.0) {
    dev_frame_0 = Borrow();
    dev_frame_1 = Borrow();
    DevicePtr dev_temp = Borrow();
    DevicePtr* src[2] = {&raw_0, &raw_1};
    DevicePtr* dst[2] = {&dev_frame_0, &dev_frame_1};
    for (int i = 0; i < 2; ++i) {
      op.Clear();
      op.PushValuePtr("dev_input", src[i]);
      op.PushValuePtr("dev_output", dst[i]);
      op.PushValuePtr("dev_temp", &dev_temp);
      op.PushValuePtr("data_size", &original);
      op.PushValuePtr("gaussian_sigma", &gaussian_sigma);
      cuop_convolution_.Execute(op);
    }
    GiveBack(dev_temp);
    if (raw_is_pooled) {
      GiveBack(raw_0);
      GiveBack(raw_1);
    }
  } else if (raw_is_pooled) {
    dev_frame_0 = raw_0;
    dev_frame_1 = raw_1;
  } else {
    dev_frame_0 = Borrow();
    dev_frame_1 = Borrow();
    const size_t bytes = dev_container_size_.pitch * dev_container_size_.height * dev_container_size_.depth;
    CheckDeviceError(f3d_copy_d2d(dev_frame_0, raw_0, bytes));
    CheckDeviceError(f3d_copy_d2d(dev_frame_1, raw_1, bytes));
  }

  DevicePtr dev_frame_0_res = Borrow(), dev_frame_1_res_br = Borrow();
  DevicePtr dev_flow_u = Borrow(), dev_flow_v = Borrow(), dev_flow_w = Borrow();
  DevicePtr dev_flow_du = Borrow(), dev_flow_dv = Borrow(), dev_flow_dw = Borrow();

  level_stats_.clear();
  DataSize4 prev_data_size = {0, 0, 0, 0};
  if (level < 0) {  // no level requested: the flow is identically zero
    const size_t rows = dev_container_size_.height * dev_container_size_.depth;
    for (DevicePtr p : {dev_flow_u, dev_flow_v, dev_flow_w})
      CheckDeviceError(f3d_memset2d(p, dev_container_size_.pitch, 0, dev_container_size_.width * sizeof(float), rows));
  }

  // `count` volumes of one size through the three passes together (three launches instead of 3 x count): a temp each
  auto resample = [&](DevicePtr* const* in, DevicePtr* const* out, size_t count, DataSize4& from, DataSize4& to) {
    OperationParameters bags[3];
    DevicePtr temps[3] = {0, 0, 0};
    for (size_t i = 0; i < count; ++i) {
      temps[i] = Borrow();
      bags[i].PushValuePtr("dev_input", in[i]);
      bags[i].PushValuePtr("dev_output", out[i]);
      bags[i].PushValuePtr("dev_temp", &temps[i]);
      bags[i].PushValuePtr("data_size", &from);
      bags[i].PushValuePtr("resample_size", &to);
    }
    cuop_resample_.ExecuteBatch(bags, count);
    for (size_t i = 0; i < count; ++i) GiveBack(temps[i]);
  };

  while (level >= 0) {
    PyramidLevel lv = GetLevel(original, warp_scale_factor, level);
    DataSize4 current = lv.size;
    float hx = lv.hx, hy = lv.hy, hz = lv.hz;
    char range_name[64];
    std::snprintf(range_name, sizeof(range_name), "level %d (%zu x %zu x %zu)", level, current.width, current.height, current.depth);
    ProfilerRange level_range(range_name);
    if (!silent)
      std::printf("Solve level %2d (%4zu x%4zu x%4zu) \n", level, current.width, current.height, current.depth);

    // frames of this level: always resampled from the ORIGINAL-size blurred frames (:274-300)
    if (level == 0) {
      std::swap(dev_frame_0, dev_frame_0_res);
      std::swap(dev_frame_1, dev_frame_1_res_br);
    } else {
      DevicePtr* const frames[2] = {&dev_frame_0, &dev_frame_1};
      DevicePtr* const resampled[2] = {&dev_frame_0_res, &dev_frame_1_res_br};
      resample(frames, resampled, 2, original, current);
    }

    // flow of the previous level brought to this size; values stay in original-voxel units (:303-345)
    if (prev_data_size.width == 0) {
      const size_t rows = dev_container_size_.height * dev_container_size_.depth;
      const size_t row_bytes = dev_container_size_.width * sizeof(float);
      CheckDeviceError(f3d_memset2d(dev_flow_u, dev_container_size_.pitch, 0, row_bytes, rows));
      CheckDeviceError(f3d_memset2d(dev_flow_v, dev_container_size_.pitch, 0, row_bytes, rows));
      CheckDeviceError(f3d_memset2d(dev_flow_w, dev_container_size_.pitch, 0, row_bytes, rows));
    } else {
      DevicePtr* const coarse[3] = {&dev_flow_u, &dev_flow_v, &dev_flow_w};
      DevicePtr* const fine[3] = {&dev_flow_du, &dev_flow_dv, &dev_flow_dw};
      resample(coarse, fine, 3, prev_data_size, current);
      std::swap(dev_flow_u, dev_flow_du);
      std::swap(dev_flow_v, dev_flow_dv);
      std::swap(dev_flow_w, dev_flow_dw);
    }

    // backward registration of frame 1 with the current flow (:348-369)
    {
      DevicePtr dev_temp = Borrow();
      op.Clear();
      op.PushValuePtr("dev_frame_0", &dev_frame_0_res);
      op.PushValuePtr("dev_frame_1", &dev_frame_1_res_br);
      op.PushValuePtr("dev_flow_u", &dev_flow_u);
      op.PushValuePtr("dev_flow_v", &dev_flow_v);
      op.PushValuePtr("dev_flow_w", &dev_flow_w);
      op.PushValuePtr("dev_output", &dev_temp);
      op.PushValuePtr("data_size", &current);
      op.PushValuePtr("hx", &hx);
      op.PushValuePtr("hy", &hy);
      op.PushValuePtr("hz", &hz);
      cuop_register_.Execute(op);
      std::swap(dev_frame_1_res_br, dev_temp);
      GiveBack(dev_temp);
    }
    if (collect_level_statistics) {
      LevelStatistics st;
      st.level = level;
      st.size = current;
      ResidualOf(dev_frame_0_res, dev_frame_1_res_br, current, st.before);
      level_stats_.push_back(st);
    }

    // difference problem: increments du, dv, dw (:372-417)
    {
      DevicePtr dev_phi = Borrow(), dev_ksi = Borrow();
      DevicePtr dev_temp_du = Borrow(), dev_temp_dv = Borrow(), dev_temp_dw = Borrow();
      op.Clear();
      op.PushValuePtr("dev_frame_0", &dev_frame_0_res);
      op.PushValuePtr("dev_frame_1", &dev_frame_1_res_br);
      op.PushValuePtr("dev_flow_u", &dev_flow_u);
      op.PushValuePtr("dev_flow_v", &dev_flow_v);
      op.PushValuePtr("dev_flow_w", &dev_flow_w);
      op.PushValuePtr("dev_flow_du", &dev_flow_du);
      op.PushValuePtr("dev_flow_dv", &dev_flow_dv);
      op.PushValuePtr("dev_flow_dw", &dev_flow_dw);
      op.PushValuePtr("dev_phi", &dev_phi);
      op.PushValuePtr("dev_ksi", &dev_ksi);
      op.PushValuePtr("dev_temp_du", &dev_temp_du);
      op.PushValuePtr("dev_temp_dv", &dev_temp_dv);
      op.PushValuePtr("dev_temp_dw", &dev_temp_dw);
      op.PushValuePtr("outer_iterations_count", &outer_iterations_count);
      op.PushValuePtr("inner_iterations_count", &inner_iterations_count);
      op.PushValuePtr("equation_alpha", &equation_alpha);
      op.PushValuePtr("equation_smoothness", &equation_smoothness);
      op.PushValuePtr("equation_data", &equation_data);
      op.PushValuePtr("data_size", &current);
      op.PushValuePtr("hx", &hx);
      op.PushValuePtr("hy", &hy);
      op.PushValuePtr("hz", &hz);
      cuop_solve_.silent = silent;
      cuop_solve_.Execute(op);
      GiveBack(dev_phi);
      GiveBack(dev_ksi);
      GiveBack(dev_temp_du);
      GiveBack(dev_temp_dv);
      GiveBack(dev_temp_dw);
    }

    // flow += increment (:420-438), then median of each component (:444-473)
    DevicePtr* flow[3] = {&dev_flow_u, &dev_flow_v, &dev_flow_w};
    DevicePtr* incr[3] = {&dev_flow_du, &dev_flow_dv, &dev_flow_dw};
    // the three components are independent through both steps: one launch each for the three of them.  The increments are
    // consumed by "+=", so their containers take the filtered flow and the roles are swapped (the reference filters through one
    // temp, a component at a time)
    {
      OperationParameters bags[3];
      for (int i = 0; i < 3; ++i) {
        bags[i].PushValuePtr("operand_0", flow[i]);
        bags[i].PushValuePtr("operand_1", incr[i]);
        bags[i].PushValuePtr("data_size", &current);
      }
      cuop_add_.ExecuteBatch(bags, 3);
    }
    {
      OperationParameters bags[3];
      for (int i = 0; i < 3; ++i) {
        bags[i].PushValuePtr("dev_input", flow[i]);
        bags[i].PushValuePtr("dev_output", incr[i]);
        bags[i].PushValuePtr("data_size", &current);
        bags[i].PushValuePtr("radius", &median_radius);
      }
      cuop_median_.ExecuteBatch(bags, 3);
      for (int i = 0; i < 3; ++i) std::swap(*flow[i], *incr[i]);
    }

    if (collect_level_statistics) {
      float mn = 0.f, mx = 0.f;
      double sum = 0.0;
      if (!CheckDeviceError(f3d_flow_stats(dev_flow_u, dev_flow_v, dev_flow_w, current.width, current.height, current.depth, nullptr,
                                           &mn, &mx, &sum))) {
        const double n = static_cast<double>(current.width) * static_cast<double>(current.height) * static_cast<double>(current.depth);
        level_stats_.back().flow = {mn, mx, static_cast<float>(sum / n)};
      }
      if (!silent) {
        const LevelStatistics& st = level_stats_.back();
        std::printf("  residual before the solve: rms %.5f  mean |.| %.5f  max %.4f;  flow after it: min %.4f  max %.4f  avg %.4f\n",
                    st.before.rms, st.before.mean_abs, st.before.max_abs, st.flow.min, st.flow.max, st.flow.avg);
      }
    }
    prev_data_size = current;
    --level;
  }

  // keep (u, v, w) until they are downloaded; everything else returns to the pool
  result_flow_[0] = dev_flow_u;
  result_flow_[1] = dev_flow_v;
  result_flow_[2] = dev_flow_w;
  GiveBack(dev_frame_0);
  GiveBack(dev_frame_1);
  GiveBack(dev_frame_0_res);
  GiveBack(dev_frame_1_res_br);
  GiveBack(dev_flow_du);
  GiveBack(dev_flow_dv);
  GiveBack(dev_flow_dw);
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
