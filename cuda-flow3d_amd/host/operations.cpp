// Host side of the six operators.  Parameter keys, argument meaning, sanity rules and the print-and-return
// error convention follow src/cuda_operations/entire_data/cuda_operation_*.cpp; every device action goes
// through the f3d C ABI.  Unlike the reference the solver does not synchronise the stream after every
// sweep (cuda_operation_solve.cpp:257): launches are queued back to back and the progress bar, when enabled,
// is refreshed once per outer iteration.
#include "operations.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <utility>

#include "common_utils.h"
#include "hip_utils.h"

// ---- base ------------------------------------------------------------------------------------------------

bool CudaOperationBase::IsInitialized() const
{
  if (!initialized_) std::printf("Error: Operation '%s' was not initialized.\n", name_);
  return initialized_;
}

void CudaOperationBase::Execute(OperationParameters&)
{
  std::printf("Warning: '%s' Execute() was not defined.\n", name_);
}

void CudaOperationBase::Destroy() { initialized_ = false; }

CudaOperationBase::~CudaOperationBase()
{
  if (initialized_) Destroy();
}

bool CudaOperationBase::InitializeContainer(const OperationParameters* params)
{
  initialized_ = false;
  if (!params) {
    std::printf("Operation: '%s'. Initialization parameters are missing.\n", GetName());
    return initialized_;
  }
  DataSize4 container_size;
  GET_PARAM_OR_RETURN_VALUE(*params, DataSize4, container_size, "container_size", initialized_);
  dev_container_size_ = container_size;
  f3d_size4 c = {container_size.width, container_size.height, container_size.depth, container_size.pitch};
  if (!CheckDeviceError(f3d_set_container(&c))) initialized_ = true;
  return initialized_;
}

// ---- add (cuda_operation_add.cpp:67-100) -----------------------------------------------------------------

void CudaOperationAdd::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  ProfilerRange range(GetName());
  DevicePtr operand_0, operand_1;
  DataSize4 data_size;
  GET_PARAM_OR_RETURN(params, DevicePtr, operand_0, "operand_0");
  GET_PARAM_OR_RETURN(params, DevicePtr, operand_1, "operand_1");
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  CheckDeviceError(f3d_add(operand_0, operand_1, data_size.width, data_size.height, data_size.depth, slab_));
}

namespace {
constexpr size_t kBatchMax = 3;  // volumes per launch of the f3d_*_n entries
bool SameBox(const DataSize4& a, const DataSize4& b) { return a.width == b.width && a.height == b.height && a.depth == b.depth; }
}  // namespace

void CudaOperationAdd::ExecuteBatch(OperationParameters* params, size_t count)
{
  if (!IsInitialized() || count == 0) return;
  DevicePtr a[kBatchMax], b[kBatchMax];
  DataSize4 size[kBatchMax];
  bool together = count <= kBatchMax;
  for (size_t i = 0; together && i < count; ++i) {
    void *p0 = params[i].GetValuePtr("operand_0"), *p1 = params[i].GetValuePtr("operand_1"), *ps = params[i].GetValuePtr("data_size");
    if (!p0 || !p1 || !ps) {
      together = false;  // Execute prints which key is missing
      break;
    }
    a[i] = *static_cast<DevicePtr*>(p0);
    b[i] = *static_cast<DevicePtr*>(p1);
    size[i] = *static_cast<DataSize4*>(ps);
    together = SameBox(size[i], size[0]);
    for (size_t j = 0; together && j < i; ++j) together = a[i] != a[j] && a[i] != b[j] && b[i] != a[j];
  }
  if (!together) {
    for (size_t i = 0; i < count; ++i) Execute(params[i]);
    return;
  }
  ProfilerRange range(GetName());
  CheckDeviceError(f3d_add_n(a, b, count, size[0].width, size[0].height, size[0].depth, slab_));
}

// ---- flow statistics (cuda_operation_stat_p.cpp:44-107, on device data) ------------------------------------------

void CudaOperationStat::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  ProfilerRange range(GetName());
  DevicePtr dev_flow_u, dev_flow_v, dev_flow_w;
  DataSize4 data_size;
  Stat3* p_stat;
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_flow_u, "dev_flow_u");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_flow_v, "dev_flow_v");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_flow_w, "dev_flow_w");
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  GET_PARAM_PTR_OR_RETURN(params, Stat3, p_stat, "stat");
  if (!silent) std::printf("Compute statistics...\n");
  double sum = 0.0;
  if (CheckDeviceError(f3d_flow_stats(dev_flow_u, dev_flow_v, dev_flow_w, data_size.width, data_size.height, data_size.depth, slab_,
                                      &p_stat->min, &p_stat->max, &sum)))
    return;
  const size_t planes = slab_ ? static_cast<size_t>(slab_->z_hi - slab_->z_lo) : data_size.depth;
  const double count = static_cast<double>(data_size.width) * static_cast<double>(data_size.height) * static_cast<double>(planes);
  p_stat->avg = count > 0 ? static_cast<float>(sum / count) : 0.f;
  if (!silent) std::printf("Min: %8.4f Max: %8.4f Avg: %8.4f\n", p_stat->min, p_stat->max, p_stat->avg);
}

// ---- Gaussian (cuda_operation_convolution.cpp:85-114, 134-184) -----------------------------------------------

void CudaOperationConvolution3D::ComputeGaussianKernel(float sigma, size_t precision, float pixel_size)
{
  kernel_radius_ = static_cast<size_t>(precision * sigma / pixel_size);
  kernel_length_ = 2 * kernel_radius_ + 1;
  if (kernel_length_ > kMaxKernelLength) {
    std::printf("Operation '%s': Gaussian radius %zu exceeds the supported %zu taps.\n", GetName(), kernel_radius_,
                kMaxKernelLength);
    kernel_length_ = 0;
    return;
  }
  const int r = static_cast<int>(kernel_radius_);
  // double-precision evaluation, pi truncated to 3.1415926, one rounding to float per tap (reference :93-96)
  const double norm = 1.0 / (sigma * std::sqrt(2.0 * 3.1415926));
  for (int i = -r; i <= r; ++i) {
    const float dist2 = i * i * pixel_size * pixel_size;
    kernel_[i + r] = static_cast<float>(norm * std::exp(-dist2 / (2.0 * sigma * sigma)));
  }
  float sum = 0.0f;  // normalised by the FLOAT sum
  for (size_t i = 0; i < kernel_length_; ++i) sum = sum + kernel_[i];
  for (size_t i = 0; i < kernel_length_; ++i) kernel_[i] = kernel_[i] / sum;
}

void CudaOperationConvolution3D::PrintConvolutionKernel() const
{
  if (kernel_length_ == 0) {
    std::printf("Error: Convolution kernel is not initialized.\n");
    return;
  }
  std::printf("Convolution kernel (radius = %zu)\n", kernel_radius_);
  for (size_t i = 0; i < kernel_length_; ++i) std::printf("%.4f ", kernel_[i]);
  std::printf("\n\n");
}

void CudaOperationConvolution3D::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  ProfilerRange range(GetName());
  DevicePtr dev_input = 0, dev_output = 0, dev_temp = 0;
  DataSize4 data_size;
  float gaussian_sigma;
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_input, "dev_input");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_output, "dev_output");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_temp, "dev_temp");
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  GET_PARAM_OR_RETURN(params, float, gaussian_sigma, "gaussian_sigma");
  if (dev_input == dev_output) {
    std::printf("Operation '%s': Error. Input buffer cannot serve as output buffer.", GetName());
    return;
  }
  ComputeGaussianKernel(gaussian_sigma, 3, 1.0);
  if (kernel_length_ == 0) return;
  if (CheckDeviceError(f3d_set_conv_taps(kernel_, kernel_length_))) return;
  const size_t w = data_size.width, h = data_size.height, d = data_size.depth, r = kernel_radius_;
  // The reference runs rows: input -> output, columns: output -> temp, slices: temp -> output (:172-181).  Rows and columns are
  // one launch here (the row-convolved volume stays in LDS): input -> temp, then slices: temp -> output -- the same bits in
  // dev_output, and dev_temp ends up holding the x/y-convolved volume as it does there.
  if (CheckDeviceError(f3d_conv_rows_cols(dev_temp, dev_input, w, h, d, r, slab_))) return;
  CheckDeviceError(f3d_conv_slices(dev_output, dev_temp, w, h, d, r, slab_));
}

// ---- median (cuda_operation_median.cpp:72-149) ------------------------------------------------------------

void CudaOperationMedian::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  ProfilerRange range(GetName());
  DevicePtr dev_input, dev_output;
  DataSize4 data_size;
  size_t radius;
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_input, "dev_input");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_output, "dev_output");
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  GET_PARAM_OR_RETURN(params, size_t, radius, "radius");
  if (dev_input == dev_output) {
    std::printf("Operation '%s': Error. Input buffer cannot serve as output buffer.", GetName());
    return;
  }
  if (radius == 1) {  // nothing to filter: copy the whole container
    CheckDeviceError(f3d_copy_d2d(dev_output, dev_input,
                                  dev_container_size_.pitch * dev_container_size_.height * dev_container_size_.depth));
    return;
  }
  if (radius % 2 == 0) {
    std::printf("Warning. Median raduis is even (%zu), decresaing by 1...\n", radius);
    radius -= 1;
  }
  if (radius >= 3 && radius <= 7) {
    CheckDeviceError(f3d_median(dev_input, data_size.width, data_size.height, data_size.depth, radius, dev_output, slab_));
  } else {
    std::printf("Error. Wrong median raduis (%zu). Supported values: 3, 5, 7\n", radius);
  }
}

void CudaOperationMedian::ExecuteBatch(OperationParameters* params, size_t count)
{
  if (!IsInitialized() || count == 0) return;
  DevicePtr in[kBatchMax], out[kBatchMax];
  DataSize4 size[kBatchMax];
  size_t radius[kBatchMax];
  bool together = count <= kBatchMax;
  for (size_t i = 0; together && i < count; ++i) {
    void *pi = params[i].GetValuePtr("dev_input"), *po = params[i].GetValuePtr("dev_output"), *ps = params[i].GetValuePtr("data_size"),
         *pr = params[i].GetValuePtr("radius");
    if (!pi || !po || !ps || !pr) {
      together = false;
      break;
    }
    in[i] = *static_cast<DevicePtr*>(pi);
    out[i] = *static_cast<DevicePtr*>(po);
    size[i] = *static_cast<DataSize4*>(ps);
    radius[i] = *static_cast<size_t*>(pr);
    // the copy (1), the even-window warning and the range error stay with Execute
    together = SameBox(size[i], size[0]) && radius[i] == radius[0] && (radius[i] == 3 || radius[i] == 5 || radius[i] == 7);
    for (size_t j = 0; together && j <= i; ++j) together = in[i] != out[j] && in[j] != out[i] && (j == i || out[i] != out[j]);
  }
  if (!together) {
    for (size_t i = 0; i < count; ++i) Execute(params[i]);
    return;
  }
  ProfilerRange range(GetName());
  CheckDeviceError(f3d_median_n(in, count, size[0].width, size[0].height, size[0].depth, radius[0], out, slab_));
}

// ---- warp (cuda_operation_registration.cpp:70-131) ----------------------------------------------------------

void CudaOperationRegistration::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  ProfilerRange range(GetName());
  DevicePtr dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, dev_flow_w, dev_output;
  float hx, hy, hz;
  DataSize4 data_size;
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_frame_0, "dev_frame_0");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_frame_1, "dev_frame_1");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_flow_u, "dev_flow_u");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_flow_v, "dev_flow_v");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_flow_w, "dev_flow_w");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_output, "dev_output");
  GET_PARAM_OR_RETURN(params, float, hx, "hx");
  GET_PARAM_OR_RETURN(params, float, hy, "hy");
  GET_PARAM_OR_RETURN(params, float, hz, "hz");
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  if (dev_frame_1 == dev_output) {
    std::printf("Operation '%s': Error. Input buffer cannot serve as output buffer.", GetName());
    return;
  }
  CheckDeviceError(f3d_warp(dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, dev_flow_w, data_size.width,
                            data_size.height, data_size.depth, hx, hy, hz, dev_output, slab_));
}

// ---- resample (cuda_operation_resample.cpp:72-175) ----------------------------------------------------------

void CudaOperationResample::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  ProfilerRange range(GetName());
  DevicePtr dev_input = 0, dev_output = 0, dev_temp = 0;
  DataSize4 data_size, resample_size;
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_input, "dev_input");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_output, "dev_output");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_temp, "dev_temp");
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  GET_PARAM_OR_RETURN(params, DataSize4, resample_size, "resample_size");
  if (dev_input == dev_output) {
    std::printf("Operation '%s': Error. Input buffer cannot serve as output buffer.", GetName());
    return;
  }
  // each pass shrinks/grows one axis; its grid spans the already-resampled axes at the new size and the
  // remaining ones at the old size
  DataSize4 pass_out = data_size;
  pass_out.width = resample_size.width;
  ResampleX(dev_input, dev_output, data_size, pass_out);
  data_size.width = pass_out.width;
  pass_out.height = resample_size.height;
  ResampleY(dev_output, dev_temp, data_size, pass_out);
  data_size.height = pass_out.height;
  pass_out.depth = resample_size.depth;
  ResampleZ(dev_temp, dev_output, data_size, pass_out);
}

void CudaOperationResample::ExecuteBatch(OperationParameters* params, size_t count)
{
  if (!IsInitialized() || count == 0) return;
  DevicePtr in[kBatchMax], out[kBatchMax], tmp[kBatchMax];
  DataSize4 from[kBatchMax], to[kBatchMax];
  bool together = count <= kBatchMax;
  for (size_t i = 0; together && i < count; ++i) {
    void *pi = params[i].GetValuePtr("dev_input"), *po = params[i].GetValuePtr("dev_output"), *pt = params[i].GetValuePtr("dev_temp"),
         *ps = params[i].GetValuePtr("data_size"), *pr = params[i].GetValuePtr("resample_size");
    if (!pi || !po || !pt || !ps || !pr) {
      together = false;
      break;
    }
    in[i] = *static_cast<DevicePtr*>(pi);
    out[i] = *static_cast<DevicePtr*>(po);
    tmp[i] = *static_cast<DevicePtr*>(pt);
    from[i] = *static_cast<DataSize4*>(ps);
    to[i] = *static_cast<DataSize4*>(pr);
    together = SameBox(from[i], from[0]) && SameBox(to[i], to[0]);
    // nine distinct volumes: x writes out, y reads out and writes tmp, z reads tmp and writes out
    for (size_t j = 0; together && j <= i; ++j) {
      const DevicePtr mine[3] = {in[i], out[i], tmp[i]}, theirs[3] = {in[j], out[j], tmp[j]};
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
          if (mine[a] == theirs[b] && !(i == j && a == b)) together = false;
    }
  }
  if (!together) {
    for (size_t i = 0; i < count; ++i) Execute(params[i]);
    return;
  }
  ProfilerRange range(GetName());
  // the pass order and the mixed sizes of Execute above
  const DataSize4 &s = from[0], &r = to[0];
  if (CheckDeviceError(f3d_resample_x_n(in, out, count, r.width, s.height, s.depth, s.width, slab_))) return;
  if (CheckDeviceError(f3d_resample_y_n(out, tmp, count, r.width, r.height, s.depth, s.height, slab_))) return;
  CheckDeviceError(f3d_resample_z_n(tmp, out, count, r.width, r.height, r.depth, s.depth, nullptr, slab_));
}

void CudaOperationResample::ResampleX(DevicePtr input, DevicePtr output, DataSize4& in, DataSize4& out) const
{
  CheckDeviceError(f3d_resample_x(input, output, out.width, out.height, out.depth, in.width, slab_));
}

void CudaOperationResample::ResampleY(DevicePtr input, DevicePtr output, DataSize4& in, DataSize4& out) const
{
  CheckDeviceError(f3d_resample_y(input, output, out.width, out.height, out.depth, in.height, slab_));
}

void CudaOperationResample::ResampleZ(DevicePtr input, DevicePtr output, DataSize4& in, DataSize4& out) const
{
  CheckDeviceError(f3d_resample_z(input, output, out.width, out.height, out.depth, in.depth, nullptr, slab_));
}

// ---- solve (cuda_operation_solve.cpp:75-281) ------------------------------------------------------------------

bool CudaOperationSolve::ScratchFits() const
{
  return scratch_size_.width == dev_container_size_.width && scratch_size_.height == dev_container_size_.height &&
         scratch_size_.depth == dev_container_size_.depth && scratch_size_.pitch == dev_container_size_.pitch;
}

void CudaOperationSolve::FreeScratch()
{
  if (phi_alt_) f3d_free(phi_alt_);
  if (ksi_alt_) f3d_free(ksi_alt_);
  phi_alt_ = ksi_alt_ = 0;
  for (DevicePtr& p : fder_) {
    if (p) f3d_free(p);
    p = 0;
  }
  scratch_size_ = {0, 0, 0, 0};
}

bool CudaOperationSolve::Initialize(const OperationParameters* params)
{
  const bool ok = InitializeContainer(params);
  // scratch of another container (an earlier Initialize without Destroy in between) would be too small or mis-pitched for the
  // fused launches, which write whole container-sized volumes into it
  if (!ScratchFits()) FreeScratch();
  return ok;
}

size_t CudaOperationSolve::ScratchVolumes()
{
  size_t n = 0;
  if (FusedSweepsEnabled() && FusedPhiKsiEnabled()) n += 2;
  if (FusedSweepsEnabled() && FrameDerivativesEnabled()) n += 4;
  return n;
}

bool CudaOperationSolve::EnsureWeightScratch()
{
  if (!ScratchFits()) FreeScratch();
  if (phi_alt_ && ksi_alt_) return true;
  scratch_size_ = dev_container_size_;
  const size_t rows = dev_container_size_.height * dev_container_size_.depth;
  for (DevicePtr* p : {&phi_alt_, &ksi_alt_}) {
    if (*p) continue;
    size_t pitch = 0;
    if (f3d_alloc_pitched(p, &pitch, dev_container_size_.width * sizeof(float), rows) != 0 || pitch != dev_container_size_.pitch) {
      if (*p) f3d_free(*p);
      *p = 0;
      return false;  // no room (or another pitch): the unfused schedule needs no scratch
    }
  }
  return true;
}

bool CudaOperationSolve::EnsureDerivativeScratch()
{
  if (!ScratchFits()) FreeScratch();
  scratch_size_ = dev_container_size_;
  const size_t rows = dev_container_size_.height * dev_container_size_.depth;
  for (DevicePtr& p : fder_) {
    if (p) continue;
    size_t pitch = 0;
    if (f3d_alloc_pitched(&p, &pitch, dev_container_size_.width * sizeof(float), rows) != 0 || pitch != dev_container_size_.pitch) {
      if (p) f3d_free(p);
      p = 0;
      return false;
    }
  }
  return true;
}

void CudaOperationSolve::Destroy()
{
  FreeScratch();
  CudaOperationBase::Destroy();
}

void CudaOperationSolve::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  ProfilerRange range(GetName());

  DevicePtr dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, dev_flow_w, dev_phi, dev_ksi;
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_frame_0, "dev_frame_0");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_frame_1, "dev_frame_1");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_flow_u, "dev_flow_u");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_flow_v, "dev_flow_v");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_flow_w, "dev_flow_w");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_phi, "dev_phi");
  GET_PARAM_OR_RETURN(params, DevicePtr, dev_ksi, "dev_ksi");

  // increments and their ping-pong partners are read BY POINTER so the caller sees the final roles
  DevicePtr *du_ptr, *dv_ptr, *dw_ptr, *tdu_ptr, *tdv_ptr, *tdw_ptr;
  GET_PARAM_PTR_OR_RETURN(params, DevicePtr, du_ptr, "dev_flow_du");
  GET_PARAM_PTR_OR_RETURN(params, DevicePtr, dv_ptr, "dev_flow_dv");
  GET_PARAM_PTR_OR_RETURN(params, DevicePtr, dw_ptr, "dev_flow_dw");
  GET_PARAM_PTR_OR_RETURN(params, DevicePtr, tdu_ptr, "dev_temp_du");
  GET_PARAM_PTR_OR_RETURN(params, DevicePtr, tdv_ptr, "dev_temp_dv");
  GET_PARAM_PTR_OR_RETURN(params, DevicePtr, tdw_ptr, "dev_temp_dw");

  size_t outer_iterations_count, inner_iterations_count;
  float equation_alpha, equation_smoothness, equation_data, hx, hy, hz;
  DataSize4 data_size;
  GET_PARAM_OR_RETURN(params, size_t, outer_iterations_count, "outer_iterations_count");
  GET_PARAM_OR_RETURN(params, size_t, inner_iterations_count, "inner_iterations_count");
  GET_PARAM_OR_RETURN(params, float, equation_alpha, "equation_alpha");
  GET_PARAM_OR_RETURN(params, float, equation_smoothness, "equation_smoothness");
  GET_PARAM_OR_RETURN(params, float, equation_data, "equation_data");
  GET_PARAM_OR_RETURN(params, float, hx, "hx");
  GET_PARAM_OR_RETURN(params, float, hy, "hy");
  GET_PARAM_OR_RETURN(params, float, hz, "hz");
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  NoteSolveWeights(equation_alpha, hx, hy, hz);

  // increments start from zero at every level: the level's box, one launch for the three (the reference clears every row of every
  // plane of the container, :183-188; nothing reads outside the box -- rows and planes mirror by address inside it)
  {
    const DevicePtr increments[3] = {*du_ptr, *dv_ptr, *dw_ptr};
    // under a z-slab window the neighbours' planes held by the container are cleared too, as the row-wise clearing did
    // (a container whose first plane lies BEFORE the volume -- rank 0 of the z-slab driver, z_base = own.lo - halo < 0 -- holds
    // no data there: the window is the part of the container inside the volume)
    f3d_slab all = {0, 0, 0};
    if (slab_) {
      const long top = static_cast<long>(slab_->z_base) + static_cast<long>(dev_container_size_.depth);
      all = {slab_->z_base, std::max(0, slab_->z_base), static_cast<int>(std::min<long>(static_cast<long>(data_size.depth), top))};
    }
    if (slab_ && all.z_lo >= all.z_hi) return;  // the container holds no plane of this level
    if (CheckDeviceError(f3d_clear_box_n(increments, 3, data_size.width, data_size.height, data_size.depth, slab_ ? &all : nullptr)))
      return;  // sweeps on uncleared increments would be a wrong result, not a slow one
  }

  f3d_event ev_start = nullptr, ev_stop = nullptr;
  if (!silent) {
    CheckDeviceError(f3d_event_create(&ev_start));
    CheckDeviceError(f3d_event_create(&ev_stop));
    CheckDeviceError(f3d_event_record(ev_start));
    Utils::PrintProgressBar(0.f);
    std::printf(" % 3.0f%%", 0.f);
  }

  const size_t w = data_size.width, h = data_size.height, d = data_size.depth;
  // Launch schedule of one outer iteration (the same bit pattern whichever way it is cut):
  //   phi/ksi (first outer iteration of a level only), then the sweeps in fused groups -- three per launch on small and mid-size
  //   levels (f3d_solve_sweep3), two per launch elsewhere (f3d_solve_sweep2) -- with one buffer swap per launch; when another outer
  //   iteration follows, the last launch also computes the phi/ksi of the NEXT iteration into the second weight pair
  //   (f3d_solve_sweep2_phi_ksi behind two sweeps, f3d_solve_sweep_phi_ksi behind one).  Defaults (5 sweeps): 2 launches per outer
  //   iteration on levels up to ~144^3, 3 above, instead of the reference's 6.
  const bool fuse_weights = FusedSweepsEnabled() && FusedPhiKsiEnabled() && outer_iterations_count > 1 &&
                            dev_container_size_.pitch % 256 == 0 && EnsureWeightScratch();
  // Small and mid-size levels take the three-stage launches (k_tri: no z-slab windows yet, frames not derivatives): with them the
  // last launch of an outer iteration that is not the level's last can carry the next weights behind TWO sweeps, so the default five
  // sweeps are (S, S, S) + (S, S, P); without them it carries them behind one, and only an odd count ends that way.
  const bool tri = FusedSweepsEnabled() && slab_ == nullptr && dev_container_size_.pitch % 256 == 0 && inner_iterations_count >= 2 &&
                   ThreeStageLaunchesPay(w, h, d);
  // How the `inner` sweeps of an outer iteration are cut into launches: groups of 3 / 2 / 1 sweeps, the last group taking the next
  // weights along when another outer iteration follows (`more`) and the group is 2 (three-stage) or 1 sweeps.
  auto next_group = [&](size_t remaining, bool more) -> size_t {
    if (!FusedSweepsEnabled()) return 1;
    const bool weights_wanted = more && fuse_weights;
    if (tri) {
      if (weights_wanted) {
        if (remaining == 2) return 2;                 // (S, S, P)
        if (remaining == 3) return 1;                 // S, then (S, S, P)
        if (remaining == 4) return 2;                 // (S, S), then (S, S, P)
      }
      if (remaining >= 3) return 3;
    }
    if (weights_wanted && remaining == 3 && !tri) return 2;   // (S, S), then (S, P)
    return remaining >= 2 ? 2 : 1;
  };
  // the weights of the last outer iteration must end in the caller's dev_phi / dev_ksi (as in the reference): every outer iteration
  // whose last group carries the next weights hands over to the other pair once; count the hand-overs and start on the right side
  auto carries_weights = [&](size_t group, size_t remaining_after, bool more) {
    return more && fuse_weights && remaining_after == 0 && ((tri && group == 2) || group == 1);
  };
  size_t hand_overs = 0;
  for (size_t i = 0; i + 1 < outer_iterations_count; ++i) {
    size_t remaining = inner_iterations_count, group = 0;
    while (remaining > 0) {
      group = next_group(remaining, true);
      remaining -= group;
    }
    if (carries_weights(group, 0, true)) ++hand_overs;
  }
  DevicePtr phi_cur = dev_phi, ksi_cur = dev_ksi, phi_nxt = phi_alt_, ksi_nxt = ksi_alt_;
  if (hand_overs % 2 == 1) {
    std::swap(phi_cur, phi_nxt);
    std::swap(ksi_cur, ksi_nxt);
  }
  bool weights_ready = false;
  // The frame derivatives fx, fy, fz, ft depend on the two frames only: computed once here, read by every two-stage fused launch of the
  // level instead of the frames (the reference recomputes them for every voxel in each of its 240 launches per level).
  // (thin volumes: the launches on frames march along y with all planes in the tile, two workgroups per CU -- faster than the z march
  // the derivative builds would take, so such a level stays on the frames)
  const bool on_derivatives = !tri && FusedSweepsEnabled() && FrameDerivativesEnabled() && slab_ == nullptr && inner_iterations_count >= 2 &&
                              f3d_fused_launches_march_along_y(w, h, d) == 0 &&
                              dev_container_size_.pitch % 256 == 0 && EnsureDerivativeScratch() &&
                              !CheckDeviceError(f3d_frame_derivatives(dev_frame_0, dev_frame_1, w, h, d, hx, hy, hz, fder_[0], fder_[1],
                                                                      fder_[2], fder_[3], nullptr));
  for (size_t i = 0; i < outer_iterations_count; ++i) {
    if (!weights_ready &&
        CheckDeviceError(f3d_phi_ksi(dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, dev_flow_w, *du_ptr, *dv_ptr, *dw_ptr, w, h, d,
                                     hx, hy, hz, equation_smoothness, equation_data, phi_cur, ksi_cur, slab_)))
      return;
    weights_ready = false;
    const bool more = i + 1 < outer_iterations_count;
    for (size_t j = 0; j < inner_iterations_count;) {
      const size_t group = next_group(inner_iterations_count - j, more);
      const bool with_weights = carries_weights(group, inner_iterations_count - j - group, more);
      int status;
      if (group == 3)
        status = f3d_solve_sweep3(dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, dev_flow_w, *du_ptr, *dv_ptr, *dw_ptr, phi_cur, ksi_cur,
                                  w, h, d, hx, hy, hz, equation_alpha, *tdu_ptr, *tdv_ptr, *tdw_ptr, slab_);
      else if (group == 2 && with_weights)
        status = f3d_solve_sweep2_phi_ksi(dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, dev_flow_w, *du_ptr, *dv_ptr, *dw_ptr, phi_cur,
                                          ksi_cur, w, h, d, hx, hy, hz, equation_alpha, equation_smoothness, equation_data, *tdu_ptr,
                                          *tdv_ptr, *tdw_ptr, phi_nxt, ksi_nxt, slab_);
      else if (group == 2 && on_derivatives)
        status = f3d_solve_sweep2_fd(fder_[0], fder_[1], fder_[2], fder_[3], dev_flow_u, dev_flow_v, dev_flow_w, *du_ptr, *dv_ptr,
                                     *dw_ptr, phi_cur, ksi_cur, w, h, d, hx, hy, hz, equation_alpha, *tdu_ptr, *tdv_ptr, *tdw_ptr, slab_);
      else if (group == 2)
        status = f3d_solve_sweep2(dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, dev_flow_w, *du_ptr, *dv_ptr, *dw_ptr, phi_cur,
                                  ksi_cur, w, h, d, hx, hy, hz, equation_alpha, *tdu_ptr, *tdv_ptr, *tdw_ptr, slab_);
      else if (with_weights && on_derivatives)
        status = f3d_solve_sweep_phi_ksi_fd(fder_[0], fder_[1], fder_[2], fder_[3], dev_flow_u, dev_flow_v, dev_flow_w, *du_ptr,
                                            *dv_ptr, *dw_ptr, phi_cur, ksi_cur, w, h, d, hx, hy, hz, equation_alpha,
                                            equation_smoothness, equation_data, *tdu_ptr, *tdv_ptr, *tdw_ptr, phi_nxt, ksi_nxt, slab_);
      else if (with_weights)
        status = f3d_solve_sweep_phi_ksi(dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, dev_flow_w, *du_ptr, *dv_ptr, *dw_ptr,
                                         phi_cur, ksi_cur, w, h, d, hx, hy, hz, equation_alpha, equation_smoothness, equation_data,
                                         *tdu_ptr, *tdv_ptr, *tdw_ptr, phi_nxt, ksi_nxt, slab_);
      else
        status = f3d_solve_sweep(dev_frame_0, dev_frame_1, dev_flow_u, dev_flow_v, dev_flow_w, *du_ptr, *dv_ptr, *dw_ptr, phi_cur,
                                 ksi_cur, w, h, d, hx, hy, hz, equation_alpha, *tdu_ptr, *tdv_ptr, *tdw_ptr, slab_);
      if (CheckDeviceError(status)) return;
      std::swap(*du_ptr, *tdu_ptr);
      std::swap(*dv_ptr, *tdv_ptr);
      std::swap(*dw_ptr, *tdw_ptr);
      if (with_weights) {
        std::swap(phi_cur, phi_nxt);
        std::swap(ksi_cur, ksi_nxt);
        weights_ready = true;
      }
      j += group;
    }
    if (!silent) {
      CheckDeviceError(f3d_stream_sync());
      const float complete = static_cast<float>(i + 1) / static_cast<float>(outer_iterations_count);
      Utils::PrintProgressBar(complete);
      std::printf(" % 3.0f%%", complete * 100);
    }
  }

  if (!silent) {
    float elapsed_ms = 0.f;
    CheckDeviceError(f3d_event_record(ev_stop));
    CheckDeviceError(f3d_event_sync(ev_stop));
    CheckDeviceError(f3d_event_elapsed_ms(&elapsed_ms, ev_start, ev_stop));
    Utils::PrintProgressBar(1.f);
    std::printf(" %8.4fs\n", elapsed_ms / 1000.);
    f3d_event_destroy(ev_start);
    f3d_event_destroy(ev_stop);
  }
}
