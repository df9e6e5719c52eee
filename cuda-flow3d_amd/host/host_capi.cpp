// C ABI of the host side (include/f3d_host.h): thin, exception-free wrappers over the C++ classes.
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "f3d_host.h"
#include "hip_utils.h"
#include "operations.h"
#include "operations_p.h"
#include "optical_flow.h"
#include "optical_flow_p.h"
#include "optical_flow_slab.h"
#include "synth.h"

struct f3d_flow_s {
  OpticalFlowE driver;
  bool device_ready = false;
};

struct f3d_slabflow_s {
  OpticalFlowSlab* driver = nullptr;
  ~f3d_slabflow_s() { delete driver; }
};

struct f3d_pflow_s {
  OpticalFlowP driver;
};

struct f3d_volume_s {
  Data3D view;
  f3d_volume_s(float* data, size_t w, size_t h, size_t d) : view(data, w, h, d) {}
};

struct f3d_op_s {
  CudaOperationBase* op = nullptr;
  ~f3d_op_s() { delete op; }
};

namespace {

void FillBag(OperationParameters& bag, f3d_flow_params& p)
{
  bag.PushValuePtr("warp_levels_count", &p.warp_levels_count);
  bag.PushValuePtr("warp_scale_factor", &p.warp_scale_factor);
  bag.PushValuePtr("outer_iterations_count", &p.outer_iterations_count);
  bag.PushValuePtr("inner_iterations_count", &p.inner_iterations_count);
  bag.PushValuePtr("equation_alpha", &p.equation_alpha);
  bag.PushValuePtr("equation_smoothness", &p.equation_smoothness);
  bag.PushValuePtr("equation_data", &p.equation_data);
  bag.PushValuePtr("median_radius", &p.median_radius);
  bag.PushValuePtr("gaussian_sigma", &p.gaussian_sigma);
}

}  // namespace

extern "C" {

void f3d_flow_default_params(f3d_flow_params* p)
{
  p->warp_levels_count = 40;
  p->warp_scale_factor = 0.95f;
  p->outer_iterations_count = 40;
  p->inner_iterations_count = 5;
  p->equation_alpha = 7.5f;
  p->equation_smoothness = 0.001f;
  p->equation_data = 0.001f;
  p->median_radius = 5;
  p->gaussian_sigma = 2.0f;
}

int f3d_flow_create(f3d_flow* flow)
{
  if (!flow) return 1;
  *flow = new (std::nothrow) f3d_flow_s;
  return *flow ? 0 : 1;
}

int f3d_flow_initialize(f3d_flow flow, size_t width, size_t height, size_t depth)
{
  if (!flow) return 1;
  if (f3d_init(-1) != 0) {
    std::fprintf(stderr, "f3d_flow_initialize: %s\n", f3d_last_error());
    return 1;
  }
  DataSize4 size = {width, height, depth, 0};
  return flow->driver.Initialize(size) ? 0 : 1;
}

int f3d_flow_compute(f3d_flow flow, const float* frame_0, const float* frame_1, const f3d_flow_params* params,
                     int silent, float* u, float* v, float* w)
{
  if (!flow || !frame_0 || !frame_1 || !params || !u || !v || !w) return 1;
  const DataSize4& c = flow->driver.ContainerSize();
  Data3D f0(const_cast<float*>(frame_0), c.width, c.height, c.depth);
  Data3D f1(const_cast<float*>(frame_1), c.width, c.height, c.depth);
  Data3D fu(u, c.width, c.height, c.depth), fv(v, c.width, c.height, c.depth), fw(w, c.width, c.height, c.depth);
  f3d_flow_params p = *params;
  OperationParameters bag;
  FillBag(bag, p);
  flow->driver.silent = silent != 0;
  flow->driver.ComputeFlow(f0, f1, fu, fv, fw, bag);
  return 0;
}

int f3d_flow_upload(f3d_flow flow, const float* frame_0, const float* frame_1)
{
  if (!flow || !frame_0 || !frame_1) return 1;
  const DataSize4& c = flow->driver.ContainerSize();
  Data3D f0(const_cast<float*>(frame_0), c.width, c.height, c.depth);
  Data3D f1(const_cast<float*>(frame_1), c.width, c.height, c.depth);
  if (!flow->driver.AllocateResidentFrames()) return 1;
  flow->driver.UploadResidentFrames(f0, f1);
  return 0;
}

int f3d_flow_compute_resident(f3d_flow flow, const f3d_flow_params* params, int silent, float* device_seconds)
{
  if (!flow || !params) return 1;
  f3d_flow_params p = *params;
  OperationParameters bag;
  FillBag(bag, p);
  flow->driver.silent = silent != 0;
  flow->driver.ComputeFlowResident(bag);
  if (device_seconds) *device_seconds = flow->driver.LastDeviceSeconds();
  return 0;
}

int f3d_flow_download(f3d_flow flow, float* u, float* v, float* w)
{
  if (!flow || !u || !v || !w) return 1;
  const DataSize4& c = flow->driver.ContainerSize();
  Data3D fu(u, c.width, c.height, c.depth), fv(v, c.width, c.height, c.depth), fw(w, c.width, c.height, c.depth);
  flow->driver.DownloadFlow(fu, fv, fw);
  return 0;
}

int f3d_flow_container(f3d_flow flow, f3d_size4* container)
{
  if (!flow || !container) return 1;
  const DataSize4& c = flow->driver.ContainerSize();
  *container = {c.width, c.height, c.depth, c.pitch};
  return 0;
}

int f3d_flow_set_level_stats(f3d_flow flow, int enable)
{
  if (!flow) return 1;
  flow->driver.collect_level_statistics = enable != 0;
  return 0;
}

int f3d_flow_level_stat_count(f3d_flow flow, size_t* count)
{
  if (!flow || !count) return 1;
  *count = flow->driver.LevelStats().size();
  return 0;
}

int f3d_flow_level_stat(f3d_flow flow, size_t index, f3d_level_stat* out)
{
  if (!flow || !out || index >= flow->driver.LevelStats().size()) return 1;
  const OpticalFlowE::LevelStatistics& st = flow->driver.LevelStats()[index];
  out->level = st.level;
  out->width = st.size.width;
  out->height = st.size.height;
  out->depth = st.size.depth;
  out->residual_rms = st.before.rms;
  out->residual_mean_abs = st.before.mean_abs;
  out->residual_max_abs = st.before.max_abs;
  out->flow_min = st.flow.min;
  out->flow_max = st.flow.max;
  out->flow_avg = st.flow.avg;
  return 0;
}

int f3d_flow_final_residual(f3d_flow flow, double registered[3], double unregistered[3])
{
  if (!flow || !registered || !unregistered) return 1;
  OpticalFlowE::Residual a, b;
  if (!flow->driver.FinalResidual(a, b)) return 1;
  registered[0] = a.rms; registered[1] = a.mean_abs; registered[2] = a.max_abs;
  unregistered[0] = b.rms; unregistered[1] = b.mean_abs; unregistered[2] = b.max_abs;
  return 0;
}

int f3d_flow_destroy(f3d_flow flow)
{
  delete flow;
  return 0;
}

int f3d_op_create(f3d_op* op, const char* name)
{
  if (!op || !name) return 1;
  const std::string n(name);
  CudaOperationBase* impl = nullptr;
  if (n == "add") impl = new CudaOperationAdd;
  else if (n == "convolution") impl = new CudaOperationConvolution3D;
  else if (n == "median") impl = new CudaOperationMedian;
  else if (n == "registration") impl = new CudaOperationRegistration;
  else if (n == "resample") impl = new CudaOperationResample;
  else if (n == "solve") impl = new CudaOperationSolve;
  else if (n == "stat") impl = new CudaOperationStat;
  else if (n == "add_p") impl = new CudaOperationAddP;
  else if (n == "resample_p") impl = new CudaOperationResampleP;
  else if (n == "registration_p") impl = new CudaOperationRegistrationP;
  else if (n == "solve_p") impl = new CudaOperationSolveP;
  else if (n == "stat_p") impl = new CudaOperationStatP;
  else if (n == "convolution_p") impl = new CudaOperationConvolution3DP;
  else if (n == "median_p") impl = new CudaOperationMedianP;
  if (!impl) return 1;
  *op = new f3d_op_s;
  (*op)->op = impl;
  return 0;
}

const char* f3d_op_name(f3d_op op) { return op ? op->op->GetName() : ""; }

int f3d_op_initialize(f3d_op op, const f3d_size4* container_size)
{
  if (!op) return 1;
  if (!container_size) return op->op->Initialize(nullptr) ? 0 : 1;
  DataSize4 c = {container_size->width, container_size->height, container_size->depth, container_size->pitch};
  OperationParameters bag;
  bag.PushValuePtr("container_size", &c);
  return op->op->Initialize(&bag) ? 0 : 1;
}

int f3d_op_execute(f3d_op op, const char* const* keys, void* const* value_ptrs, size_t count)
{
  if (!op) return 1;
  OperationParameters bag;
  for (size_t i = 0; i < count; ++i) bag.PushValuePtr(keys[i], value_ptrs[i]);
  if (auto* solve = dynamic_cast<CudaOperationSolve*>(op->op)) solve->silent = true;
  if (auto* solve_p = dynamic_cast<CudaOperationSolveP*>(op->op)) solve_p->silent = true;
  if (auto* stat_p = dynamic_cast<CudaOperationStatP*>(op->op)) stat_p->silent = true;
  op->op->Execute(bag);
  return 0;
}

int f3d_op_execute_batch(f3d_op op, const char* const* keys, void* const* value_ptrs, const size_t* counts, size_t bags)
{
  if (!op || !counts || bags == 0) return 1;
  std::vector<OperationParameters> bag(bags);
  size_t at = 0;
  for (size_t b = 0; b < bags; ++b)
    for (size_t i = 0; i < counts[b]; ++i, ++at) bag[b].PushValuePtr(keys[at], value_ptrs[at]);
  if (auto* add = dynamic_cast<CudaOperationAdd*>(op->op)) return add->ExecuteBatch(bag.data(), bags), 0;
  if (auto* median = dynamic_cast<CudaOperationMedian*>(op->op)) return median->ExecuteBatch(bag.data(), bags), 0;
  if (auto* resample = dynamic_cast<CudaOperationResample*>(op->op)) return resample->ExecuteBatch(bag.data(), bags), 0;
  return 1;  // the other operators take one bag at a time
}

int f3d_op_set_slab(f3d_op op, const f3d_slab* slab)
{
  if (!op) return 1;
  op->op->SetSlab(slab);
  return 0;
}

int f3d_op_destroy(f3d_op op)
{
  if (op) op->op->Destroy();
  delete op;
  return 0;
}

int f3d_op_solve_p_last(f3d_op op, int* chunk, int* outer_per_pass, int* halo, size_t* passes, int* overlapped)
{
  auto* solve_p = op ? dynamic_cast<CudaOperationSolveP*>(op->op) : nullptr;
  if (!solve_p) return 1;
  if (chunk) *chunk = solve_p->LastPlan().chunk;
  if (outer_per_pass) *outer_per_pass = solve_p->LastPlan().outer_per_pass;
  if (halo) *halo = solve_p->LastPlan().halo;
  if (passes) *passes = solve_p->LastPasses();
  if (overlapped) *overlapped = solve_p->LastPlan().overlapped ? 1 : 0;
  return 0;
}

int f3d_op_solve_p_fused_weights(f3d_op op, int* fused)
{
  auto* solve_p = op ? dynamic_cast<CudaOperationSolveP*>(op->op) : nullptr;
  if (!solve_p || !fused) return 1;
  *fused = solve_p->LastFusedWeights() ? 1 : 0;
  return 0;
}

int f3d_volume_wrap(f3d_volume* vol, float* data, size_t width, size_t height, size_t depth)
{
  if (!vol || !data || width == 0 || height == 0 || depth == 0) return 1;
  *vol = new (std::nothrow) f3d_volume_s(data, width, height, depth);
  return *vol ? 0 : 1;
}

void* f3d_volume_object(f3d_volume vol) { return vol ? &vol->view : nullptr; }

float* f3d_volume_data(f3d_volume vol) { return vol ? vol->view.DataPtr() : nullptr; }

int f3d_volume_destroy(f3d_volume vol)
{
  delete vol;
  return 0;
}

int f3d_pflow_create(f3d_pflow* flow)
{
  if (!flow) return 1;
  *flow = new (std::nothrow) f3d_pflow_s;
  return *flow ? 0 : 1;
}

int f3d_pflow_initialize(f3d_pflow flow, size_t width, size_t height, size_t depth)
{
  if (!flow) return 1;
  if (f3d_init(-1) != 0) {
    std::fprintf(stderr, "f3d_pflow_initialize: %s\n", f3d_last_error());
    return 1;
  }
  DataSize4 size = {width, height, depth, 0};
  return flow->driver.Initialize(size) ? 0 : 1;
}

int f3d_pflow_compute(f3d_pflow flow, const float* frame_0, const float* frame_1, size_t width, size_t height, size_t depth,
                      const f3d_flow_params* params, int silent, float* u, float* v, float* w, float* device_seconds)
{
  if (!flow || !frame_0 || !frame_1 || !params || !u || !v || !w) return 1;
  Data3D f0(const_cast<float*>(frame_0), width, height, depth), f1(const_cast<float*>(frame_1), width, height, depth);
  Data3D fu(u, width, height, depth), fv(v, width, height, depth), fw(w, width, height, depth);
  f3d_flow_params p = *params;
  OperationParameters bag;
  FillBag(bag, p);
  flow->driver.silent = silent != 0;
  flow->driver.ComputeFlow(f0, f1, fu, fv, fw, bag);
  if (device_seconds) *device_seconds = flow->driver.LastDeviceSeconds();
  // the driver hands every caller volume its own storage back; anything else would lose the result
  return (f0.DataPtr() == frame_0 && f1.DataPtr() == frame_1 && fu.DataPtr() == u && fv.DataPtr() == v && fw.DataPtr() == w) ? 0 : 1;
}

int f3d_pflow_stats(f3d_pflow flow, size_t* solve_passes, size_t* streamed_levels, size_t* resident_levels)
{
  if (!flow) return 1;
  if (solve_passes) *solve_passes = flow->driver.LastSolvePasses();
  if (streamed_levels) *streamed_levels = flow->driver.LastStreamedLevels();
  if (resident_levels) *resident_levels = flow->driver.LastResidentLevels();
  return 0;
}

int f3d_pflow_levels_registered_inside(f3d_pflow flow, size_t* levels)
{
  if (!flow || !levels) return 1;
  *levels = flow->driver.LastLevelsRegisteredInside();
  return 0;
}

int f3d_pflow_levels_with_constants_on_device(f3d_pflow flow, size_t* levels)
{
  if (!flow || !levels) return 1;
  *levels = flow->driver.LastLevelsWithConstantsOnDevice();
  return 0;
}

int f3d_pflow_originals_on_device(f3d_pflow flow, int* yes)
{
  if (!flow || !yes) return 1;
  *yes = flow->driver.LastOriginalsOnDevice() ? 1 : 0;
  return 0;
}

int f3d_pflow_set_full_pipeline(f3d_pflow flow, int enabled)
{
  if (!flow) return 1;
  flow->driver.full_pipeline = enabled != 0;
  return 0;
}

int f3d_pflow_set_resident(f3d_pflow flow, int enabled)
{
  if (!flow) return 1;
  flow->driver.resident_coarse_levels = enabled != 0;
  return 0;
}

int f3d_pflow_operator_seconds(f3d_pflow flow, double* seconds6)
{
  if (!flow || !seconds6) return 1;
  for (int i = 0; i < 6; ++i) seconds6[i] = flow->driver.LastOperatorSeconds()[i];
  return 0;
}

int f3d_pflow_destroy(f3d_pflow flow)
{
  delete flow;
  return 0;
}

size_t f3d_piecemeal_budget_bytes(void) { return PiecemealBudgetBytes(); }

int f3d_plan_solve_piecemeal(size_t budget_bytes, size_t width, size_t height, int depth, int inner_iterations, int outer_iterations,
                             int forced_outer_per_pass, int overlap_mode, int* chunk, int* outer_per_pass, int* halo, int* max_planes,
                             int* overlapped)
{
  const SolvePiecemealPlan plan = PlanSolvePiecemeal(budget_bytes, width, height, depth, inner_iterations, outer_iterations,
                                                     forced_outer_per_pass, overlap_mode);
  if (overlapped) *overlapped = plan.overlapped ? 1 : 0;
  if (chunk) *chunk = plan.chunk;
  if (outer_per_pass) *outer_per_pass = plan.outer_per_pass;
  if (halo) *halo = plan.halo;
  if (max_planes) *max_planes = plan.max_planes;
  return 0;
}

size_t f3d_max_warp_level(size_t width, size_t height, size_t depth, float scale_factor)
{
  return OpticalFlowBase::GetMaxWarpLevel(width, height, depth, scale_factor);
}

int f3d_level_geometry(size_t width, size_t height, size_t depth, float scale_factor, int level, f3d_size4* size,
                       float* hx, float* hy, float* hz)
{
  if (!size || !hx || !hy || !hz) return 1;
  DataSize4 original = {width, height, depth, 0};
  PyramidLevel lv = OpticalFlowBase::GetLevel(original, scale_factor, level);
  *size = {lv.size.width, lv.size.height, lv.size.depth, 0};
  *hx = lv.hx;
  *hy = lv.hy;
  *hz = lv.hz;
  return 0;
}

int f3d_gaussian_taps(float sigma, float* taps, size_t capacity, size_t* radius)
{
  if (!taps || !radius) return 1;
  CudaOperationConvolution3D conv;
  conv.ComputeGaussianKernel(sigma, 3, 1.0);
  const size_t n = 2 * conv.KernelRadius() + 1;
  if (n > capacity || n > 51) return 1;
  std::memcpy(taps, conv.Kernel(), n * sizeof(float));
  *radius = conv.KernelRadius();
  return 0;
}

int f3d_raw_read_u8(const char* path, size_t width, size_t height, size_t depth, float* out)
{
  Data3D vol;
  if (!vol.ReadRAWFromFileU8(path, width, height, depth)) return 1;
  std::memcpy(out, vol.DataPtr(), width * height * depth * sizeof(float));
  return 0;
}

int f3d_raw_read_f32(const char* path, size_t width, size_t height, size_t depth, float* out)
{
  Data3D vol;
  if (!vol.ReadRAWFromFileF32(path, width, height, depth)) return 1;
  std::memcpy(out, vol.DataPtr(), width * height * depth * sizeof(float));
  return 0;
}

int f3d_raw_write_u8(const char* path, const float* in, size_t width, size_t height, size_t depth)
{
  Data3D vol(const_cast<float*>(in), width, height, depth);
  return vol.WriteRAWToFileU8(path) ? 0 : 1;
}

int f3d_raw_write_f32(const char* path, const float* in, size_t width, size_t height, size_t depth)
{
  Data3D vol(const_cast<float*>(in), width, height, depth);
  return vol.WriteRAWToFileF32(path) ? 0 : 1;
}

int f3d_vtk_write_flow(const char* path, const float* u, const float* v, const float* w, size_t width, size_t height,
                       size_t depth)
{
  Data3D fu(const_cast<float*>(u), width, height, depth), fv(const_cast<float*>(v), width, height, depth),
      fw(const_cast<float*>(w), width, height, depth);
  return Data3D::WriteFlowToFileVTK(path, fu, fv, fw) ? 0 : 1;
}

int f3d_synth_pair(size_t width, size_t height, size_t depth, float* frame_0, float* frame_1)
{
  if (!frame_0 || !frame_1 || width == 0 || height == 0 || depth == 0) return 1;
  f3d_synth::TranslatedGaussianPair(width, height, depth, frame_0, frame_1);
  return 0;
}

int f3d_synth_planes(size_t width, size_t height, size_t depth, size_t z_lo, size_t z_hi, float* frame_0, float* frame_1,
                      float* frame_0_max)
{
  if (!frame_0 || !frame_1 || !frame_0_max || z_lo > z_hi || z_hi > depth) return 1;
  *frame_0_max = f3d_synth::TranslatedGaussianPlanes(width, height, depth, z_lo, z_hi, frame_0, frame_1);
  return 0;
}

int f3d_slabflow_create(f3d_slabflow* flow, int n_ranks, const int* local_ranks, int n_local, int halo_capacity)
{
  if (!flow || !local_ranks || n_local < 1 || n_ranks < 1) return 1;
  *flow = new (std::nothrow) f3d_slabflow_s;
  if (!*flow) return 1;
  (*flow)->driver = new OpticalFlowSlab(n_ranks, std::vector<int>(local_ranks, local_ranks + n_local),
                                        halo_capacity > 0 ? halo_capacity : 16);
  return 0;
}

int f3d_slabflow_initialize(f3d_slabflow flow, size_t width, size_t height, size_t depth)
{
  if (!flow) return 1;
  if (f3d_init(-1) != 0) {
    std::fprintf(stderr, "f3d_slabflow_initialize: %s\n", f3d_last_error());
    return 1;
  }
  DataSize4 size = {width, height, depth, 0};
  return flow->driver->Initialize(size) ? 0 : 1;
}

namespace {
struct FullVolumes {
  Data3D f0, f1;
  FullVolumes(const float* a, const float* b, size_t w, size_t h, size_t d)
      : f0(const_cast<float*>(a), w, h, d), f1(const_cast<float*>(b), w, h, d) {}
};
}  // namespace

int f3d_slabflow_upload(f3d_slabflow flow, const float* frame_0, const float* frame_1)
{
  if (!flow || !frame_0 || !frame_1) return 1;
  const DataSize4 s = flow->driver->FullSize();
  FullVolumes v(frame_0, frame_1, s.width, s.height, s.depth);
  flow->driver->UploadFrames(v.f0, v.f1);
  return flow->driver->failed() ? 1 : 0;
}

int f3d_slabflow_compute_resident(f3d_slabflow flow, const f3d_flow_params* params, float* device_seconds)
{
  if (!flow || !params) return 1;
  f3d_flow_params p = *params;
  OperationParameters bag;
  FillBag(bag, p);
  const bool ok = flow->driver->ComputeResident(bag);
  if (device_seconds) *device_seconds = flow->driver->LastDeviceSeconds();
  return ok ? 0 : 1;
}

int f3d_slabflow_download(f3d_slabflow flow, float* u, float* v, float* w)
{
  if (!flow || !u || !v || !w) return 1;
  const DataSize4 s = flow->driver->FullSize();
  Data3D fu(u, s.width, s.height, s.depth), fv(v, s.width, s.height, s.depth), fw(w, s.width, s.height, s.depth);
  flow->driver->DownloadFlow(fu, fv, fw);
  return flow->driver->failed() ? 1 : 0;
}

int f3d_slabflow_compute(f3d_slabflow flow, const float* frame_0, const float* frame_1, const f3d_flow_params* params,
                         float* u, float* v, float* w)
{
  if (f3d_slabflow_upload(flow, frame_0, frame_1) != 0) return 1;
  if (f3d_slabflow_compute_resident(flow, params, nullptr) != 0) return 1;
  return f3d_slabflow_download(flow, u, v, w);
}

int f3d_slabflow_overlapped_iterations(f3d_slabflow flow, size_t* count)
{
  if (!flow || !count) return 1;
  *count = flow->driver->OverlappedIterations();
  return 0;
}

int f3d_slabflow_batched_exchanges(f3d_slabflow flow, size_t* count)
{
  if (!flow || !flow->driver || !count) return 1;
  *count = flow->driver->BatchedExchanges();
  return 0;
}

int f3d_slabflow_stage_exchanges(f3d_slabflow flow, size_t* count)
{
  if (!flow || !flow->driver || !count) return 1;
  *count = flow->driver->StageExchanges();
  return 0;
}

int f3d_slabflow_set_exchange_per_stage(f3d_slabflow flow, int per_stage)
{
  if (!flow || !flow->driver) return 1;
  flow->driver->SetExchangePerStage(per_stage != 0);
  return 0;
}

int f3d_slabflow_gathered_warps(f3d_slabflow flow, size_t* count)
{
  if (!flow || !flow->driver || !count) return 1;
  *count = flow->driver->GatheredWarps();
  return 0;
}

int f3d_slabflow_destroy(f3d_slabflow flow)
{
  delete flow;
  return 0;
}

int f3d_plan_owned(int depth, int rank, int n_ranks, int* lo, int* hi)
{
  if (!lo || !hi || n_ranks < 1 || rank < 0 || rank >= n_ranks || depth < 0) return 1;
  const PlaneRange r = OwnedPlanes(depth, rank, n_ranks);
  *lo = r.lo;
  *hi = r.hi;
  return 0;
}

int f3d_plan_exchange(int depth, int rank, int n_ranks, int need_lo, int need_hi, int* peer, int* send_lo, int* send_hi,
                      int* recv_lo, int* recv_hi, int capacity)
{
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return -1;
  const std::vector<HaloTransfer> plan = PlanHaloExchange(depth, rank, n_ranks, need_lo, need_hi);
  if (static_cast<int>(plan.size()) > capacity) return -1;
  for (size_t i = 0; i < plan.size(); ++i) {
    peer[i] = plan[i].peer;
    send_lo[i] = plan[i].send.lo;
    send_hi[i] = plan[i].send.hi;
    recv_lo[i] = plan[i].recv.lo;
    recv_hi[i] = plan[i].recv.hi;
  }
  return static_cast<int>(plan.size());
}

int f3d_plan_resample_source(int in_depth, int out_depth, int out_lo, int out_hi, int* lo, int* hi)
{
  if (!lo || !hi || in_depth < 1 || out_depth < 1) return 1;
  PlaneRange out;
  out.lo = out_lo;
  out.hi = out_hi;
  const PlaneRange r = ResampleSourcePlanes(in_depth, out_depth, out);
  *lo = r.lo;
  *hi = r.hi;
  return 0;
}

int f3d_host_shutdown(void)
{
  if (!f3d_is_initialized()) return 0;
  PiecemealReleaseArena();
  f3d_comm_destroy();
  return f3d_shutdown();
}

}  // extern "C"
