// Piecemeal operators: z-chunks of host volumes streamed through one device arena (see operations_p.h).  Parameter keys
// and the print-and-return error convention follow src/cuda_operations/partial_data/cuda_operation_*_p.cpp; the chunking
// itself is new.  All copies and launches of one Execute are queued on the library stream in order; Execute returns after
// the stream has drained, so the host volumes are complete.
#include "operations_p.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <utility>
#include <vector>

#include "common_utils.h"
#include "hip_utils.h"
#include "slab_plan.h"

namespace {

constexpr size_t kAlign = 256;
constexpr size_t kFieldSkew = 17 * kAlign;  // same stagger as f3d_alloc_pitched: fields must not share every stride

size_t RoundUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

// One device block shared by all piecemeal operators; grows to the largest request and is reused across calls.
struct Arena {
  DevicePtr base = 0;
  size_t bytes = 0;
} g_arena;
size_t g_reserved = 0;

// Copy queues and the events that order them against the kernels (overlapped solver schedule); created on first use.
struct Pipeline {
  f3d_queue up = nullptr, down = nullptr;
  f3d_event uploaded[2] = {nullptr, nullptr}, computed[2] = {nullptr, nullptr}, downloaded[2] = {nullptr, nullptr};
  f3d_event kept[2] = {nullptr, nullptr};   // the planes a chunk set took over from the other set have been copied (solver)
  bool ready = false;
} g_pipe;

void ReleaseAtExit();

bool PipelineReady()
{
  ReleaseAtExit();
  if (g_pipe.ready) return true;
  bool ok = !CheckDeviceError(f3d_queue_create(&g_pipe.up)) && !CheckDeviceError(f3d_queue_create(&g_pipe.down));
  for (int i = 0; i < 2 && ok; ++i)
    ok = !CheckDeviceError(f3d_event_create(&g_pipe.uploaded[i])) && !CheckDeviceError(f3d_event_create(&g_pipe.computed[i])) &&
         !CheckDeviceError(f3d_event_create(&g_pipe.downloaded[i])) && !CheckDeviceError(f3d_event_create(&g_pipe.kept[i]));
  g_pipe.ready = ok;
  return ok;
}

void PipelineRelease()
{
  for (int i = 0; i < 2; ++i) {
    f3d_event_destroy(g_pipe.uploaded[i]);
    f3d_event_destroy(g_pipe.computed[i]);
    f3d_event_destroy(g_pipe.downloaded[i]);
    f3d_event_destroy(g_pipe.kept[i]);
  }
  f3d_queue_destroy(g_pipe.up);
  f3d_queue_destroy(g_pipe.down);
  g_pipe = Pipeline();
}

void ArenaFree()
{
  if (g_arena.base) {
    f3d_stream_sync();
    CheckDeviceError(f3d_free(g_arena.base));
  }
  g_arena.base = 0;
  g_arena.bytes = 0;
}

// The arena, the copy queues and their events are released by f3d_host_shutdown() (the language bindings call it from
// their own exit hook, bin/flow3d at the end of main) or when the last piecemeal operator is destroyed.  The process exit
// handler below is only the last resort for callers that do neither, and it touches the device only while the library is
// still initialised: after f3d_shutdown() -- or once the HIP runtime may be tearing down -- it drops the handles unfreed.
void ReleaseAtExit()
{
  static const bool registered = (std::atexit([] {
                                    if (f3d_is_initialized()) PiecemealReleaseArena();
                                  }),
                                  true);
  (void)registered;
}

DevicePtr ArenaReserve(size_t bytes)
{
  ReleaseAtExit();
  if (bytes <= g_arena.bytes) return g_arena.base;
  ArenaFree();
  size_t pitch = 0;
  DevicePtr p = 0;
  if (CheckDeviceError(f3d_alloc_pitched(&p, &pitch, bytes, 1))) return 0;
  g_arena.base = p;
  g_arena.bytes = bytes;
  return p;
}

// Compact container of one level: rows of `pitch` bytes, `H` rows per plane.
struct ChunkBox {
  size_t W = 0, H = 0, pitch = 0, plane = 0;
  ChunkBox(size_t w, size_t h) : W(w), H(h), pitch(RoundUp(w * sizeof(float), kAlign)), plane(pitch * h) {}
  size_t FieldBytes(size_t planes) const { return RoundUp(planes * plane, kAlign) + kFieldSkew; }
  // planes that `buffers` buffers may hold in total
  size_t TotalPlanes(size_t budget, size_t buffers) const
  {
    const size_t overhead = buffers * (kFieldSkew + kAlign);
    return budget > overhead ? (budget - overhead) / plane : 0;
  }
};

// Carves buffers out of the arena, one after the other.
class Carver {
 public:
  explicit Carver(const ChunkBox& box) : box_(box) {}
  void Add(size_t planes) { sizes_.push_back(planes); }
  bool Commit()
  {
    size_t total = 0;
    for (size_t p : sizes_) total += box_.FieldBytes(p);
    const DevicePtr base = ArenaReserve(total);
    if (!base) return false;
    size_t at = 0;
    for (size_t p : sizes_) {
      ptrs_.push_back(base + at);
      at += box_.FieldBytes(p);
    }
    return true;
  }
  DevicePtr operator[](size_t i) const { return ptrs_[i]; }

 private:
  const ChunkBox& box_;
  std::vector<size_t> sizes_;
  std::vector<DevicePtr> ptrs_;
};

// Switches the device library to a chunk geometry and puts the caller's container back afterwards, so resident
// operators initialised earlier keep working.
class ContainerScope {
 public:
  ContainerScope(const ChunkBox& box, size_t planes)
  {
    had_ = f3d_get_container(&old_) == 0 && old_.pitch != 0;
    f3d_size4 c = {box.W, box.H, planes, box.pitch};
    ok_ = !CheckDeviceError(f3d_set_container(&c));
  }
  ~ContainerScope()
  {
    if (had_) f3d_set_container(&old_);
  }
  bool ok() const { return ok_; }

 private:
  f3d_size4 old_ = {0, 0, 0, 0};
  bool had_ = false, ok_ = false;
};

bool Fits(const Data3D& v, const DataSize4& s) { return s.width <= v.Width() && s.height <= v.Height() && s.depth <= v.Depth(); }

float* PlanePtr(Data3D& v, int z) { return v.DataPtr() + static_cast<size_t>(z) * v.Width() * v.Height(); }

bool Upload(DevicePtr dst, const ChunkBox& b, int dev_plane0, Data3D& v, size_t w, size_t h, int z0, int count, f3d_queue queue = nullptr)
{
  return !CheckDeviceError(
      f3d_copy_planes_h2d_on(queue, dst, b.pitch, b.H, dev_plane0, PlanePtr(v, z0), v.Width(), v.Height(), w, h, count));
}

bool Download(Data3D& v, size_t w, size_t h, int z0, int count, DevicePtr src, const ChunkBox& b, int dev_plane0, f3d_queue queue = nullptr)
{
  return !CheckDeviceError(
      f3d_copy_planes_d2h_on(queue, PlanePtr(v, z0), v.Width(), v.Height(), w, h, count, src, b.pitch, b.H, dev_plane0));
}

// pointer under which container plane (z - new_base) is the plane the buffer holds for z at (z - old_base)
DevicePtr Rebase(DevicePtr p, const ChunkBox& b, int old_base, int new_base)
{
  const int64_t shift = (static_cast<int64_t>(new_base) - old_base) * static_cast<int64_t>(b.plane);
  return static_cast<DevicePtr>(static_cast<int64_t>(p) + shift);
}

// largest |value| of a level's sub-box of a host volume (all cores); a NaN anywhere gives infinity: "no bound"
float HostAbsMax(Data3D& v, size_t w, size_t h, int d)
{
  float m = 0.f;
  const long long rows = static_cast<long long>(h) * d;
#pragma omp parallel for reduction(max : m) schedule(static)
  for (long long row = 0; row < rows; ++row) {
    const float* p = PlanePtr(v, static_cast<int>(row / static_cast<long long>(h))) + static_cast<size_t>(row % static_cast<long long>(h)) * v.Width();
    float rm = 0.f;
    for (size_t x = 0; x < w; ++x) {
      const float a = std::fabs(p[x]);
      rm = !(a <= rm) ? (a == a ? a : std::numeric_limits<float>::infinity()) : rm;
    }
    if (rm > m) m = rm;
  }
  return m;
}

void LowMemory(const char* name)
{
  std::printf("Operation '%s': Error. Low GPU memory. Data cannot be partitioned properly.\n", name);
}

}  // namespace

size_t PiecemealBudgetBytes()
{
  if (const char* e = std::getenv("F3D_P_BUDGET_MB")) {
    const double mb = std::atof(e);
    if (mb > 0) {
      const size_t bytes = static_cast<size_t>(mb * 1024.0 * 1024.0);
      return bytes > g_reserved ? bytes - g_reserved : 0;
    }
  }
  size_t free_b = 0, total_b = 0;
  if (CheckDeviceError(f3d_mem_info(&free_b, &total_b))) return 0;
  return static_cast<size_t>(0.85 * static_cast<double>(free_b + g_arena.bytes));
}

void PiecemealSetReservedBytes(size_t bytes) { g_reserved = bytes; }

size_t PiecemealMinResampleBytes(size_t width, size_t height)
{
  // one output plane reads at most ceil(delta) + 1 source planes, delta <= the depth shrink of the coarsest level; 16 covers
  // scale factors down to 0.95^40 with room to spare, and three buffers hold that span
  const ChunkBox box(width, height);
  return 4 * (kFieldSkew + kAlign) + (3 * 16 + 1) * box.plane;
}

void PiecemealReleaseArena()
{
  ArenaFree();
  PipelineRelease();
}

namespace {

// Best number of outer iterations per residency for one schedule.  Cost per owned voxel of one full solve: link bytes at
// ~50 GB/s (eight fields up, three down, per pass) and device bytes at ~5 TB/s (300 B per voxel and outer iteration on windows
// that average chunk + halo planes).  Serial: the three add up.  Overlapped (two chunk sets): the slower of link and device, plus
// the other once per level for filling the pipeline -- the LINK being up + down: measured per level, the two directions do not hide
// each other (LABBOOK, round 4).
// `total_planes` planes are there for `buffers` chunk buffers and, per plane of halo, `staging_per_halo` planes of staging (the
// overlapped schedule keeps the increments two neighbouring chunks share: 3 fields x 2 x halo planes).
// `fields_once`: of the fields_up, those whose halo planes do not travel again with the next chunk (the overlapped schedule hands the
// planes two neighbouring chunks share from one chunk set to the other on the device)
SolvePiecemealPlan PlanSchedule(size_t total_planes, size_t buffers, size_t staging_per_halo, int depth, int step, int outer_iterations,
                                int forced, bool overlapped, double fields_up = 8.0, double fields_once = 0.0)
{
  SolvePiecemealPlan plan;
  const size_t cap = static_cast<size_t>(std::numeric_limits<int>::max());
  plan.max_planes = static_cast<int>(std::min(total_planes / buffers, cap));
  plan.overlapped = overlapped;
  for (int n = 1; n <= outer_iterations; ++n) {
    if (forced > 0 && n != std::min(forced, outer_iterations)) continue;
    const int halo = n * step;
    const size_t staging = staging_per_halo * static_cast<size_t>(halo);
    if (staging >= total_planes) break;
    const int planes = static_cast<int>(std::min((total_planes - staging) / buffers, cap));
    const int chunk = planes - 2 * halo;
    if (chunk < 1) break;
    const double passes = std::ceil(static_cast<double>(outer_iterations) / n);
    const double wide = static_cast<double>(chunk + 2 * halo) / chunk, mid = static_cast<double>(chunk + halo) / chunk;
    const double up = passes * ((fields_up - fields_once) * wide + fields_once) * 4.0 / 50e9, down = passes * 3.0 * 4.0 / 50e9;
    const double device = outer_iterations * 300.0 * mid / 5e12;
    double cost = up + down + device;
    if (overlapped) {
      const double slowest = std::max(up + down, device);
      const double chunks = std::ceil(static_cast<double>(depth) / chunk) * passes;
      cost = slowest + (cost - slowest) / std::max(1.0, chunks);
    }
    if (plan.chunk == 0 || cost < plan.cost) {
      plan.cost = cost;
      plan.chunk = chunk;
      plan.outer_per_pass = n;
      plan.halo = halo;
      plan.max_planes = planes;
    }
  }
  return plan;
}

}  // namespace

SolvePiecemealPlan PlanSolvePiecemeal(size_t budget_bytes, size_t width, size_t height, int depth, int inner_iterations,
                                      int outer_iterations, int forced_outer_per_pass, int overlap_mode, int fields, int constant_fields)
{
  SolvePiecemealPlan plan;
  const ChunkBox box(width, height);
  const size_t cap = static_cast<size_t>(std::numeric_limits<int>::max());
  if (constant_fields > 0) {
    // whole-level fields beside the chunk sets: what is left of the budget is planned as before, with fewer fields per set and fewer
    // fields up per residency; the one upload of the constants is added to the cost
    plan.constants_on_device = true;
    const size_t held = static_cast<size_t>(constant_fields) * box.FieldBytes(static_cast<size_t>(std::max(depth, 0)));
    if (depth <= 0 || outer_iterations <= 0 || held >= budget_bytes) return plan;
    const size_t nf = static_cast<size_t>(fields), left = budget_bytes - held;
    const size_t nf2 = nf + static_cast<size_t>(8 - constant_fields);   // two sets: the fields that travel twice, the compute-only ones once
    const int serial_planes = static_cast<int>(std::min(box.TotalPlanes(left, nf) / nf, cap));
    if (serial_planes >= depth) return plan;   // (a level that fits whole is one residency anyway: nothing to keep)
    const int step = inner_iterations + 1;
    const double up = 8.0 - constant_fields;
    SolvePiecemealPlan serial = PlanSchedule(box.TotalPlanes(left, nf), nf, 0, depth, step, outer_iterations, forced_outer_per_pass, false, up);
    SolvePiecemealPlan overlapped =
        PlanSchedule(box.TotalPlanes(left, nf2 + 3), nf2, 6, depth, step, outer_iterations, forced_outer_per_pass, true, up, up);
    SolvePiecemealPlan best = overlap_mode == 0 ? serial : overlap_mode == 1 ? overlapped
                              : (overlapped.chunk > 0 && (serial.chunk == 0 || overlapped.cost < serial.cost)) ? overlapped : serial;
    best.constants_on_device = true;
    best.cost += constant_fields * 4.0 / 50e9;
    return best;
  }
  // 13 fields per chunk set (eight inputs, phi, ksi, three outputs), 15 with the second weight pair of the fused last sweep
  // Two chunk sets hold the fields that travel (eight inputs, of which three come back) twice and the compute-only ones (phi, ksi, the
  // sweeps' ping-pong partners) ONCE: the kernels of the two sets run one after the other on one stream, only the copies overlap.
  // The overlapped schedule also keeps, in three staging buffers of 2 x halo planes, the increments two neighbouring chunks share as
  // they arrived: with the frames' and the flow's shared planes handed on from set to set, every plane of every field then travels
  // once per pass.
  const size_t nf = static_cast<size_t>(fields), nf2 = nf + 8;
  const int serial_planes = static_cast<int>(std::min(box.TotalPlanes(budget_bytes, nf) / nf, cap));
  plan.max_planes = serial_planes;
  if (depth <= 0 || outer_iterations <= 0) return plan;
  if (serial_planes >= depth) {  // the level fits: one residency for the whole solve, no halo
    plan.chunk = depth;
    plan.outer_per_pass = outer_iterations;
    plan.halo = 0;
    plan.cost = (8.0 + 3.0) * 4.0 / 50e9 + outer_iterations * 300.0 / 5e12;
    return plan;
  }
  const int step = inner_iterations + 1;
  const SolvePiecemealPlan serial =
      PlanSchedule(box.TotalPlanes(budget_bytes, nf), nf, 0, depth, step, outer_iterations, forced_outer_per_pass, false);
  const SolvePiecemealPlan overlapped =
      PlanSchedule(box.TotalPlanes(budget_bytes, nf2 + 3), nf2, 6, depth, step, outer_iterations, forced_outer_per_pass, true, 8.0, 8.0);
  if (overlap_mode == 0) return serial;
  if (overlap_mode == 1) return overlapped;
  return (overlapped.chunk > 0 && (serial.chunk == 0 || overlapped.cost < serial.cost)) ? overlapped : serial;
}

namespace {
int g_live_operators = 0;
}

bool CudaOperationPiecemealBase::Initialize(const OperationParameters*)
{
  size_t free_b = 0, total_b = 0;
  const bool was = initialized_;
  initialized_ = !CheckDeviceError(f3d_mem_info(&free_b, &total_b));  // needs a live device context, nothing else
  if (initialized_ && !was) ++g_live_operators;
  if (!initialized_ && was) --g_live_operators;
  return initialized_;
}

void CudaOperationPiecemealBase::Destroy()
{
  if (initialized_ && --g_live_operators == 0) PiecemealReleaseArena();
  initialized_ = false;
}

// ---- add (cuda_operation_add_p.cpp:52-214) ----------------------------------------------------------------------

void CudaOperationAddP::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  Data3D *p_operand_0, *p_operand_1;
  DataSize4 data_size;
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_operand_0, "operand_0");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_operand_1, "operand_1");
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  if (!Fits(*p_operand_0, data_size) || !Fits(*p_operand_1, data_size)) {
    std::printf("Error: Operation '%s'. Wrong dimensions.\n", GetName());
    return;
  }
  const size_t W = data_size.width, H = data_size.height;
  const int D = static_cast<int>(data_size.depth);
  if (W == 0 || H == 0 || D == 0) return;
  const ChunkBox box(W, H);
  const int chunk = static_cast<int>(std::min<size_t>(box.TotalPlanes(PiecemealBudgetBytes(), 2) / 2, D));
  if (chunk < 1) return LowMemory(GetName());
  Carver buf(box);
  buf.Add(chunk);
  buf.Add(chunk);
  if (!buf.Commit()) return;
  ContainerScope scope(box, chunk);
  if (!scope.ok()) return;
  for (int z0 = 0; z0 < D; z0 += chunk) {
    const int z1 = std::min(D, z0 + chunk);
    const f3d_slab slab = {z0, z0, z1};
    if (!Upload(buf[0], box, 0, *p_operand_0, W, H, z0, z1 - z0) || !Upload(buf[1], box, 0, *p_operand_1, W, H, z0, z1 - z0)) return;
    if (CheckDeviceError(f3d_add(buf[0], buf[1], W, H, D, &slab))) return;
    if (!Download(*p_operand_0, W, H, z0, z1 - z0, buf[0], box, 0)) return;
  }
  CheckDeviceError(f3d_stream_sync());
}

// ---- statistics (cuda_operation_stat_p.cpp:44-107) ----------------------------------------------------------------

void CudaOperationStatP::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  Data3D *p_flow_u, *p_flow_v, *p_flow_w;
  DataSize4 data_size;
  Stat3* p_stat;
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_flow_u, "flow_u");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_flow_v, "flow_v");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_flow_w, "flow_w");
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  GET_PARAM_PTR_OR_RETURN(params, Stat3, p_stat, "stat");
  if (!silent) std::printf("Compute statistics...\n");
  Data3D* vols[3] = {p_flow_u, p_flow_v, p_flow_w};
  for (Data3D* v : vols)
    if (!Fits(*v, data_size)) {
      std::printf("Error: Operation '%s'. Wrong dimensions.\n", GetName());
      return;
    }
  const size_t W = data_size.width, H = data_size.height;
  const int D = static_cast<int>(data_size.depth);
  if (W == 0 || H == 0 || D == 0) return;
  const ChunkBox box(W, H);
  const int chunk = static_cast<int>(std::min<size_t>(box.TotalPlanes(PiecemealBudgetBytes(), 3) / 3, D));
  if (chunk < 1) return LowMemory(GetName());
  Carver buf(box);
  for (int i = 0; i < 3; ++i) buf.Add(chunk);
  if (!buf.Commit()) return;
  ContainerScope scope(box, chunk);
  if (!scope.ok()) return;
  float mn = std::numeric_limits<float>::max(), mx = 0.f;
  double sum = 0.0;
  for (int z0 = 0; z0 < D; z0 += chunk) {
    const int z1 = std::min(D, z0 + chunk);
    const f3d_slab slab = {z0, z0, z1};
    for (int i = 0; i < 3; ++i)
      if (!Upload(buf[i], box, 0, *vols[i], W, H, z0, z1 - z0)) return;
    float cmn = 0.f, cmx = 0.f;
    double csum = 0.0;
    if (CheckDeviceError(f3d_flow_stats(buf[0], buf[1], buf[2], W, H, D, &slab, &cmn, &cmx, &csum))) return;
    mn = std::fmin(mn, cmn);
    mx = std::fmax(mx, cmx);
    sum += csum;
  }
  p_stat->min = mn;
  p_stat->max = mx;
  p_stat->avg = static_cast<float>(sum / (static_cast<double>(W) * static_cast<double>(H) * static_cast<double>(D)));
  if (!silent) std::printf("Min: %8.4f Max: %8.4f Avg: %8.4f\n", p_stat->min, p_stat->max, p_stat->avg);
}

// ---- resample (cuda_operation_resample_p.cpp:63-116: X input->output, then Y and Z in place) -----------------------

void CudaOperationResampleP::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  DataSize4 data_size, resample_size;
  Data3D *input_ptr, *output_ptr;
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  GET_PARAM_OR_RETURN(params, DataSize4, resample_size, "resample_size");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, input_ptr, "input");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, output_ptr, "output");
  Data3D& input = *input_ptr;
  Data3D& output = *output_ptr;
  // the reference's three checks (cuda_operation_resample_p.cpp:79-98)
  const bool downsample_check = output.Width() >= input.Width() && output.Height() >= input.Height() && output.Depth() >= input.Depth();
  if (!Fits(input, data_size) || !Fits(output, resample_size) || !downsample_check) {
    std::printf("Error: Operation '%s'. Wrong dimensions.\n", GetName());
    return;
  }
  Run(&input, 0, 0, 0, data_size, resample_size, output_ptr, 0, 0, 0);
}

bool CudaOperationResampleP::ExecuteToDevice(Data3D& input, const DataSize4& data_size, const DataSize4& resample_size, DevicePtr dst,
                                             size_t dst_pitch, size_t dst_rows)
{
  if (!IsInitialized()) return false;
  if (!Fits(input, data_size) || !dst || resample_size.height > dst_rows || resample_size.width * sizeof(float) > dst_pitch) {
    std::printf("Error: Operation '%s'. Wrong dimensions.\n", GetName());
    return false;
  }
  return Run(&input, 0, 0, 0, data_size, resample_size, nullptr, dst, dst_pitch, dst_rows);
}

bool CudaOperationResampleP::ExecuteDeviceToDevice(DevicePtr src, size_t src_pitch, size_t src_rows, const DataSize4& data_size,
                                                   const DataSize4& resample_size, DevicePtr dst, size_t dst_pitch, size_t dst_rows)
{
  if (!IsInitialized()) return false;
  if (!src || !dst || data_size.height > src_rows || data_size.width * sizeof(float) > src_pitch || resample_size.height > dst_rows ||
      resample_size.width * sizeof(float) > dst_pitch) {
    std::printf("Error: Operation '%s'. Wrong dimensions.\n", GetName());
    return false;
  }
  return Run(nullptr, src, src_pitch, src_rows, data_size, resample_size, nullptr, dst, dst_pitch, dst_rows);
}

bool CudaOperationResampleP::Run(Data3D* input, DevicePtr src_dev, size_t src_pitch, size_t src_rows, const DataSize4& data_size,
                                 const DataSize4& resample_size, Data3D* output, DevicePtr dst, size_t dst_pitch, size_t dst_rows)
{
  const size_t Wi = data_size.width, Hi = data_size.height, Wo = resample_size.width, Ho = resample_size.height;
  const int Di = static_cast<int>(data_size.depth), Do = static_cast<int>(resample_size.depth);
  if (Wi == 0 || Hi == 0 || Di == 0 || Wo == 0 || Ho == 0 || Do == 0) return true;

  // Buffers per chunk of `c` output planes that read `s` source planes: source, x pass, x+y pass (s planes each), result
  // (c planes), all in one container geometry that holds both boxes.
  // Host volume to host volume on page-locked memory (the driver's frames and flow components): TWO such buffer sets and the solver's
  // copy queues -- the upload of chunk k+1 runs beside the kernels and the download of chunk k.  A level's resampling is link time
  // (the source planes up, the result down, a few kernel passes in between); in order on one stream the two directions add up.
  // F3D_P_OVERLAP=0 keeps the one-stream order.
  const ChunkBox box(std::max(Wi, Wo), std::max(Hi, Ho));
  bool overlapped = input && output;
  if (overlapped) {
    const char* e = std::getenv("F3D_P_OVERLAP");
    int a = 0, b = 0;
    overlapped = !(e && std::atoi(e) == 0) && f3d_host_is_pinned(input->DataPtr(), &a) == 0 && a &&
                 f3d_host_is_pinned(output->DataPtr(), &b) == 0 && b && PipelineReady();
  }
  auto source_of = [&](int z0, int z1) { return ResampleSourcePlanes(Di, Do, PlaneRange{z0, z1}); };
  auto chunk_for = [&](size_t total) {
    auto fits = [&](int c) {
      size_t worst = 0;
      for (int z0 = 0; z0 < Do; z0 += c) worst = std::max<size_t>(worst, source_of(z0, std::min(Do, z0 + c)).size());
      return 3 * worst + static_cast<size_t>(c) <= total;
    };
    int chunk = Do;
    if (!fits(chunk)) {  // largest chunk that fits, by bisection (the source span grows with the chunk)
      int lo = 0, hi = Do;
      while (hi - lo > 1) {
        const int mid = lo + (hi - lo) / 2;
        (fits(mid) ? lo : hi) = mid;
      }
      chunk = lo;
    }
    return chunk;
  };
  int chunk = 0;
  if (overlapped) {
    // at least eight chunks where the budget would take the level in fewer: the first upload and the last download are the part that
    // nothing runs beside.  (Two sets that do not fit, or a level of a few planes, fall back to one set.)
    chunk = std::min(chunk_for(box.TotalPlanes(PiecemealBudgetBytes(), 8) / 2), std::max(1, (Do + 7) / 8));
    if (chunk < 1 || (Do + chunk - 1) / chunk < 3) overlapped = false;
  }
  if (!overlapped) chunk = chunk_for(box.TotalPlanes(PiecemealBudgetBytes(), 4));
  if (chunk < 1) {
    LowMemory(GetName());
    return false;
  }
  const int n_sets = overlapped ? 2 : 1;
  size_t span = 0;
  for (int z0 = 0; z0 < Do; z0 += chunk) span = std::max<size_t>(span, source_of(z0, std::min(Do, z0 + chunk)).size());
  Carver carve(box);
  for (int s = 0; s < n_sets; ++s) {
    for (int i = 0; i < 3; ++i) carve.Add(span);
    carve.Add(chunk);
  }
  if (!carve.Commit()) return false;
  ContainerScope scope(box, std::max<size_t>(span, chunk));
  if (!scope.ok()) return false;
  const f3d_queue q_up = overlapped ? g_pipe.up : nullptr, q_down = overlapped ? g_pipe.down : nullptr;
  bool set_used[2] = {false, false};

  // In place (input == output) a chunk's result lands on host planes [z0, z1) of the volume it is read from.  Going up
  // in z is safe when the depth shrinks or stays (later chunks read planes >= floor(z1 * delta) >= z1), going down when
  // it grows (earlier chunks read planes < ceil(z0 * delta) <= z0) -- with the copies of neighbouring chunks in flight at
  // the same time as well: the planes one chunk writes and the planes a later chunk reads never meet.
  const int n_chunks = (Do + chunk - 1) / chunk;
  const bool descending = Do > Di;
  for (int k = 0; k < n_chunks; ++k) {
    const int z0 = (descending ? n_chunks - 1 - k : k) * chunk, z1 = std::min(Do, z0 + chunk);
    const PlaneRange src = source_of(z0, z1);
    const f3d_slab in_slab = {src.lo, src.lo, src.hi}, out_slab = {z0, z0, z1};
    const int set = overlapped ? (k & 1) : 0;
    const DevicePtr b_src = carve[4 * set], b_x = carve[4 * set + 1], b_xy = carve[4 * set + 2], b_out = carve[4 * set + 3];
    if (overlapped && set_used[set]) {
      // the set's previous chunk must have left: its result is still being read by the download queue
      if (CheckDeviceError(f3d_queue_wait_event(q_up, g_pipe.downloaded[set]))) return false;
      if (CheckDeviceError(f3d_queue_wait_event(nullptr, g_pipe.downloaded[set]))) return false;
    }
    if (input) {
      if (!Upload(b_src, box, 0, *input, Wi, Hi, src.lo, src.size(), q_up)) return false;
    } else if (CheckDeviceError(f3d_copy_rect_d2d(b_src, box.pitch, box.H, 0, src_dev, src_pitch, src_rows, src.lo, Wi, Hi, src.size()))) {
      return false;
    }
    if (overlapped) {
      if (CheckDeviceError(f3d_event_record_on(g_pipe.uploaded[set], q_up))) return false;
      if (CheckDeviceError(f3d_queue_wait_event(nullptr, g_pipe.uploaded[set]))) return false;
    }
    if (CheckDeviceError(f3d_resample_x(b_src, b_x, Wo, Hi, Di, Wi, &in_slab))) return false;
    if (CheckDeviceError(f3d_resample_y(b_x, b_xy, Wo, Ho, Di, Hi, &in_slab))) return false;
    if (CheckDeviceError(f3d_resample_z(b_xy, b_out, Wo, Ho, Do, Di, &in_slab, &out_slab))) return false;
    if (overlapped) {
      if (CheckDeviceError(f3d_event_record_on(g_pipe.computed[set], nullptr))) return false;
      if (CheckDeviceError(f3d_queue_wait_event(q_down, g_pipe.computed[set]))) return false;
    }
    if (output) {
      if (!Download(*output, Wo, Ho, z0, z1 - z0, b_out, box, 0, q_down)) return false;
    } else if (CheckDeviceError(f3d_copy_rect_d2d(dst, dst_pitch, dst_rows, z0, b_out, box.pitch, box.H, 0, Wo, Ho, z1 - z0))) {
      return false;
    }
    if (overlapped) {
      if (CheckDeviceError(f3d_event_record_on(g_pipe.downloaded[set], q_down))) return false;
      set_used[set] = true;
    }
  }
  if (overlapped && CheckDeviceError(f3d_queue_sync(q_down))) return false;
  return !CheckDeviceError(f3d_stream_sync());
}

// ---- registration (cuda_operation_register_p.cpp:54-139: the reference warps on the CPU) ---------------------------

void CudaOperationRegistrationP::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  Data3D *p_frame_0, *p_frame_1, *p_flow_u, *p_flow_v, *p_flow_w, *p_temp;
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_frame_0, "frame_0");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_frame_1, "frame_1");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_flow_u, "flow_u");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_flow_v, "flow_v");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_flow_w, "flow_w");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_temp, "temp");
  float hx, hy, hz;
  DataSize4 data_size;
  size_t max_mag;
  GET_PARAM_OR_RETURN(params, float, hx, "hx");
  GET_PARAM_OR_RETURN(params, float, hy, "hy");
  GET_PARAM_OR_RETURN(params, float, hz, "hz");
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  GET_PARAM_OR_RETURN(params, size_t, max_mag, "max_mag");
  (void)max_mag;  // the halo comes from the flow itself, chunk by chunk
  Data3D* in[4] = {p_frame_0, p_flow_u, p_flow_v, p_flow_w};
  for (Data3D* v : {p_frame_0, p_frame_1, p_flow_u, p_flow_v, p_flow_w, p_temp})
    if (!Fits(*v, data_size)) {
      std::printf("Error: Operation '%s'. Wrong dimensions.\n", GetName());
      return;
    }
  if (p_frame_1 == p_temp) {
    std::printf("Operation '%s': Error. Input buffer cannot serve as output buffer.", GetName());
    return;
  }
  const size_t W = data_size.width, H = data_size.height;
  const int D = static_cast<int>(data_size.depth);
  if (W == 0 || H == 0 || D == 0) return;

  // frame_0, u, v, w and the result hold `chunk` planes; frame_1 gets the rest for chunk + 2 * reach planes
  const ChunkBox box(W, H);
  const size_t total = box.TotalPlanes(PiecemealBudgetBytes(), 6);
  int chunk, f1_cap;
  if (total >= 6 * static_cast<size_t>(D)) {
    chunk = f1_cap = D;
  } else {
    chunk = static_cast<int>(std::min<size_t>(total / 8, D));
    f1_cap = chunk < 1 ? 0 : static_cast<int>(std::min<size_t>(total - 5 * static_cast<size_t>(chunk), D));
  }
  if (chunk < 1 || f1_cap < 1) return LowMemory(GetName());
  Carver buf(box);
  for (int i = 0; i < 5; ++i) buf.Add(chunk);
  buf.Add(f1_cap);
  if (!buf.Commit()) return;
  ContainerScope scope(box, std::max(chunk, f1_cap));
  if (!scope.ok()) return;
  const DevicePtr dev_out = buf[4], dev_f1 = buf[5];

  for (int z0 = 0; z0 < D; z0 += chunk) {
    const int z1 = std::min(D, z0 + chunk);
    for (int i = 0; i < 4; ++i)
      if (!Upload(buf[i], box, 0, *in[i], W, H, z0, z1 - z0)) return;
    const f3d_slab own = {z0, z0, z1};
    float max_w = 0.f;
    if (CheckDeviceError(f3d_abs_max(buf[3], W, H, D, &own, &max_w))) return;
    // frame_1 planes a voxel of plane z can read: z - reach .. z + reach (registration_3d.cu:46-80)
    const float planes = std::ceil(max_w / hz);
    const int reach = (planes == planes && planes < static_cast<float>(D)) ? static_cast<int>(planes) + 1 : D;
    int sub = z1 - z0;
    if (std::min(D, z1 + reach) - std::max(0, z0 - reach) > f1_cap) sub = f1_cap - 2 * reach;
    if (sub < 1) {
      std::printf("Operation '%s': Error. Low GPU memory: a flow of %.1f planes in z needs %d planes of frame_1, %d fit.\n", GetName(),
                  max_w / hz, 2 * reach + 1, f1_cap);
      return;
    }
    for (int s0 = z0; s0 < z1; s0 += sub) {
      const int s1 = std::min(z1, s0 + sub);
      const int lo = std::max(0, s0 - reach), hi = std::min(D, s1 + reach);
      if (!Upload(dev_f1, box, 0, *p_frame_1, W, H, lo, hi - lo)) return;
      // one z_base (frame_1's) for every operand: the chunk buffers are addressed as if they started at plane `lo`
      const f3d_slab win = {lo, s0, s1};
      if (CheckDeviceError(f3d_warp(Rebase(buf[0], box, z0, lo), dev_f1, Rebase(buf[1], box, z0, lo), Rebase(buf[2], box, z0, lo),
                                    Rebase(buf[3], box, z0, lo), W, H, D, hx, hy, hz, Rebase(dev_out, box, z0, lo), &win)))
        return;
    }
    if (!Download(*p_temp, W, H, z0, z1 - z0, dev_out, box, 0)) return;
  }
  if (CheckDeviceError(f3d_stream_sync())) return;
  p_frame_1->Swap(*p_temp);
}

// ---- solve (cuda_operation_solve_p.cpp:60-207) --------------------------------------------------------------------

void CudaOperationSolveP::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  Data3D *p_frame_0, *p_frame_1, *p_flow_u, *p_flow_v, *p_flow_w, *p_flow_du, *p_flow_dv, *p_flow_dw, *p_temp_du, *p_temp_dv, *p_temp_dw;
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_frame_0, "frame_0");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_frame_1, "frame_1");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_flow_u, "flow_u");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_flow_v, "flow_v");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_flow_w, "flow_w");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_flow_du, "flow_du");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_flow_dv, "flow_dv");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_flow_dw, "flow_dw");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_temp_du, "temp_du");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_temp_dv, "temp_dv");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_temp_dw, "temp_dw");
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

  Data3D* fixed[5] = {p_frame_0, p_frame_1, p_flow_u, p_flow_v, p_flow_w};
  Data3D* inc[3] = {p_flow_du, p_flow_dv, p_flow_dw};
  Data3D* next[3] = {p_temp_du, p_temp_dv, p_temp_dw};
  for (Data3D* v : {p_frame_0, p_frame_1, p_flow_u, p_flow_v, p_flow_w, p_flow_du, p_flow_dv, p_flow_dw, p_temp_du, p_temp_dv, p_temp_dw})
    if (!Fits(*v, data_size)) {
      std::printf("Error: Operation '%s'. Wrong dimensions.\n", GetName());
      return;
    }
  const size_t W = data_size.width, H = data_size.height;
  const int D = static_cast<int>(data_size.depth);
  const int K = static_cast<int>(inner_iterations_count), outer = static_cast<int>(outer_iterations_count);
  last_plan_ = SolvePiecemealPlan();
  last_passes_ = 0;
  last_added_ = false;
  last_registered_ = false;
  if (register_frame_1 && registered_frame_1 && (!Fits(*registered_frame_1, data_size) || registered_frame_1 == p_frame_1)) {
    std::printf("Error: Operation '%s'. Wrong volume for the registered frame.\n", GetName());
    return;
  }
  if (W == 0 || H == 0 || D == 0) return;

  // The increments start at zero (cuda_operation_solve_p.cpp:152-154); with no sweep to run that is also the result.
  if (outer == 0 || K == 0) {
    for (Data3D* v : inc) v->ZeroData();
    return;
  }

  int forced = 0, overlap_mode = -1;
  if (const char* e = std::getenv("F3D_P_OUTER_PER_PASS")) forced = std::atoi(e);
  if (const char* e = std::getenv("F3D_P_OVERLAP")) overlap_mode = std::atoi(e);
  // Copies beside the kernels only from page-locked volumes: from pageable memory the runtime stages every copy through
  // its own buffers and blocks the caller, which serialises the three queues anyway.
  bool all_pinned = true;
  for (Data3D* v : {p_frame_0, p_frame_1, p_flow_u, p_flow_v, p_flow_w, p_flow_du, p_flow_dv, p_flow_dw, p_temp_du, p_temp_dv, p_temp_dw}) {
    int yes = 0;
    if (f3d_host_is_pinned(v->DataPtr(), &yes) != 0 || !yes) all_pinned = false;
  }
  if (register_frame_1 && registered_frame_1) {
    int yes = 0;
    if (f3d_host_is_pinned(registered_frame_1->DataPtr(), &yes) != 0 || !yes) all_pinned = false;
  }
  if (!all_pinned) overlap_mode = 0;
  // The last sweep of an outer iteration and the weights of the next one in ONE launch wherever another outer iteration follows
  // inside the residency (f3d_solve_sweep_phi_ksi_edges, as the resident operator and the z-slab driver do): it needs a second
  // weight pair in every chunk set, 15 fields instead of 13.  Measured (1024^3 on a 16 GB budget, tools/r3_job7.sh): where a level
  // goes through in CHUNKS the solver is bound by the link, not by the device, and 15 fields mean fewer planes per chunk, deeper
  // relative halos and more residencies -- 32.8 s of solver against 30.2 s with the 13.  So the fused launch is taken only where the
  // level fits the budget with all 15 fields (one residency, no halo: the device is what is left to save on); F3D_P_FUSED=1
  // forces it for chunked levels too (the tests run both).
  bool fuse_weights = FusedSweepsEnabled() && FusedPhiKsiEnabled() && K % 2 == 1 && outer > 1;
  const bool fuse_when_chunked = std::getenv("F3D_P_FUSED") && std::atoi(std::getenv("F3D_P_FUSED")) == 1;
  auto make_plan = [&](int fields) {
    SolvePiecemealPlan p = PlanSolvePiecemeal(PiecemealBudgetBytes(), W, H, D, K, outer, forced, overlap_mode, fields);
    if (p.overlapped && !PipelineReady()) p = PlanSolvePiecemeal(PiecemealBudgetBytes(), W, H, D, K, outer, forced, 0, fields);
    return p;
  };
  SolvePiecemealPlan plan = make_plan(fuse_weights ? 15 : 13);
  if (fuse_weights && (plan.chunk < 1 || plan.outer_per_pass < 2 || (plan.halo > 0 && !fuse_when_chunked))) {
    fuse_weights = false;
    plan = make_plan(13);
  }
  // A level that goes through in chunks sends eight fields up per residency, and five of them -- the two frames, u, v, w -- are the
  // same every time.  Where the budget holds those five for the WHOLE level beside (smaller) chunk sets of the other eight, they go up
  // once, in the first pass, and a residency moves three fields up and three down; the model prices both layouts (deeper relative
  // halos and more residencies against five fields less per residency).  F3D_P_CONSTANTS=0 keeps every field in the chunk sets.
  const char* kc_env = std::getenv("F3D_P_CONSTANTS");
  if (plan.chunk >= 1 && plan.halo > 0 && !(kc_env && kc_env[0] == '0')) {
    const int set_fields = fuse_weights ? 10 : 8;
    SolvePiecemealPlan kept = PlanSolvePiecemeal(PiecemealBudgetBytes(), W, H, D, K, outer, forced, overlap_mode, set_fields, 5);
    if (kept.overlapped && !PipelineReady()) kept = PlanSolvePiecemeal(PiecemealBudgetBytes(), W, H, D, K, outer, forced, 0, set_fields, 5);
    const bool force = kc_env && kc_env[0] == '1';
    // Which layout: by the fields each moves over the link in BOTH directions, added up.  The schedule model prices two copy queues
    // as the slower of the two directions; measured per level (1024^3 on 16 GB, profiles/r04_piecemeal_per_level.txt) a layout that
    // loads both directions evenly gets no such discount -- 3 up + 3 down per residency beat 8 up + 3 down by what the SUM says (levels of
    // 590-650 planes: -10 ... -34 %), and lost where the sum said so although the slower direction alone promised -16 %.
    auto link_fields = [&](const SolvePiecemealPlan& p, double fields_up, double once_per_pass, double once) {
      const double passes = std::ceil(static_cast<double>(outer) / p.outer_per_pass);
      return passes * ((fields_up - once_per_pass) * static_cast<double>(p.chunk + 2 * p.halo) / p.chunk + once_per_pass + 3.0) + once;
    };
    // (the overlapped schedule hands the constants' shared planes from chunk set to chunk set: they travel once per pass there)
    const bool pays = kept.chunk >= 1 && kept.halo > 0 && link_fields(kept, 3.0, kept.overlapped ? 3.0 : 0.0, 5.0) <
                                                            0.95 * link_fields(plan, 8.0, plan.overlapped ? 8.0 : 0.0, 0.0);
    if (kept.chunk >= 1 && kept.halo > 0 && (pays || force) && !(fuse_weights && kept.outer_per_pass < 2)) plan = kept;
  }
  const bool constants = plan.constants_on_device && plan.chunk >= 1;
  last_plan_ = plan;
  last_fused_weights_ = fuse_weights;
  if (plan.chunk < 1) return LowMemory(GetName());
  const int chunk = plan.chunk, halo = plan.halo, planes = std::min(D, chunk + 2 * halo);
  const int n_sets = plan.overlapped ? 2 : 1;

  // Registration inside the first residency (register_frame_1): frame-1 planes a voxel of plane z reads are z - warp_reach ..
  // z + warp_reach (registration_3d.cu:46-80), bounded for the whole level by the largest |w| (area resampling and the median do not
  // raise it, but the volume is right here).  The unregistered planes of a residency's window go into the buffers of temp_du / dv /
  // dw -- nothing writes them before the first sweep -- in up to three pieces of `planes` planes each, the piece's own reach included;
  // a reach that leaves less than a third of a buffer for the planes themselves cannot be served that way.
  const bool registering = register_frame_1;
  int warp_reach = 0;
  if (registering) {
    const float deep = std::ceil(HostAbsMax(*p_flow_w, W, H, D) / hz);
    warp_reach = (deep == deep && deep < static_cast<float>(D)) ? static_cast<int>(deep) + 1 : D;
    if (planes < D && 3 * (planes - 2 * warp_reach) < planes) return;   // (planes == D: one piece holds the whole frame)
  }
  Data3D* const registered_to = registering ? (registered_frame_1 ? registered_frame_1 : p_flow_du) : nullptr;

  enum { F0, F1, FU, FV, FW, DU, DV, DW, PHI, KSI, TDU, TDV, TDW, PHI2, KSI2, kAllFields };
  const int kFields = fuse_weights ? 15 : 13;
  const ChunkBox box(W, H);
  Carver carve(box);
  const int kConstants = constants ? 5 : 0;   // F0 .. FW hold the whole level (plane z at plane z), once for all chunk sets
  // what a chunk set owns: the fields that travel, F0 .. DW (DU .. DW with the constants held); phi, ksi and the sweeps' ping-pong
  // partners are compute-only and carved once, for both sets (kernels of the two sets follow each other on the library stream)
  const int kOwn = 8 - kConstants, kShared = kFields - 8;
  for (int i = 0; i < kConstants; ++i) carve.Add(static_cast<size_t>(D));
  for (int i = 0; i < n_sets * kOwn + kShared; ++i) carve.Add(planes);
  const bool want_staging = plan.overlapped && halo > 0;   // (three buffers of 2 x halo planes: the planner has left room for them)
  for (int i = 0; i < (want_staging ? 3 : 0); ++i) carve.Add(static_cast<size_t>(2 * halo));
  if (!carve.Commit()) return;
  ContainerScope scope(box, planes);
  if (!scope.ok()) return;
  DevicePtr sets[2][kAllFields] = {};
  for (int s = 0; s < n_sets; ++s) {
    for (int i = kConstants; i < 8; ++i) sets[s][i] = carve[kConstants + s * kOwn + (i - kConstants)];
    for (int i = 8; i < kFields; ++i) sets[s][i] = carve[kConstants + n_sets * kOwn + (i - 8)];
  }
  DevicePtr staging_bufs[3] = {0, 0, 0};
  for (int i = 0; i < (want_staging ? 3 : 0); ++i) staging_bufs[i] = carve[kConstants + n_sets * kOwn + kShared + i];
  int constants_up_to = 0, registered_up_to = 0;   // planes of the whole-level fields that have arrived / been registered so far
  // Serial: copies and kernels in order on the library stream.  Overlapped: uploads on one queue, downloads on another,
  // kernels on the library stream; a chunk set is reused once the download of its previous chunk has finished.
  const f3d_queue q_up = plan.overlapped ? g_pipe.up : nullptr, q_down = plan.overlapped ? g_pipe.down : nullptr;
  bool set_used[2] = {false, false};

  if (!silent) {
    std::printf("%d x %d planes, %d outer iterations per pass%s\n", (D + chunk - 1) / chunk, chunk, plan.outer_per_pass,
                plan.overlapped ? ", copies beside the kernels" : "");
    Utils::PrintProgressBar(0.f);
  }
  const size_t rows = static_cast<size_t>(planes) * H;
  const bool fused = FusedSweepsEnabled();
  size_t chunk_counter = 0;
  // Overlapped schedule, every field in the chunk sets: the windows of two neighbouring chunks share 2 x reach planes, and for the five
  // fields nothing writes -- the two frames, u, v, w -- those planes are in the OTHER chunk set already when a chunk starts: they are
  // copied across on the device (kernels' stream) and only the planes above them come over the link, so per pass each plane of
  // those fields travels once instead of (chunk + 2 x halo) / chunk times.  In the first pass the registered frame is handed on the
  // same way (registered once per plane).  F3D_P_HANDOVER=0 uploads every window whole.
  // The increments -- which the sweeps overwrite -- are handed on through three staging buffers: the planes the NEXT chunk shares with
  // this one are put aside as they arrived, before this chunk's first kernel, and the next chunk takes them from there.
  const char* ho_env = std::getenv("F3D_P_HANDOVER");
  const bool handover_any = plan.overlapped && halo > 0 && !(ho_env && ho_env[0] == '0');
  const bool handover = handover_any && !constants;
  const DevicePtr* staging = handover_any ? &staging_bufs[0] : nullptr;
  for (int i0 = 0; i0 < outer; i0 += plan.outer_per_pass) {
    const int n = std::min(plan.outer_per_pass, outer - i0);
    const int reach = halo ? n * (K + 1) : 0;  // planes of input this pass reads beyond the chunk
    int prev_set = -1, prev_hi = 0, prev_base = 0;   // the chunk before this one in the pass: its set, the top of its window, its z_base
    for (int z0 = 0; z0 < D; z0 += chunk, ++chunk_counter) {
      const int z1 = std::min(D, z0 + chunk);
      const int set = plan.overlapped ? static_cast<int>(chunk_counter & 1) : 0;
      DevicePtr buf[kAllFields];   // the buffer names of this residency: the sweeps trade DU .. DW with TDU .. TDW, phi / ksi with their seconds
      std::copy(sets[set], sets[set] + kAllFields, buf);
      const int base = halo ? z0 - halo : 0;  // global plane held by container plane 0
      const int lo = std::max(0, z0 - reach), hi = std::min(D, z1 + reach);
      auto window = [&](int grow) { return f3d_slab{base, std::max(0, z0 - grow), std::min(D, z1 + grow)}; };
      if (plan.overlapped && set_used[set]) {
        // the set's previous chunk must have left: its increments are still being read by the download queue
        if (CheckDeviceError(f3d_queue_wait_event(q_up, g_pipe.downloaded[set]))) return;
        if (CheckDeviceError(f3d_queue_wait_event(nullptr, g_pipe.downloaded[set]))) return;
      }
      const bool register_here = registering && i0 == 0;
      // the fields that do not change: the window of this residency into the chunk set -- or, held for the whole level, the planes
      // that have not arrived yet, in the first pass only (the chunks go up in z; a whole-level buffer is addressed like a chunk
      // buffer through a pointer under which its plane z sits at container plane z - base)
      if (constants)
        for (int i = 0; i < 5; ++i) buf[F0 + i] = Rebase(carve[i], box, 0, base);
      // ... or handed over from the chunk before: planes lo .. kept_hi of this window are in the other set (its kernels have read them,
      // nothing writes them); the copy runs on the kernels' stream, which has waited for this set's last download above
      int kept_hi = lo;
      if (handover && prev_set >= 0 && prev_set != set && prev_hi > lo) {
        kept_hi = std::min(prev_hi, hi);
        for (int i = 0; i < 5; ++i)
          if (CheckDeviceError(f3d_copy_rect_d2d(buf[F0 + i], box.pitch, box.H, static_cast<size_t>(lo - base), sets[prev_set][F0 + i], box.pitch,
                                                 box.H, static_cast<size_t>(lo - prev_base), W, H, static_cast<size_t>(kept_hi - lo))))
            return;
        // the other set is uploaded into again by the chunk after this one: not before these copies have read it
        if (CheckDeviceError(f3d_event_record_on(g_pipe.kept[set], nullptr))) return;
      }
      if (handover && prev_set >= 0 && set_used[set] && CheckDeviceError(f3d_queue_wait_event(q_up, g_pipe.kept[prev_set]))) return;
      const int c_lo = constants ? std::max(lo, constants_up_to) : kept_hi;
      if (!constants || i0 == 0) {
        for (int i = 0; i < 5; ++i)
          if (!(register_here && i == 1) && hi > c_lo && !Upload(buf[F0 + i], box, c_lo - base, *fixed[i], W, H, c_lo, hi - c_lo, q_up)) return;
        if (constants) constants_up_to = std::max(constants_up_to, hi);
      }
      // the unregistered frame 1 for the window lo .. hi, in pieces: piece p serves the output planes s[p] .. s[p+1] from the frame's
      // planes in_lo[p] .. (at most `planes` of them) held from plane 0 of its buffer
      const int r_lo = constants ? std::max(lo, registered_up_to) : kept_hi;   // (held or handed over: only what is not registered yet)
      int n_pieces = 0, piece_s[4] = {r_lo, r_lo, r_lo, r_lo}, piece_in[3] = {0, 0, 0};
      if (register_here) {
        if (constants) registered_up_to = std::max(registered_up_to, hi);
        for (int s0 = r_lo; s0 < hi;) {
          if (n_pieces == 3) return LowMemory(GetName());   // (ruled out above)
          const int in_lo = std::max(0, s0 - warp_reach);
          int s1 = hi;
          if (std::min(D, s1 + warp_reach) - in_lo > planes) s1 = in_lo + planes - warp_reach;
          if (s1 <= s0) return LowMemory(GetName());
          const int in_hi = std::min(D, s1 + warp_reach);
          // (into the set's own increment buffers: they start at zero in this pass, cleared below once the warp has read the pieces;
          // the sweeps' partners belong to both chunk sets and may be at work for the other one while this copy runs)
          if (!Upload(buf[DU + n_pieces], box, 0, *fixed[1], W, H, in_lo, in_hi - in_lo, q_up)) return;
          piece_in[n_pieces] = in_lo;
          piece_s[++n_pieces] = s1;
          s0 = s1;
        }
      }
      // the increments as the pass found them: zero in the first pass; afterwards the planes shared with the chunk before come from
      // the staging buffers (kernels' stream; this set's last download has been waited for above), the rest over the link
      const int inc_lo = (staging && i0 > 0 && prev_set >= 0 && prev_hi > lo) ? std::min(prev_hi, hi) : lo;
      for (int i = 0; i < 3; ++i) {
        if (i0 == 0) {
          if (!register_here && CheckDeviceError(f3d_memset2d(buf[DU + i], box.pitch, 0, W * sizeof(float), rows))) return;
          continue;
        }
        if (inc_lo > lo && CheckDeviceError(f3d_copy_rect_d2d(buf[DU + i], box.pitch, box.H, static_cast<size_t>(lo - base), staging[i], box.pitch,
                                                              box.H, 0, W, H, static_cast<size_t>(inc_lo - lo))))
          return;
        if (hi > inc_lo && !Upload(buf[DU + i], box, inc_lo - base, *inc[i], W, H, inc_lo, hi - inc_lo, q_up)) return;
      }
      if (plan.overlapped) {
        if (CheckDeviceError(f3d_event_record_on(g_pipe.uploaded[set], q_up))) return;
        if (CheckDeviceError(f3d_queue_wait_event(nullptr, g_pipe.uploaded[set]))) return;
      }
      // ... and what the NEXT chunk shares of them goes aside before the first kernel of this one writes a partner of theirs
      if (staging && i0 > 0 && z1 < D) {
        const int next_lo = std::max(0, z1 - reach);
        for (int i = 0; i < 3; ++i)
          if (hi > next_lo && CheckDeviceError(f3d_copy_rect_d2d(staging[i], box.pitch, box.H, 0, buf[DU + i], box.pitch, box.H,
                                                                 static_cast<size_t>(next_lo - base), W, H, static_cast<size_t>(hi - next_lo))))
            return;
      }
      // the registered frame of the window: every operand under the chunk's z_base, the piece's buffer rebased to it
      for (int p = 0; p < n_pieces; ++p) {
        const f3d_slab piece = {base, piece_s[p], piece_s[p + 1]};
        if (CheckDeviceError(f3d_warp(buf[F0], Rebase(buf[DU + p], box, piece_in[p], base), buf[FU], buf[FV], buf[FW], W, H, D, hx, hy, hz,
                                      buf[F1], &piece)))
          return;
      }
      if (register_here)
        for (int i = 0; i < 3; ++i)
          if (CheckDeviceError(f3d_memset2d(buf[DU + i], box.pitch, 0, W * sizeof(float), rows))) return;
      const DevicePtr registered_planes = buf[F1];   // (the buffer names below trade places with every sweep; F1 does not)
      // Outer iteration j of this pass leaves the increments valid on the chunk widened by g = (n-1-j)(K+1) planes:
      // phi/ksi on g + K, sweep s on g + K-1-s; a fused pair runs on the window of its second sweep.
      bool weights_ready = false;
      for (int j = 0; j < n; ++j) {
        const int g = halo ? (n - 1 - j) * (K + 1) : 0;
        const f3d_slab pw = window(g + K);
        if (!weights_ready &&
            CheckDeviceError(f3d_phi_ksi(buf[F0], buf[F1], buf[FU], buf[FV], buf[FW], buf[DU], buf[DV], buf[DW], W, H, D, hx, hy, hz,
                                         equation_smoothness, equation_data, buf[PHI], buf[KSI], &pw)))
          return;
        weights_ready = false;
        for (int s = 0; s < K;) {
          const bool pair = fused && s + 2 <= K;
          const f3d_slab sw = window(g + K - 1 - s - (pair ? 1 : 0));
          if (fuse_weights && !pair && s == K - 1 && j + 1 < n) {
            // The next outer iteration wants its weights on window(g - 1): this sweep's window shrunk by one plane wherever it does
            // not end at a face of the volume (the weights of a plane need the new increments of both z neighbours).  The launch
            // computes the sweep on those edge planes anyway and stores it there too (keep_below / keep_above).
            const int lo_in = sw.z_lo > 0 ? 1 : 0, hi_in = sw.z_hi < D ? 1 : 0;
            const f3d_slab ww = {sw.z_base, sw.z_lo + lo_in, sw.z_hi - hi_in};
            if (ww.z_hi - ww.z_lo >= 1) {
              if (CheckDeviceError(f3d_solve_sweep_phi_ksi_edges(buf[F0], buf[F1], buf[FU], buf[FV], buf[FW], buf[DU], buf[DV], buf[DW],
                                                                 buf[PHI], buf[KSI], W, H, D, hx, hy, hz, equation_alpha,
                                                                 equation_smoothness, equation_data, buf[TDU], buf[TDV], buf[TDW],
                                                                 buf[PHI2], buf[KSI2], &ww, lo_in, hi_in)))
                return;
              std::swap(buf[DU], buf[TDU]);
              std::swap(buf[DV], buf[TDV]);
              std::swap(buf[DW], buf[TDW]);
              std::swap(buf[PHI], buf[PHI2]);
              std::swap(buf[KSI], buf[KSI2]);
              weights_ready = true;
              s += 1;
              continue;
            }
          }
          const int status =
              pair ? f3d_solve_sweep2(buf[F0], buf[F1], buf[FU], buf[FV], buf[FW], buf[DU], buf[DV], buf[DW], buf[PHI], buf[KSI], W, H, D,
                                      hx, hy, hz, equation_alpha, buf[TDU], buf[TDV], buf[TDW], &sw)
                   : f3d_solve_sweep(buf[F0], buf[F1], buf[FU], buf[FV], buf[FW], buf[DU], buf[DV], buf[DW], buf[PHI], buf[KSI], W, H, D,
                                     hx, hy, hz, equation_alpha, buf[TDU], buf[TDV], buf[TDW], &sw);
          if (CheckDeviceError(status)) return;
          std::swap(buf[DU], buf[TDU]);
          std::swap(buf[DV], buf[TDV]);
          std::swap(buf[DW], buf[TDW]);
          s += pair ? 2 : 1;
        }
      }
      // the last residency of the chunk: the flow update on the device, on the planes the chunk owns (see add_increments_to_flow)
      if (add_increments_to_flow && i0 + n >= outer) {
        const DevicePtr sums[3] = {buf[DU], buf[DV], buf[DW]}, flows[3] = {buf[FU], buf[FV], buf[FW]};
        const f3d_slab own = {base, z0, z1};
        if (CheckDeviceError(f3d_add_n(sums, flows, 3, W, H, D, &own))) return;
      }
      if (plan.overlapped) {
        // an odd number of trades leaves the result in the sweeps' partners, which the next chunk's kernels write while this chunk's
        // download runs: the planes the chunk owns move into the set's own buffers first (a device copy, on the kernels' stream)
        for (int i = 0; i < 3; ++i)
          if (buf[DU + i] != sets[set][DU + i]) {
            if (CheckDeviceError(f3d_copy_rect_d2d(sets[set][DU + i], box.pitch, box.H, static_cast<size_t>(z0 - base), buf[DU + i], box.pitch, box.H,
                                                   static_cast<size_t>(z0 - base), W, H, static_cast<size_t>(z1 - z0))))
              return;
            buf[DU + i] = sets[set][DU + i];
          }
        if (CheckDeviceError(f3d_event_record_on(g_pipe.computed[set], nullptr))) return;
        if (CheckDeviceError(f3d_queue_wait_event(q_down, g_pipe.computed[set]))) return;
      }
      for (int i = 0; i < 3; ++i)
        if (!Download(*next[i], W, H, z0, z1 - z0, buf[DU + i], box, z0 - base, q_down)) return;
      if (register_here && !Download(*registered_to, W, H, z0, z1 - z0, registered_planes, box, z0 - base, q_down)) return;
      if (plan.overlapped) {
        if (CheckDeviceError(f3d_event_record_on(g_pipe.downloaded[set], q_down))) return;
        set_used[set] = true;
      }
      prev_set = set;
      prev_hi = hi;
      prev_base = base;
    }
    // the pass is complete when its last download is: the next pass reads what this one wrote
    if (plan.overlapped && CheckDeviceError(f3d_queue_sync(q_down))) return;
    if (CheckDeviceError(f3d_stream_sync())) return;
    if (registering && i0 == 0) {
      // the registered frame is complete on the host.  In a volume of its own: later residencies read that.  In the storage of
      // flow_du: "frame_1" takes that storage (it IS the registered frame from here on, as after the registration operator) and the
      // unregistered frame's storage goes where flow_du's would have gone below -- to temp_du, the next pass's download target.
      if (registered_frame_1) fixed[1] = registered_frame_1;
      else p_frame_1->Swap(*inc[0]);
      last_registered_ = true;
    }
    for (int i = 0; i < 3; ++i) inc[i]->Swap(*next[i]);
    ++last_passes_;
    if (add_increments_to_flow && i0 + n >= outer) last_added_ = true;
    if (!silent) {
      const float complete = static_cast<float>(i0 + n) / static_cast<float>(outer);
      Utils::PrintProgressBar(complete);
      std::printf(" % 3.0f%%", complete * 100);
    }
  }
  if (!silent) std::printf("\n");
}

// ---- Gaussian on host volumes (cuda_operation_convolution.cpp:134-184 chunked) ------------------------------------

void CudaOperationConvolution3DP::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  Data3D *p_input, *p_output;
  DataSize4 data_size;
  float gaussian_sigma;
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_input, "input");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_output, "output");
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  GET_PARAM_OR_RETURN(params, float, gaussian_sigma, "gaussian_sigma");
  if (p_input == p_output || p_input->DataPtr() == p_output->DataPtr()) {
    std::printf("Operation '%s': Error. Input buffer cannot serve as output buffer.", GetName());
    return;
  }
  if (!Fits(*p_input, data_size) || !Fits(*p_output, data_size)) {
    std::printf("Error: Operation '%s'. Wrong dimensions.\n", GetName());
    return;
  }
  if (!(gaussian_sigma > 0.f)) {
    std::printf("Operation '%s': Error. gaussian_sigma must be positive.\n", GetName());
    return;
  }
  taps_.ComputeGaussianKernel(gaussian_sigma, 3, 1.0);  // prints and leaves no taps when the radius exceeds the 51-tap limit
  const int R = static_cast<int>(taps_.KernelRadius());
  if (2 * R + 1 > 51) return;
  if (CheckDeviceError(f3d_set_conv_taps(taps_.Kernel(), 2 * static_cast<size_t>(R) + 1))) return;
  const size_t W = data_size.width, H = data_size.height;
  const int D = static_cast<int>(data_size.depth);
  if (W == 0 || H == 0 || D == 0) return;

  // three buffers of chunk + 2 R planes: rows and columns on the chunk widened by the tap radius, slices on the chunk
  const ChunkBox box(W, H);
  const size_t total = box.TotalPlanes(PiecemealBudgetBytes(), 3) / 3;
  const int planes = static_cast<int>(std::min<size_t>(total, static_cast<size_t>(D)));
  const int chunk = planes >= D ? D : planes - 2 * R;
  if (chunk < 1) return LowMemory(GetName());
  Carver buf(box);
  for (int i = 0; i < 3; ++i) buf.Add(planes);
  if (!buf.Commit()) return;
  ContainerScope scope(box, planes);
  if (!scope.ok()) return;
  for (int z0 = 0; z0 < D; z0 += chunk) {
    const int z1 = std::min(D, z0 + chunk);
    const int lo = std::max(0, z0 - R), hi = std::min(D, z1 + R);
    const int base = chunk == D ? 0 : z0 - R;
    const f3d_slab wide = {base, lo, hi}, own = {base, z0, z1};
    if (!Upload(buf[0], box, lo - base, *p_input, W, H, lo, hi - lo)) return;
    if (CheckDeviceError(f3d_conv_rows_cols(buf[2], buf[0], W, H, D, R, &wide))) return;  // rows + columns, one launch
    if (CheckDeviceError(f3d_conv_slices(buf[1], buf[2], W, H, D, R, &own))) return;
    if (!Download(*p_output, W, H, z0, z1 - z0, buf[1], box, z0 - base)) return;
  }
  CheckDeviceError(f3d_stream_sync());
}

// ---- median on host volumes (cuda_operation_median.cpp:72-149 chunked) ---------------------------------------------

void CudaOperationMedianP::Execute(OperationParameters& params)
{
  if (!IsInitialized()) return;
  Data3D *p_input, *p_output;
  DataSize4 data_size;
  size_t radius;
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_input, "input");
  GET_PARAM_PTR_OR_RETURN(params, Data3D, p_output, "output");
  GET_PARAM_OR_RETURN(params, DataSize4, data_size, "data_size");
  GET_PARAM_OR_RETURN(params, size_t, radius, "radius");
  if (!Fits(*p_input, data_size) || !Fits(*p_output, data_size)) {
    std::printf("Error: Operation '%s'. Wrong dimensions.\n", GetName());
    return;
  }
  const size_t W = data_size.width, H = data_size.height;
  const int D = static_cast<int>(data_size.depth);
  if (W == 0 || H == 0 || D == 0) return;
  const bool in_place = p_input == p_output || p_input->DataPtr() == p_output->DataPtr();
  if (radius != 1 && radius % 2 == 0) {
    std::printf("Warning. Median raduis is even (%zu), decresaing by 1...\n", radius);
    radius -= 1;
  }
  if (radius == 1) {  // nothing to filter
    if (!in_place)
      for (int z = 0; z < D; ++z)
        for (size_t y = 0; y < H; ++y)
          std::copy_n(PlanePtr(*p_input, z) + y * p_input->Width(), W, PlanePtr(*p_output, z) + y * p_output->Width());
    return;
  }
  if (radius < 3 || radius > 7) {
    std::printf("Error. Wrong median raduis (%zu). Supported values: 3, 5, 7\n", radius);
    return;
  }
  const int half = static_cast<int>(radius) / 2;

  // input: chunk + 2 half planes, output: chunk planes.  A later chunk reads `half` planes an earlier one has already
  // replaced in the host volume when the filter runs in place, so those planes are carried over on the device: the tail of
  // the input buffer moves to its head before the rest of the next chunk is uploaded.
  const ChunkBox box(W, H);
  const size_t total = box.TotalPlanes(PiecemealBudgetBytes(), 2);
  int chunk = D;
  if (total < 2 * static_cast<size_t>(D)) chunk = static_cast<int>((total - std::min<size_t>(total, 2 * static_cast<size_t>(half))) / 2);
  if (chunk < 2 * half + 1 && chunk < D) return LowMemory(GetName());
  const int in_planes = chunk >= D ? D : chunk + 2 * half;  // plane 0 of the buffer is global plane z0 - half, also below plane 0
  Carver buf(box);
  buf.Add(in_planes);
  buf.Add(chunk);
  if (!buf.Commit()) return;
  ContainerScope scope(box, in_planes);
  if (!scope.ok()) return;
  int have_hi = 0;  // one past the last input plane the buffer holds from the previous chunk
  for (int z0 = 0; z0 < D; z0 += chunk) {
    const int z1 = std::min(D, z0 + chunk);
    const int lo = std::max(0, z0 - half), hi = std::min(D, z1 + half);
    const int base = chunk == D ? 0 : z0 - half;
    int fresh_lo = lo;
    if (z0 > 0) {
      // planes [z0 - half, have_hi) sit at the tail of the buffer from the previous chunk (from its plane `chunk` on)
      const int carried = std::min(have_hi, z0 + half) - (z0 - half);
      if (CheckDeviceError(f3d_copy_planes(buf[0], 0, buf[0], chunk, carried, W, H))) return;
      fresh_lo = z0 - half + carried;
    }
    have_hi = hi;
    if (hi > fresh_lo && !Upload(buf[0], box, fresh_lo - base, *p_input, W, H, fresh_lo, hi - fresh_lo)) return;
    const f3d_slab own = {base, z0, z1};
    if (CheckDeviceError(f3d_median(buf[0], W, H, D, radius, Rebase(buf[1], box, z0, base), &own))) return;
    if (!Download(*p_output, W, H, z0, z1 - z0, buf[1], box, 0)) return;
  }
  CheckDeviceError(f3d_stream_sync());
}
