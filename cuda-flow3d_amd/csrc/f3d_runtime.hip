// libf3d_hip.so runtime: context, stream, pitched memory, 3-D copies, events, per-kernel timing.
// Replaces the CUDA driver-API uses listed in SURVEY.md 2c (cuInit ... cuEventElapsedTime); each entry
// point cites its reference call site in include/f3d.h.
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <algorithm>
#include <csignal>
#include <fcntl.h>
#include <map>
#include <mutex>
#include <unistd.h>
#include <vector>

#include "f3d_internal.h"

namespace {

thread_local char g_error[512] = "";

// What a sequence of launches shares: the stream they are ordered on, the container geometry the kernels address with, the
// taps of the blur.  The library has one such lane of its own (the "library stream" of include/f3d.h); a driver that wants to run
// beside another one in the same process makes a lane of its own and binds it to its thread (f3d_lane_*).
struct Lane {
  hipStream_t stream = nullptr;
  f3d_size4 container = {0, 0, 0, 0};
  f3d::ConvTaps taps = {{0}, 0};
};
struct State {
  bool ready = false;
  int device = -1;
  Lane lane;   // the default lane
  hipDeviceProp_t prop;
} S;
thread_local Lane* t_lane = nullptr;   // the calling thread's lane; null = the default one
inline Lane& cur() { return t_lane ? *t_lane : S.lane; }
std::mutex g_mutex;   // allocation table and event pool: the only library state two lanes share

struct ProfRec {
  int kernel;
  size_t voxels;
  hipEvent_t start, stop;
};
struct Prof {
  bool enabled = false;
  unsigned mask = ~0u;  // kernels that get events while enabled (bit = kernel id)
  std::vector<ProfRec> pending;
  std::vector<hipEvent_t> pool;
  double ms[F3D_K_COUNT][2] = {};  // [kernel][0 = all, 1 = filtered] is rebuilt on read
  hipEvent_t open_start[F3D_K_COUNT] = {};
  size_t open_voxels[F3D_K_COUNT] = {};
} P;

hipEvent_t take_event()
{
  if (!P.pool.empty()) {
    hipEvent_t e = P.pool.back();
    P.pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

}  // namespace

namespace f3d {

int fail(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return 1;
}

int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
  return fail("HIP error %d (%s) in %s at %s:%d", static_cast<int>(e), hipGetErrorString(e), what, file, line);
}

hipStream_t stream() { return cur().stream; }
bool ready() { return S.ready; }
const f3d_size4& container() { return cur().container; }
const ConvTaps& conv_taps() { return cur().taps; }

bool make_geo(F3dGeo* g, size_t w, size_t h, size_t d, const f3d_slab* slab, const char* who)
{
  const f3d_size4& c = cur().container;
  if (c.pitch == 0 || c.height == 0) {
    fail("%s: f3d_set_container() has not been called", who);
    return false;
  }
  if (w == 0 || h == 0 || d == 0 || w > c.width || h > c.height || w * sizeof(float) > c.pitch) {
    fail("%s: level %zux%zux%zu does not fit the container %zux%zux%zu (pitch %zu B)", who, w, h, d, c.width,
         c.height, c.depth, c.pitch);
    return false;
  }
  g->W = static_cast<int>(w);
  g->H = static_cast<int>(h);
  g->D = static_cast<int>(d);
  g->Hc = static_cast<int>(c.height);
  g->pitch = static_cast<int>(c.pitch / sizeof(float));
  if (slab) {
    g->z_base = slab->z_base;
    g->z_lo = slab->z_lo;
    g->z_hi = slab->z_hi;
    if (g->z_lo < 0 || g->z_hi > g->D || g->z_lo > g->z_hi || g->z_lo < g->z_base ||
        static_cast<size_t>(g->z_hi - g->z_base) > c.depth) {
      fail("%s: slab planes [%d,%d) base %d outside level depth %d / container depth %zu", who, g->z_lo, g->z_hi,
           g->z_base, g->D, c.depth);
      return false;
    }
  } else {
    g->z_base = 0;
    g->z_lo = 0;
    g->z_hi = g->D;
    if (d > c.depth) {
      fail("%s: depth %zu exceeds container depth %zu", who, d, c.depth);
      return false;
    }
  }
  return true;
}

void prof_begin(int kernel, size_t voxels)
{
  if (t_lane) return;   // the event bracket belongs to the default lane (bench.py, tools/kbench.py)
  if (!P.enabled || !((P.mask >> kernel) & 1u)) return;
  hipEvent_t e = take_event();
  if (!e) return;
  (void)hipEventRecord(e, cur().stream);
  P.open_start[kernel] = e;
  P.open_voxels[kernel] = voxels;
}

void prof_end(int kernel)
{
  if (t_lane) return;
  if (!P.enabled || !P.open_start[kernel]) return;
  hipEvent_t e = take_event();
  if (!e) return;
  (void)hipEventRecord(e, cur().stream);
  P.pending.push_back({kernel, P.open_voxels[kernel], P.open_start[kernel], e});
  P.open_start[kernel] = nullptr;
}

}  // namespace f3d

std::map<void*, void*> g_alloc_base;  // staggered allocations: user pointer -> hipMalloc pointer

// rows x words dwords of a pitched array set to one 32-bit pattern (grid.y strides over the rows)
__global__ __launch_bounds__(256) void k_fill_rows(unsigned* __restrict__ p, size_t pitch_words, unsigned word, unsigned words,
                                                   unsigned rows)
{
  const unsigned x = blockIdx.x * 256u + threadIdx.x;
  if (x >= words) return;
  for (unsigned r = blockIdx.y; r < rows; r += gridDim.y) p[static_cast<size_t>(r) * pitch_words + x] = word;
}

extern "C" {

const char* f3d_last_error(void) { return g_error; }

int f3d_device_count(int* count)
{
  if (!count) return f3d::fail("f3d_device_count: null argument");
  F3D_HIP(hipGetDeviceCount(count));
  return 0;
}

int f3d_init(int device)
{
  if (S.ready) return 0;
  int n = 0;
  F3D_HIP(hipGetDeviceCount(&n));
  if (n == 0) return f3d::fail("f3d_init: no HIP device is visible");
  if (device < 0) {
    const char* lr = std::getenv("LOCAL_RANK");
    device = lr ? std::atoi(lr) % n : 0;
  }
  if (device >= n) return f3d::fail("f3d_init: device %d requested but only %d visible", device, n);
  F3D_HIP(hipSetDevice(device));
  F3D_HIP(hipGetDeviceProperties(&S.prop, device));
  if (std::strncmp(S.prop.gcnArchName, "gfx950", 6) != 0) {
    return f3d::fail("f3d_init: device %d is %s; this library carries gfx950 (MI355X) code only", device,
                     S.prop.gcnArchName);
  }
  F3D_HIP(hipStreamCreateWithFlags(&S.lane.stream, hipStreamNonBlocking));
  S.device = device;
  S.ready = true;
  return 0;
}

int f3d_shutdown(void)
{
  if (!S.ready) return 0;
  (void)hipStreamSynchronize(S.lane.stream);
  for (auto& r : P.pending) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  P.pending.clear();
  for (auto e : P.pool) (void)hipEventDestroy(e);
  P.pool.clear();
  (void)hipStreamDestroy(S.lane.stream);
  S.lane.stream = nullptr;
  t_lane = nullptr;
  S.ready = false;
  return 0;
}

int f3d_is_initialized(void) { return S.ready ? 1 : 0; }

// ---- lanes: a stream and a container geometry of one's own --------------------------------------------------------------------
struct f3d_lane_s {
  Lane lane;
};

int f3d_lane_create(f3d_lane* lane)
{
  F3D_REQUIRE_READY("f3d_lane_create");
  if (!lane) return f3d::fail("f3d_lane_create: null argument");
  F3D_HIP(hipSetDevice(S.device));   // the device is per-thread state of the runtime: a worker thread starts on device 0
  f3d_lane_s* l = new f3d_lane_s();
  const hipError_t e = hipStreamCreateWithFlags(&l->lane.stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete l;
    return f3d::hip_fail(e, "hipStreamCreateWithFlags", __FILE__, __LINE__);
  }
  *lane = l;
  return 0;
}

int f3d_lane_make_current(f3d_lane lane)
{
  F3D_REQUIRE_READY("f3d_lane_make_current");
  F3D_HIP(hipSetDevice(S.device));
  t_lane = lane ? &lane->lane : nullptr;
  return 0;
}

int f3d_lane_is_private(void) { return t_lane ? 1 : 0; }

int f3d_lane_get_current(f3d_lane* lane)
{
  if (!lane) return f3d::fail("f3d_lane_get_current: null argument");
  // f3d_lane_s holds nothing but its Lane: the handle is the address of that member
  *lane = t_lane ? reinterpret_cast<f3d_lane>(reinterpret_cast<char*>(t_lane) - offsetof(f3d_lane_s, lane)) : nullptr;
  return 0;
}

int f3d_lane_destroy(f3d_lane lane)
{
  if (!lane) return 0;
  if (t_lane == &lane->lane) t_lane = nullptr;
  if (S.ready && lane->lane.stream) {
    (void)hipStreamSynchronize(lane->lane.stream);
    (void)hipStreamDestroy(lane->lane.stream);
  }
  delete lane;
  return 0;
}

namespace {
struct Roctx {
  bool tried = false;
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
} X;

bool roctx_ready()
{
  if (X.tried) return X.push != nullptr;
  X.tried = true;
  const char* want = std::getenv("F3D_ROCTX");
  bool on = want && want[0] == '1';
  if (!want) {  // a rocprofiler tool library in the process: ranges are wanted
    for (const char* var : {"LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"}) {
      const char* v = std::getenv(var);
      if (v && std::strstr(v, "rocprof")) on = true;
    }
  }
  if (!on) return false;
  void* h = nullptr;
  for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return false;
  X.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
  X.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
  if (!X.push || !X.pop) X.push = nullptr;
  return X.push != nullptr;
}
}  // namespace

int f3d_range_push(const char* name)
{
  if (name && roctx_ready()) X.push(name);
  return 0;
}

int f3d_range_pop(void)
{
  if (roctx_ready()) X.pop();
  return 0;
}

// ---- load map on a fatal signal ------------------------------------------------------------------------------------------
// A native stack trace without the load addresses of the libraries is a list of numbers (profiles/r02_pmc_only_crash_stack.txt
// had to be resolved after the fact by fingerprinting return sites, tools/resolve_frames.py).  With this switched on, a fatal
// signal first copies /proc/self/maps into a file -- open / read / write only, all async-signal-safe -- then puts the previous
// handler back and returns, so the faulting instruction faults again into whoever was there before (Python's faulthandler, a
// profiler's handler, the default action).
namespace {
char g_maps_path[512];
struct sigaction g_prev_action[NSIG];
const int kFatal[] = {SIGSEGV, SIGBUS, SIGILL, SIGFPE, SIGABRT};

void put_hex(int fd, const char* label, unsigned long long v)
{
  char buf[96];
  int n = 0;
  while (label[n] && n < 60) { buf[n] = label[n]; ++n; }
  buf[n++] = '0'; buf[n++] = 'x';
  for (int shift = 60; shift >= 0; shift -= 4) buf[n++] = "0123456789abcdef"[(v >> shift) & 0xf];
  buf[n++] = '\n';
  (void)!write(fd, buf, n);
}

void maps_on_fatal_signal(int sig, siginfo_t* info, void*)
{
  const int out = open(g_maps_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (out >= 0) {
    put_hex(out, "signal ", static_cast<unsigned long long>(sig));
    put_hex(out, "fault address ", reinterpret_cast<unsigned long long>(info ? info->si_addr : nullptr));
    const int in = open("/proc/self/maps", O_RDONLY);
    if (in >= 0) {
      char buf[4096];
      for (;;) {
        const ssize_t n = read(in, buf, sizeof(buf));
        if (n <= 0) break;
        (void)!write(out, buf, static_cast<size_t>(n));
      }
      close(in);
    }
    close(out);
  }
  sigaction(sig, &g_prev_action[sig], nullptr);
  if (!info || info->si_code <= 0 || sig == SIGABRT) raise(sig);  // sent, not faulted: deliver it to the previous handler now
}
}  // namespace

int f3d_crash_maps_enable(const char* path)
{
  if (!path || !path[0] || std::strlen(path) >= sizeof(g_maps_path)) return f3d::fail("f3d_crash_maps_enable: bad path");
  const bool first = g_maps_path[0] == 0;
  std::strcpy(g_maps_path, path);
  if (!first) return 0;  // handlers are in place, only the file name changed
  struct sigaction sa;
  std::memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = maps_on_fatal_signal;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  for (int sig : kFatal) sigaction(sig, &sa, &g_prev_action[sig]);
  return 0;
}

int f3d_device_name(char* name, size_t capacity)
{
  F3D_REQUIRE_READY("f3d_device_name");
  if (!name || capacity == 0) return f3d::fail("f3d_device_name: null argument");
  std::snprintf(name, capacity, "%s (%s)", S.prop.name, S.prop.gcnArchName);
  return 0;
}

int f3d_mem_info(size_t* free_bytes, size_t* total_bytes)
{
  F3D_REQUIRE_READY("f3d_mem_info");
  F3D_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return 0;
}

int f3d_lds_per_workgroup(int* bytes)
{
  F3D_REQUIRE_READY("f3d_lds_per_workgroup");
  *bytes = static_cast<int>(S.prop.sharedMemPerBlock);
  return 0;
}

int f3d_alloc_pitched(f3d_devptr* ptr, size_t* pitch, size_t width_bytes, size_t rows)
{
  F3D_REQUIRE_READY("f3d_alloc_pitched");
  if (!ptr || !pitch || width_bytes == 0 || rows == 0) return f3d::fail("f3d_alloc_pitched: bad argument");
  const size_t align = 256;  // whole 256-B wave rows; also keeps every row 16-B aligned for dwordx4 access
  size_t p = (width_bytes + align - 1) / align * align;
  // Successive allocations are staggered by 17 x 256 B (modulo 64 KiB; F3D_ALLOC_SKEW = other byte count, 0 = off).
  // Equally sized containers otherwise start at the same offset of every power-of-two stride of the memory system, and the
  // one-sweep kernel, which streams 13 of them in lockstep, runs 10 % slower at 512^3 (1.70 -> 1.54 ms; any non-zero
  // stagger gives the same gain; phi/ksi and the fused pair do not care).
  static const size_t skew_unit = [] {
    const char* e = std::getenv("F3D_ALLOC_SKEW");
    return e ? static_cast<size_t>(std::atol(e)) / 256 * 256 : static_cast<size_t>(17 * 256);
  }();
  static size_t serial = 0;
  void* d = nullptr;
  F3D_HIP(hipMalloc(&d, p * rows + align + (skew_unit ? 65536 : 0)));
  std::lock_guard<std::mutex> lock(g_mutex);
  const size_t skew = skew_unit ? (serial++ * skew_unit) % 65536 : 0;
  char* user = static_cast<char*>(d) + skew;
  if (skew_unit) g_alloc_base[user] = d;
  *ptr = static_cast<f3d_devptr>(reinterpret_cast<uintptr_t>(user));
  *pitch = p;
  return 0;
}

int f3d_free(f3d_devptr ptr)
{
  F3D_REQUIRE_READY("f3d_free");
  void* user = f3d_ptr<void>(ptr);
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_alloc_base.find(user);
    if (it != g_alloc_base.end()) {
      user = it->second;
      g_alloc_base.erase(it);
    }
  }
  F3D_HIP(hipFree(user));
  return 0;
}

int f3d_memset2d(f3d_devptr ptr, size_t pitch, int value, size_t width_bytes, size_t rows)
{
  F3D_REQUIRE_READY("f3d_memset2d");
  if (width_bytes == 0 || rows == 0) return 0;
  // The driver clears du, dv, dw with this once per level: dword-aligned rows go through a plain fill kernel (the
  // runtime's 2-D fill reaches ~0.3 TB/s on a 512^3 sub-box), anything else through hipMemset2DAsync.
  if (pitch % 4 == 0 && width_bytes % 4 == 0 && (static_cast<uintptr_t>(ptr) % 4) == 0 && width_bytes <= pitch &&
      rows <= 0x7fffffffu) {
    const unsigned b = static_cast<unsigned>(value) & 0xffu;
    const unsigned word = b | (b << 8) | (b << 16) | (b << 24);
    const unsigned words = static_cast<unsigned>(width_bytes / 4);
    const dim3 block(256, 1, 1), grid((words + 255) / 256, static_cast<unsigned>(rows > 65535 ? 65535 : rows), 1);
    hipLaunchKernelGGL(k_fill_rows, grid, block, 0, cur().stream, f3d_ptr<unsigned>(ptr), static_cast<size_t>(pitch / 4), word, words,
                       static_cast<unsigned>(rows));
    F3D_HIP(hipGetLastError());
    return 0;
  }
  F3D_HIP(hipMemset2DAsync(f3d_ptr<void>(ptr), pitch, value, width_bytes, rows, cur().stream));
  return 0;
}

// Dense host volume <-> pitched container.  A volume as wide as the pitch is one contiguous range and moves with a plain 1-D
// copy; otherwise the rows move as 2-D copies of at most kRowsPerCopy rows -- one 2-D copy of height x depth rows (262 144 at
// 512^3) into pageable memory is the call under which the one recorded crash of this library happened (profiles/
// r03_crash_frames.md: inside the packet interceptor a profiler had registered with the HSA runtime, reached from the blit
// dispatch of exactly such a copy), and nothing is gained by handing the runtime the whole volume as a single rectangle.
namespace {
constexpr size_t kRowsPerCopy = 32768;
int copy_dense(char* dev, size_t dev_pitch, size_t dev_height, char* host, size_t width, size_t height, size_t depth, bool to_device)
{
  const size_t wb = width * sizeof(float);
  const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
  if (height == dev_height && wb == dev_pitch) {
    const size_t bytes = wb * height * depth;
    F3D_HIP(to_device ? hipMemcpyAsync(dev, host, bytes, kind, cur().stream) : hipMemcpyAsync(host, dev, bytes, kind, cur().stream));
    return 0;
  }
  // planes per call: whole planes, at most kRowsPerCopy rows (sub-boxes lower than the container go plane by plane)
  const size_t per_call = height == dev_height ? std::max<size_t>(1, kRowsPerCopy / height) : 1;
  for (size_t z = 0; z < depth; z += per_call) {
    const size_t n = std::min(per_call, depth - z);
    char* d = dev + z * dev_height * dev_pitch;
    char* h = host + z * height * wb;
    if (to_device) F3D_HIP(hipMemcpy2DAsync(d, dev_pitch, h, wb, wb, height * n, kind, cur().stream));
    else F3D_HIP(hipMemcpy2DAsync(h, wb, d, dev_pitch, wb, height * n, kind, cur().stream));
  }
  return 0;
}
}  // namespace

int f3d_copy3d_h2d(f3d_devptr dst, size_t dev_pitch, size_t dev_height, size_t dev_plane0, const float* src,
                   size_t width, size_t height, size_t depth)
{
  F3D_REQUIRE_READY("f3d_copy3d_h2d");
  if (height > dev_height || width * sizeof(float) > dev_pitch) return f3d::fail("f3d_copy3d_h2d: volume exceeds container");
  char* d = f3d_ptr<char>(dst) + dev_plane0 * dev_height * dev_pitch;
  if (copy_dense(d, dev_pitch, dev_height, reinterpret_cast<char*>(const_cast<float*>(src)), width, height, depth, true)) return 1;
  F3D_HIP(hipStreamSynchronize(cur().stream));
  return 0;
}

int f3d_copy3d_d2h(float* dst, size_t width, size_t height, size_t depth, f3d_devptr src, size_t dev_pitch,
                   size_t dev_height, size_t dev_plane0)
{
  F3D_REQUIRE_READY("f3d_copy3d_d2h");
  if (height > dev_height || width * sizeof(float) > dev_pitch) return f3d::fail("f3d_copy3d_d2h: volume exceeds container");
  char* s = f3d_ptr<char>(src) + dev_plane0 * dev_height * dev_pitch;
  if (copy_dense(s, dev_pitch, dev_height, reinterpret_cast<char*>(dst), width, height, depth, false)) return 1;
  F3D_HIP(hipStreamSynchronize(cur().stream));
  return 0;
}

// Plane-range copies between a host volume with its own row / plane strides (a level's sub-box inside a full-size dense
// volume) and a pitched container.  Asynchronous on the library stream: with page-locked host memory (f3d_host_register)
// the call returns at once and the caller orders reuse of the host planes with f3d_stream_sync().
struct f3d_queue_s {
  hipStream_t stream;
};
static hipStream_t stream_of(f3d_queue q) { return q ? q->stream : cur().stream; }

int f3d_copy_planes_h2d(f3d_devptr dst, size_t dev_pitch, size_t dev_height, size_t dev_plane0, const float* src,
                        size_t src_row_floats, size_t src_rows, size_t width, size_t height, size_t depth)
{
  return f3d_copy_planes_h2d_on(nullptr, dst, dev_pitch, dev_height, dev_plane0, src, src_row_floats, src_rows, width, height, depth);
}

int f3d_copy_planes_h2d_on(f3d_queue queue, f3d_devptr dst, size_t dev_pitch, size_t dev_height, size_t dev_plane0, const float* src,
                           size_t src_row_floats, size_t src_rows, size_t width, size_t height, size_t depth)
{
  F3D_REQUIRE_READY("f3d_copy_planes_h2d");
  const hipStream_t stream = stream_of(queue);
  if (depth == 0) return 0;
  if (!src || height > dev_height || width * sizeof(float) > dev_pitch || width > src_row_floats || height > src_rows)
    return f3d::fail("f3d_copy_planes_h2d: %zux%zu does not fit the container or the host volume", width, height);
  char* d = f3d_ptr<char>(dst) + dev_plane0 * dev_height * dev_pitch;
  const size_t wb = width * sizeof(float), sb = src_row_floats * sizeof(float);
  if (height == dev_height && height == src_rows) {
    F3D_HIP(hipMemcpy2DAsync(d, dev_pitch, src, sb, wb, height * depth, hipMemcpyHostToDevice, stream));
  } else {  // one 3-D rectangle copy, not a copy per plane: a coarse level is hundreds of small planes
    hipMemcpy3DParms p = {};
    p.srcPtr = make_hipPitchedPtr(const_cast<float*>(src), sb, src_row_floats, src_rows);
    p.dstPtr = make_hipPitchedPtr(d, dev_pitch, dev_pitch / sizeof(float), dev_height);
    p.extent = make_hipExtent(wb, height, depth);
    p.kind = hipMemcpyHostToDevice;
    F3D_HIP(hipMemcpy3DAsync(&p, stream));
  }
  return 0;
}

int f3d_copy_planes_d2h(float* dst, size_t dst_row_floats, size_t dst_rows, size_t width, size_t height, size_t depth,
                        f3d_devptr src, size_t dev_pitch, size_t dev_height, size_t dev_plane0)
{
  return f3d_copy_planes_d2h_on(nullptr, dst, dst_row_floats, dst_rows, width, height, depth, src, dev_pitch, dev_height, dev_plane0);
}

int f3d_copy_planes_d2h_on(f3d_queue queue, float* dst, size_t dst_row_floats, size_t dst_rows, size_t width, size_t height,
                           size_t depth, f3d_devptr src, size_t dev_pitch, size_t dev_height, size_t dev_plane0)
{
  F3D_REQUIRE_READY("f3d_copy_planes_d2h");
  const hipStream_t stream = stream_of(queue);
  if (depth == 0) return 0;
  if (!dst || height > dev_height || width * sizeof(float) > dev_pitch || width > dst_row_floats || height > dst_rows)
    return f3d::fail("f3d_copy_planes_d2h: %zux%zu does not fit the container or the host volume", width, height);
  const char* s = f3d_ptr<const char>(src) + dev_plane0 * dev_height * dev_pitch;
  const size_t wb = width * sizeof(float), db = dst_row_floats * sizeof(float);
  if (height == dev_height && height == dst_rows) {
    F3D_HIP(hipMemcpy2DAsync(dst, db, s, dev_pitch, wb, height * depth, hipMemcpyDeviceToHost, stream));
  } else {
    hipMemcpy3DParms p = {};
    p.srcPtr = make_hipPitchedPtr(const_cast<char*>(s), dev_pitch, dev_pitch / sizeof(float), dev_height);
    p.dstPtr = make_hipPitchedPtr(dst, db, dst_row_floats, dst_rows);
    p.extent = make_hipExtent(wb, height, depth);
    p.kind = hipMemcpyDeviceToHost;
    F3D_HIP(hipMemcpy3DAsync(&p, stream));
  }
  return 0;
}

int f3d_copy_rect_d2d(f3d_devptr dst, size_t dst_pitch, size_t dst_rows, size_t dst_plane0, f3d_devptr src, size_t src_pitch,
                      size_t src_rows, size_t src_plane0, size_t width, size_t height, size_t depth)
{
  F3D_REQUIRE_READY("f3d_copy_rect_d2d");
  if (depth == 0) return 0;
  const size_t wb = width * sizeof(float);
  if (height > dst_rows || height > src_rows || wb > dst_pitch || wb > src_pitch)
    return f3d::fail("f3d_copy_rect_d2d: %zux%zu does not fit one of the containers", width, height);
  hipMemcpy3DParms p = {};
  p.srcPtr = make_hipPitchedPtr(f3d_ptr<char>(src) + src_plane0 * src_rows * src_pitch, src_pitch, src_pitch / sizeof(float), src_rows);
  p.dstPtr = make_hipPitchedPtr(f3d_ptr<char>(dst) + dst_plane0 * dst_rows * dst_pitch, dst_pitch, dst_pitch / sizeof(float), dst_rows);
  p.extent = make_hipExtent(wb, height, depth);
  p.kind = hipMemcpyDeviceToDevice;
  F3D_HIP(hipMemcpy3DAsync(&p, cur().stream));
  return 0;
}

int f3d_host_register(void* ptr, size_t bytes)
{
  F3D_REQUIRE_READY("f3d_host_register");
  if (!ptr || bytes == 0) return f3d::fail("f3d_host_register: empty range");
  F3D_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return 0;
}

int f3d_host_is_pinned(const void* ptr, int* yes)
{
  F3D_REQUIRE_READY("f3d_host_is_pinned");
  if (!ptr || !yes) return f3d::fail("f3d_host_is_pinned: null argument");
  hipPointerAttribute_t attr;
  const hipError_t r = hipPointerGetAttributes(&attr, ptr);
  if (r != hipSuccess) {
    (void)hipGetLastError();  // ordinary pageable memory is not an error for the caller
    *yes = 0;
    return 0;
  }
  *yes = attr.type == hipMemoryTypeHost ? 1 : 0;
  return 0;
}

int f3d_host_unregister(void* ptr)
{
  F3D_REQUIRE_READY("f3d_host_unregister");
  F3D_HIP(hipHostUnregister(ptr));
  return 0;
}

int f3d_copy_d2d(f3d_devptr dst, f3d_devptr src, size_t bytes)
{
  F3D_REQUIRE_READY("f3d_copy_d2d");
  F3D_HIP(hipMemcpyAsync(f3d_ptr<void>(dst), f3d_ptr<const void>(src), bytes, hipMemcpyDeviceToDevice, cur().stream));
  return 0;
}

int f3d_set_container(const f3d_size4* c)
{
  if (!c || c->width == 0 || c->height == 0 || c->depth == 0 || c->pitch < c->width * sizeof(float) ||
      c->pitch % sizeof(float) != 0)
    return f3d::fail("f3d_set_container: invalid container size");
  cur().container = *c;
  return 0;
}

int f3d_get_container(f3d_size4* c)
{
  if (!c) return f3d::fail("f3d_get_container: null output");
  *c = cur().container;
  return 0;
}

int f3d_set_conv_taps(const float* taps, size_t count)
{
  if (!taps || count == 0 || count > 51 || count % 2 == 0)
    return f3d::fail("f3d_set_conv_taps: need an odd tap count of at most 51, got %zu", count);
  std::memcpy(cur().taps.k, taps, count * sizeof(float));
  cur().taps.count = static_cast<int>(count);
  return 0;
}

struct f3d_event_s {
  hipEvent_t ev;
};

int f3d_event_create(f3d_event* ev)
{
  F3D_REQUIRE_READY("f3d_event_create");
  f3d_event e = new f3d_event_s;
  hipError_t r = hipEventCreate(&e->ev);
  if (r != hipSuccess) {
    delete e;
    return f3d::hip_fail(r, "hipEventCreate", __FILE__, __LINE__);
  }
  *ev = e;
  return 0;
}

int f3d_event_record(f3d_event ev)
{
  F3D_REQUIRE_READY("f3d_event_record");
  F3D_HIP(hipEventRecord(ev->ev, cur().stream));
  return 0;
}

int f3d_queue_create(f3d_queue* queue)
{
  F3D_REQUIRE_READY("f3d_queue_create");
  if (!queue) return f3d::fail("f3d_queue_create: null output");
  f3d_queue q = new f3d_queue_s;
  hipError_t r = hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking);
  if (r != hipSuccess) {
    delete q;
    return f3d::hip_fail(r, "hipStreamCreateWithFlags", __FILE__, __LINE__);
  }
  *queue = q;
  return 0;
}

int f3d_queue_destroy(f3d_queue queue)
{
  if (!queue) return 0;
  (void)hipStreamSynchronize(queue->stream);
  (void)hipStreamDestroy(queue->stream);
  delete queue;
  return 0;
}

int f3d_queue_sync(f3d_queue queue)
{
  F3D_REQUIRE_READY("f3d_queue_sync");
  F3D_HIP(hipStreamSynchronize(stream_of(queue)));
  return 0;
}

int f3d_event_record_on(f3d_event ev, f3d_queue queue)
{
  F3D_REQUIRE_READY("f3d_event_record_on");
  if (!ev) return f3d::fail("f3d_event_record_on: null event");
  F3D_HIP(hipEventRecord(ev->ev, stream_of(queue)));
  return 0;
}

int f3d_queue_wait_event(f3d_queue queue, f3d_event ev)
{
  F3D_REQUIRE_READY("f3d_queue_wait_event");
  if (!ev) return f3d::fail("f3d_queue_wait_event: null event");
  F3D_HIP(hipStreamWaitEvent(stream_of(queue), ev->ev, 0));
  return 0;
}

int f3d_event_sync(f3d_event ev)
{
  F3D_HIP(hipEventSynchronize(ev->ev));
  return 0;
}

int f3d_event_elapsed_ms(float* ms, f3d_event start, f3d_event stop)
{
  F3D_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
  return 0;
}

int f3d_event_destroy(f3d_event ev)
{
  if (!ev) return 0;
  (void)hipEventDestroy(ev->ev);
  delete ev;
  return 0;
}

int f3d_stream_sync(void)
{
  F3D_REQUIRE_READY("f3d_stream_sync");
  F3D_HIP(hipStreamSynchronize(cur().stream));
  return 0;
}

int f3d_prof_enable(int enable)
{
  P.enabled = enable != 0;
  return 0;
}

int f3d_prof_select(unsigned kernel_mask)
{
  P.mask = kernel_mask;
  return 0;
}

int f3d_prof_reset(void)
{
  F3D_REQUIRE_READY("f3d_prof_reset");
  F3D_HIP(hipStreamSynchronize(cur().stream));
  for (auto& r : P.pending) {
    P.pool.push_back(r.start);
    P.pool.push_back(r.stop);
  }
  P.pending.clear();
  return 0;
}

int f3d_prof_read(int kernel, size_t min_voxels, double* total_ms, uint64_t* launches, double* total_voxels)
{
  F3D_REQUIRE_READY("f3d_prof_read");
  if (kernel < 0 || kernel >= F3D_K_COUNT) return f3d::fail("f3d_prof_read: bad kernel id %d", kernel);
  F3D_HIP(hipStreamSynchronize(cur().stream));
  double ms = 0, vox = 0;
  uint64_t n = 0;
  for (auto& r : P.pending) {
    if (r.kernel != kernel || r.voxels < min_voxels) continue;
    float t = 0;
    F3D_HIP(hipEventElapsedTime(&t, r.start, r.stop));
    ms += t;
    vox += static_cast<double>(r.voxels);
    ++n;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (total_voxels) *total_voxels = vox;
  return 0;
}

}  // extern "C"
