// TEST INFRASTRUCTURE ONLY -- a host-memory stand-in for libf3d_hip.so.
//
// Implements the C ABI of include/f3d.h on plain host memory with the ORACLE's kernels (oracle/f3d_oracle.c) as the compute.
// It exists so that the product's HOST code -- the drivers and operators of cuda-flow3d_amd/host (pyramid loop, slab planner
// and exchanges, out-of-core chunking, container pool, parameter bags) -- can run where there is no GPU: under
// AddressSanitizer / UndefinedBehaviorSanitizer (`make -C tests/cpu_device asan`) and in the `-m "not gpu"` tests, where a whole
// ComputeFlow of the product's drivers on this backend must equal the oracle's own whole-pipeline function bit for bit.
// Nothing in cuda-flow3d_amd/ knows about this file, and the product never loads it: the tests build it into a directory of its
// own under the library name the host library links against and point the package at that directory.
//
// "Device pointers" are host pointers; every launcher is synchronous; queues and events are bookkeeping only; RCCL entry points
// serve a single rank.  Launch semantics follow the device library where they are observable: the fused launches compute their
// first stage on the window widened by one plane, exactly like k_pair8 / k_sweep7 do.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "f3d.h"
#include "f3d_oracle.h"

namespace {

thread_local char g_error[512] = "";
bool g_ready = false;
f3d_size4 g_container = {0, 0, 0, 0};
float g_taps[51];
int g_tap_count = 0;
std::map<void*, size_t> g_allocs;   // base -> bytes
std::map<const void*, size_t> g_pinned;

int fail(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return 1;
}

template <typename T>
T* P(f3d_devptr p) { return reinterpret_cast<T*>(static_cast<uintptr_t>(p)); }

struct Geo {
  orc_geom g;
  int W, H, D;
};

bool make_geo(Geo* o, size_t w, size_t h, size_t d, const f3d_slab* slab, const char* who)
{
  const f3d_size4& c = g_container;
  if (c.pitch == 0 || c.height == 0) return fail("%s: f3d_set_container() has not been called", who), false;
  if (w == 0 || h == 0 || d == 0 || w > c.width || h > c.height || w * sizeof(float) > c.pitch)
    return fail("%s: level %zux%zux%zu does not fit the container %zux%zux%zu (pitch %zu B)", who, w, h, d, c.width, c.height, c.depth, c.pitch), false;
  o->W = static_cast<int>(w); o->H = static_cast<int>(h); o->D = static_cast<int>(d);
  o->g.Hc = static_cast<int>(c.height);
  o->g.pitch_f = static_cast<int>(c.pitch / sizeof(float));
  if (slab) {
    o->g.z_base = slab->z_base; o->g.z_lo = slab->z_lo; o->g.z_hi = slab->z_hi;
    if (slab->z_lo < 0 || slab->z_hi > o->D || slab->z_lo > slab->z_hi || slab->z_lo < slab->z_base ||
        static_cast<size_t>(slab->z_hi - slab->z_base) > c.depth)
      return fail("%s: slab planes [%d,%d) base %d outside level depth %d / container depth %zu", who, slab->z_lo, slab->z_hi,
                  slab->z_base, o->D, c.depth), false;
  } else {
    o->g.z_base = 0; o->g.z_lo = 0; o->g.z_hi = o->D;
    if (d > c.depth) return fail("%s: depth %zu exceeds container depth %zu", who, d, c.depth), false;
  }
  return true;
}

// a scratch container with the current geometry, NaN-poisoned
std::vector<float> scratch()
{
  const size_t n = g_container.pitch / sizeof(float) * g_container.height * g_container.depth;
  return std::vector<float>(n, std::nanf(""));
}

orc_geom widen(const Geo& o, int by)
{
  orc_geom g = o.g;
  g.z_lo = std::max(0, g.z_lo - by);
  g.z_hi = std::min(o.D, g.z_hi + by);
  return g;
}

struct Event { std::chrono::steady_clock::time_point t; };

}  // namespace

struct f3d_event_s { Event e; };
struct f3d_queue_s { int unused; };

extern "C" {

int f3d_init(int) { g_ready = true; return 0; }
int f3d_shutdown(void) { g_ready = false; return 0; }
int f3d_is_initialized(void) { return g_ready ? 1 : 0; }
int f3d_device_count(int* count) { if (count) *count = 1; return 0; }
int f3d_device_name(char* name, size_t capacity) { std::snprintf(name, capacity, "host-memory test backend (oracle kernels)"); return 0; }
int f3d_mem_info(size_t* free_bytes, size_t* total_bytes)
{
  const char* e = std::getenv("F3D_CPU_DEVICE_MEM_MB");
  const size_t total = (e ? static_cast<size_t>(std::atof(e)) : 4096) * 1024 * 1024;
  size_t used = 0;
  for (auto& a : g_allocs) used += a.second;
  *total_bytes = total;
  *free_bytes = used < total ? total - used : 0;
  return 0;
}
int f3d_lds_per_workgroup(int* bytes) { *bytes = 160 * 1024; return 0; }
const char* f3d_last_error(void) { return g_error; }

int f3d_alloc_pitched(f3d_devptr* ptr, size_t* pitch, size_t width_bytes, size_t rows)
{
  if (!ptr || !pitch || width_bytes == 0 || rows == 0) return fail("f3d_alloc_pitched: bad arguments");
  const size_t p = (width_bytes + 255) / 256 * 256;
  void* m = std::malloc(p * rows);
  if (!m) return fail("f3d_alloc_pitched: out of host memory");
  std::memset(m, 0xFF, p * rows);  // NaN poison: a stale read shows
  g_allocs[m] = p * rows;
  *ptr = static_cast<f3d_devptr>(reinterpret_cast<uintptr_t>(m));
  *pitch = p;
  return 0;
}
int f3d_free(f3d_devptr ptr)
{
  void* m = P<void>(ptr);
  auto it = g_allocs.find(m);
  if (it == g_allocs.end()) return fail("f3d_free: unknown pointer");
  g_allocs.erase(it);
  std::free(m);
  return 0;
}
int f3d_memset2d(f3d_devptr ptr, size_t pitch, int value, size_t width_bytes, size_t rows)
{
  char* b = P<char>(ptr);
  for (size_t r = 0; r < rows; ++r) std::memset(b + r * pitch, value, width_bytes);
  return 0;
}
int f3d_copy_planes_h2d(f3d_devptr dst, size_t dev_pitch, size_t dev_height, size_t dev_plane0, const float* src,
                        size_t src_row_floats, size_t src_rows, size_t width, size_t height, size_t depth)
{
  char* d = P<char>(dst);
  for (size_t z = 0; z < depth; ++z)
    for (size_t y = 0; y < height; ++y)
      std::memcpy(d + ((dev_plane0 + z) * dev_height + y) * dev_pitch, src + (z * src_rows + y) * src_row_floats, width * sizeof(float));
  return 0;
}
int f3d_copy_planes_d2h(float* dst, size_t dst_row_floats, size_t dst_rows, size_t width, size_t height, size_t depth,
                        f3d_devptr src, size_t dev_pitch, size_t dev_height, size_t dev_plane0)
{
  const char* s = P<const char>(src);
  for (size_t z = 0; z < depth; ++z)
    for (size_t y = 0; y < height; ++y)
      std::memcpy(dst + (z * dst_rows + y) * dst_row_floats, s + ((dev_plane0 + z) * dev_height + y) * dev_pitch, width * sizeof(float));
  return 0;
}
int f3d_copy3d_h2d(f3d_devptr dst, size_t dev_pitch, size_t dev_height, size_t dev_plane0, const float* src, size_t width,
                   size_t height, size_t depth)
{
  return f3d_copy_planes_h2d(dst, dev_pitch, dev_height, dev_plane0, src, width, height, width, height, depth);
}
int f3d_copy3d_d2h(float* dst, size_t width, size_t height, size_t depth, f3d_devptr src, size_t dev_pitch, size_t dev_height,
                   size_t dev_plane0)
{
  return f3d_copy_planes_d2h(dst, width, height, width, height, depth, src, dev_pitch, dev_height, dev_plane0);
}
int f3d_queue_create(f3d_queue* q) { *q = new f3d_queue_s{0}; return 0; }
int f3d_queue_destroy(f3d_queue q) { delete q; return 0; }
int f3d_queue_sync(f3d_queue) { return 0; }
int f3d_event_record_on(f3d_event ev, f3d_queue) { if (ev) ev->e.t = std::chrono::steady_clock::now(); return 0; }
int f3d_queue_wait_event(f3d_queue, f3d_event) { return 0; }
int f3d_copy_planes_h2d_on(f3d_queue, f3d_devptr dst, size_t dev_pitch, size_t dev_height, size_t dev_plane0, const float* src,
                           size_t src_row_floats, size_t src_rows, size_t width, size_t height, size_t depth)
{
  return f3d_copy_planes_h2d(dst, dev_pitch, dev_height, dev_plane0, src, src_row_floats, src_rows, width, height, depth);
}
int f3d_copy_planes_d2h_on(f3d_queue, float* dst, size_t dst_row_floats, size_t dst_rows, size_t width, size_t height, size_t depth,
                           f3d_devptr src, size_t dev_pitch, size_t dev_height, size_t dev_plane0)
{
  return f3d_copy_planes_d2h(dst, dst_row_floats, dst_rows, width, height, depth, src, dev_pitch, dev_height, dev_plane0);
}
int f3d_copy_rect_d2d(f3d_devptr dst, size_t dst_pitch, size_t dst_rows, size_t dst_plane0, f3d_devptr src, size_t src_pitch,
                      size_t src_rows, size_t src_plane0, size_t width, size_t height, size_t depth)
{
  char* d = P<char>(dst);
  const char* s = P<const char>(src);
  for (size_t z = 0; z < depth; ++z)
    for (size_t y = 0; y < height; ++y)
      std::memcpy(d + ((dst_plane0 + z) * dst_rows + y) * dst_pitch, s + ((src_plane0 + z) * src_rows + y) * src_pitch, width * sizeof(float));
  return 0;
}
// (the out-of-core driver registers its scratch volumes from a helper thread while the main thread works)
static std::mutex g_pinned_lock;
int f3d_host_register(void* ptr, size_t bytes)
{
  std::lock_guard<std::mutex> hold(g_pinned_lock);
  g_pinned[ptr] = bytes;
  return 0;
}
int f3d_host_unregister(void* ptr)
{
  std::lock_guard<std::mutex> hold(g_pinned_lock);
  return g_pinned.erase(ptr) ? 0 : fail("f3d_host_unregister: not registered");
}
int f3d_host_is_pinned(const void* ptr, int* yes)
{
  std::lock_guard<std::mutex> hold(g_pinned_lock);
  *yes = 0;
  for (auto& p : g_pinned)
    if (static_cast<const char*>(ptr) >= static_cast<const char*>(p.first) &&
        static_cast<const char*>(ptr) < static_cast<const char*>(p.first) + p.second)
      *yes = 1;
  return 0;
}
int f3d_copy_d2d(f3d_devptr dst, f3d_devptr src, size_t bytes) { std::memcpy(P<void>(dst), P<const void>(src), bytes); return 0; }
int f3d_set_container(const f3d_size4* c)
{
  if (!c || c->width == 0 || c->pitch < c->width * sizeof(float)) return fail("f3d_set_container: bad container");
  g_container = *c;
  return 0;
}
int f3d_get_container(f3d_size4* c) { *c = g_container; return 0; }

int f3d_event_create(f3d_event* ev) { *ev = new f3d_event_s(); return 0; }
int f3d_event_record(f3d_event ev) { ev->e.t = std::chrono::steady_clock::now(); return 0; }
int f3d_event_sync(f3d_event) { return 0; }
int f3d_event_elapsed_ms(float* ms, f3d_event a, f3d_event b)
{
  *ms = std::chrono::duration<float, std::milli>(b->e.t - a->e.t).count();
  return 0;
}
int f3d_event_destroy(f3d_event ev) { delete ev; return 0; }
int f3d_stream_sync(void) { return 0; }

int f3d_phi_ksi(f3d_devptr f0, f3d_devptr f1, f3d_devptr u, f3d_devptr v, f3d_devptr w, f3d_devptr du, f3d_devptr dv, f3d_devptr dw,
                size_t width, size_t height, size_t depth, float hx, float hy, float hz, float eps_s, float eps_d, f3d_devptr phi,
                f3d_devptr ksi, const f3d_slab* slab)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_phi_ksi")) return 1;
  if (o.g.z_lo == o.g.z_hi) return 0;
  orc_phi_ksi(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), P<float>(du), P<float>(dv), P<float>(dw), o.W, o.H,
              o.D, hx, hy, hz, eps_s, eps_d, P<float>(phi), P<float>(ksi), &o.g);
  return 0;
}
int f3d_phi_ksi_zones(f3d_devptr f0, f3d_devptr f1, f3d_devptr u, f3d_devptr v, f3d_devptr w, f3d_devptr du, f3d_devptr dv, f3d_devptr dw,
                      size_t width, size_t height, size_t depth, float hx, float hy, float hz, float eps_s, float eps_d, f3d_devptr phi,
                      f3d_devptr ksi, const f3d_slab* zone_a, const f3d_slab* zone_b)
{
  if (!zone_a || !zone_b || zone_a->z_base != zone_b->z_base) return fail("f3d_phi_ksi_zones: two windows of one container are required");
  const int first = f3d_phi_ksi(f0, f1, u, v, w, du, dv, dw, width, height, depth, hx, hy, hz, eps_s, eps_d, phi, ksi, zone_a);
  return first ? first : f3d_phi_ksi(f0, f1, u, v, w, du, dv, dw, width, height, depth, hx, hy, hz, eps_s, eps_d, phi, ksi, zone_b);
}
int f3d_solve_sweep(f3d_devptr f0, f3d_devptr f1, f3d_devptr u, f3d_devptr v, f3d_devptr w, f3d_devptr du, f3d_devptr dv, f3d_devptr dw,
                    f3d_devptr phi, f3d_devptr ksi, size_t width, size_t height, size_t depth, float hx, float hy, float hz, float alpha,
                    f3d_devptr tdu, f3d_devptr tdv, f3d_devptr tdw, const f3d_slab* slab)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_solve_sweep")) return 1;
  if (o.g.z_lo == o.g.z_hi) return 0;
  orc_solve_sweep(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), P<float>(du), P<float>(dv), P<float>(dw),
                  P<float>(phi), P<float>(ksi), o.W, o.H, o.D, hx, hy, hz, alpha, P<float>(tdu), P<float>(tdv), P<float>(tdw), &o.g);
  return 0;
}
int f3d_solve_sweep2(f3d_devptr f0, f3d_devptr f1, f3d_devptr u, f3d_devptr v, f3d_devptr w, f3d_devptr du, f3d_devptr dv, f3d_devptr dw,
                     f3d_devptr phi, f3d_devptr ksi, size_t width, size_t height, size_t depth, float hx, float hy, float hz, float alpha,
                     f3d_devptr tdu, f3d_devptr tdv, f3d_devptr tdw, const f3d_slab* slab)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_solve_sweep2")) return 1;
  if (o.g.z_lo == o.g.z_hi) return 0;
  std::vector<float> a = scratch(), b = scratch(), c = scratch();
  const orc_geom wide = widen(o, 1);  // the first sweep on one plane more on either side
  orc_solve_sweep(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), P<float>(du), P<float>(dv), P<float>(dw),
                  P<float>(phi), P<float>(ksi), o.W, o.H, o.D, hx, hy, hz, alpha, a.data(), b.data(), c.data(), &wide);
  orc_solve_sweep(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), a.data(), b.data(), c.data(), P<float>(phi),
                  P<float>(ksi), o.W, o.H, o.D, hx, hy, hz, alpha, P<float>(tdu), P<float>(tdv), P<float>(tdw), &o.g);
  return 0;
}
int f3d_fused_launches_march_along_y(size_t, size_t, size_t) { return 0; }
int f3d_solve_sweep3(f3d_devptr f0, f3d_devptr f1, f3d_devptr u, f3d_devptr v, f3d_devptr w, f3d_devptr du, f3d_devptr dv, f3d_devptr dw,
                     f3d_devptr phi, f3d_devptr ksi, size_t width, size_t height, size_t depth, float hx, float hy, float hz, float alpha,
                     f3d_devptr tdu, f3d_devptr tdv, f3d_devptr tdw, const f3d_slab* slab)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_solve_sweep3")) return 1;
  if (o.g.z_lo == o.g.z_hi) return 0;
  std::vector<float> a = scratch(), b = scratch(), c = scratch(), d = scratch(), e = scratch(), f = scratch();
  const orc_geom wide2 = widen(o, 2), wide1 = widen(o, 1);  // the first sweep on two planes more on either side, the second on one
  orc_solve_sweep(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), P<float>(du), P<float>(dv), P<float>(dw),
                  P<float>(phi), P<float>(ksi), o.W, o.H, o.D, hx, hy, hz, alpha, a.data(), b.data(), c.data(), &wide2);
  orc_solve_sweep(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), a.data(), b.data(), c.data(), P<float>(phi),
                  P<float>(ksi), o.W, o.H, o.D, hx, hy, hz, alpha, d.data(), e.data(), f.data(), &wide1);
  orc_solve_sweep(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), d.data(), e.data(), f.data(), P<float>(phi),
                  P<float>(ksi), o.W, o.H, o.D, hx, hy, hz, alpha, P<float>(tdu), P<float>(tdv), P<float>(tdw), &o.g);
  return 0;
}
int f3d_solve_sweep2_phi_ksi(f3d_devptr f0, f3d_devptr f1, f3d_devptr u, f3d_devptr v, f3d_devptr w, f3d_devptr du, f3d_devptr dv,
                             f3d_devptr dw, f3d_devptr phi, f3d_devptr ksi, size_t width, size_t height, size_t depth, float hx, float hy,
                             float hz, float alpha, float eps_s, float eps_d, f3d_devptr tdu, f3d_devptr tdv, f3d_devptr tdw,
                             f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_solve_sweep2_phi_ksi")) return 1;
  if (phi_next == phi || ksi_next == ksi || phi_next == ksi || ksi_next == phi)
    return fail("f3d_solve_sweep2_phi_ksi: phi_next / ksi_next must not alias phi / ksi");
  if (o.g.z_lo == o.g.z_hi) return 0;
  std::vector<float> a = scratch(), b = scratch(), c = scratch(), d = scratch(), e = scratch(), f = scratch();
  const orc_geom wide2 = widen(o, 2), wide1 = widen(o, 1);
  orc_solve_sweep(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), P<float>(du), P<float>(dv), P<float>(dw),
                  P<float>(phi), P<float>(ksi), o.W, o.H, o.D, hx, hy, hz, alpha, a.data(), b.data(), c.data(), &wide2);
  orc_solve_sweep(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), a.data(), b.data(), c.data(), P<float>(phi),
                  P<float>(ksi), o.W, o.H, o.D, hx, hy, hz, alpha, d.data(), e.data(), f.data(), &wide1);
  orc_phi_ksi(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), d.data(), e.data(), f.data(), o.W, o.H, o.D, hx, hy, hz,
              eps_s, eps_d, P<float>(phi_next), P<float>(ksi_next), &o.g);
  const size_t plane = static_cast<size_t>(o.g.Hc) * o.g.pitch_f;
  float* outs[3] = {P<float>(tdu), P<float>(tdv), P<float>(tdw)};
  const float* ins[3] = {d.data(), e.data(), f.data()};
  for (int k = 0; k < 3; ++k)
    for (int z = o.g.z_lo; z < o.g.z_hi; ++z)
      for (int y = 0; y < o.H; ++y)
        std::memcpy(outs[k] + (z - o.g.z_base) * plane + static_cast<size_t>(y) * o.g.pitch_f,
                    ins[k] + (z - o.g.z_base) * plane + static_cast<size_t>(y) * o.g.pitch_f, o.W * sizeof(float));
  return 0;
}
int f3d_solve_sweep_phi_ksi_edges(f3d_devptr f0, f3d_devptr f1, f3d_devptr u, f3d_devptr v, f3d_devptr w, f3d_devptr du, f3d_devptr dv,
                                  f3d_devptr dw, f3d_devptr phi, f3d_devptr ksi, size_t width, size_t height, size_t depth, float hx,
                                  float hy, float hz, float alpha, float eps_s, float eps_d, f3d_devptr tdu, f3d_devptr tdv,
                                  f3d_devptr tdw, f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab, int keep_below,
                                  int keep_above)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_solve_sweep_phi_ksi")) return 1;
  if (phi_next == phi || ksi_next == ksi || phi_next == ksi || ksi_next == phi)
    return fail("f3d_solve_sweep_phi_ksi: phi_next / ksi_next must not alias phi / ksi");
  if (o.g.z_lo == o.g.z_hi) return 0;
  std::vector<float> a = scratch(), b = scratch(), c = scratch();
  const orc_geom wide = widen(o, 1);
  orc_solve_sweep(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), P<float>(du), P<float>(dv), P<float>(dw),
                  P<float>(phi), P<float>(ksi), o.W, o.H, o.D, hx, hy, hz, alpha, a.data(), b.data(), c.data(), &wide);
  orc_phi_ksi(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), a.data(), b.data(), c.data(), o.W, o.H, o.D, hx, hy, hz,
              eps_s, eps_d, P<float>(phi_next), P<float>(ksi_next), &o.g);
  // the sweep's result for the planes of the window
  const size_t plane = static_cast<size_t>(o.g.Hc) * o.g.pitch_f;
  float* outs[3] = {P<float>(tdu), P<float>(tdv), P<float>(tdw)};
  const float* ins[3] = {a.data(), b.data(), c.data()};
  for (int k = 0; k < 3; ++k)
    for (int z = o.g.z_lo - (keep_below && o.g.z_lo > 0 ? 1 : 0); z < o.g.z_hi + (keep_above && o.g.z_hi < o.D ? 1 : 0); ++z)
      for (int y = 0; y < o.H; ++y)
        std::memcpy(outs[k] + (z - o.g.z_base) * plane + static_cast<size_t>(y) * o.g.pitch_f,
                    ins[k] + (z - o.g.z_base) * plane + static_cast<size_t>(y) * o.g.pitch_f, o.W * sizeof(float));
  return 0;
}
int f3d_solve_sweep_phi_ksi(f3d_devptr f0, f3d_devptr f1, f3d_devptr u, f3d_devptr v, f3d_devptr w, f3d_devptr du, f3d_devptr dv,
                            f3d_devptr dw, f3d_devptr phi, f3d_devptr ksi, size_t width, size_t height, size_t depth, float hx, float hy,
                            float hz, float alpha, float eps_s, float eps_d, f3d_devptr tdu, f3d_devptr tdv, f3d_devptr tdw,
                            f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab)
{
  return f3d_solve_sweep_phi_ksi_edges(f0, f1, u, v, w, du, dv, dw, phi, ksi, width, height, depth, hx, hy, hz, alpha, eps_s, eps_d, tdu, tdv,
                                       tdw, phi_next, ksi_next, slab, 0, 0);
}
// Frame derivatives: computed for real (same expressions as the device kernel), and the frames they came from are remembered by the
// address of fx so that the _fd launchers can hand the oracle's frame-based kernels what they need.
static std::map<f3d_devptr, std::pair<f3d_devptr, f3d_devptr>> g_frames_of;
int f3d_frame_derivatives(f3d_devptr frame_0, f3d_devptr frame_1, size_t width, size_t height, size_t depth, float hx, float hy, float hz,
                          f3d_devptr fx, f3d_devptr fy, f3d_devptr fz, f3d_devptr ft, const f3d_slab* slab)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_frame_derivatives")) return 1;
  const float *a = P<float>(frame_0), *b = P<float>(frame_1);
  const size_t plane = static_cast<size_t>(o.g.Hc) * o.g.pitch_f;
  auto at = [&](int x, int y, int z) { return (z - o.g.z_base) * plane + static_cast<size_t>(y) * o.g.pitch_f + x; };
  auto mir = [](int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - i - 2 : i); };
  for (int z = o.g.z_lo; z < o.g.z_hi; ++z)
    for (int y = 0; y < o.H; ++y)
      for (int x = 0; x < o.W; ++x) {
        const size_t c = at(x, y, z);
        const size_t xm = at(mir(x - 1, o.W), y, z), xp = at(mir(x + 1, o.W), y, z);
        const size_t ym = at(x, mir(y - 1, o.H), z), yp = at(x, mir(y + 1, o.H), z);
        const size_t zm = at(x, y, mir(z - 1, o.D)), zp = at(x, y, mir(z + 1, o.D));
        P<float>(fx)[c] = (a[xp] - a[xm] + b[xp] - b[xm]) / (4.f * hx);
        P<float>(fy)[c] = (a[yp] - a[ym] + b[yp] - b[ym]) / (4.f * hy);
        P<float>(fz)[c] = (a[zp] - a[zm] + b[zp] - b[zm]) / (4.f * hz);
        P<float>(ft)[c] = b[c] - a[c];
      }
  g_frames_of[fx] = {frame_0, frame_1};
  return 0;
}
int f3d_solve_sweep2_fd(f3d_devptr fx, f3d_devptr, f3d_devptr, f3d_devptr, f3d_devptr u, f3d_devptr v, f3d_devptr w, f3d_devptr du, f3d_devptr dv,
                        f3d_devptr dw, f3d_devptr phi, f3d_devptr ksi, size_t width, size_t height, size_t depth, float hx, float hy, float hz,
                        float alpha, f3d_devptr tdu, f3d_devptr tdv, f3d_devptr tdw, const f3d_slab* slab)
{
  auto it = g_frames_of.find(fx);
  if (it == g_frames_of.end()) return fail("f3d_solve_sweep2_fd: these derivatives were not made by f3d_frame_derivatives");
  return f3d_solve_sweep2(it->second.first, it->second.second, u, v, w, du, dv, dw, phi, ksi, width, height, depth, hx, hy, hz, alpha, tdu, tdv,
                          tdw, slab);
}
int f3d_solve_sweep_phi_ksi_fd(f3d_devptr fx, f3d_devptr, f3d_devptr, f3d_devptr, f3d_devptr u, f3d_devptr v, f3d_devptr w, f3d_devptr du,
                               f3d_devptr dv, f3d_devptr dw, f3d_devptr phi, f3d_devptr ksi, size_t width, size_t height, size_t depth, float hx,
                               float hy, float hz, float alpha, float eps_s, float eps_d, f3d_devptr tdu, f3d_devptr tdv, f3d_devptr tdw,
                               f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab)
{
  auto it = g_frames_of.find(fx);
  if (it == g_frames_of.end()) return fail("f3d_solve_sweep_phi_ksi_fd: these derivatives were not made by f3d_frame_derivatives");
  return f3d_solve_sweep_phi_ksi(it->second.first, it->second.second, u, v, w, du, dv, dw, phi, ksi, width, height, depth, hx, hy, hz, alpha,
                                 eps_s, eps_d, tdu, tdv, tdw, phi_next, ksi_next, slab);
}
int f3d_solve_sweep_phi_ksi_edges_fd(f3d_devptr fx, f3d_devptr, f3d_devptr, f3d_devptr, f3d_devptr u, f3d_devptr v, f3d_devptr w, f3d_devptr du,
                                     f3d_devptr dv, f3d_devptr dw, f3d_devptr phi, f3d_devptr ksi, size_t width, size_t height, size_t depth,
                                     float hx, float hy, float hz, float alpha, float eps_s, float eps_d, f3d_devptr tdu, f3d_devptr tdv,
                                     f3d_devptr tdw, f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab, int keep_below,
                                     int keep_above)
{
  auto it = g_frames_of.find(fx);
  if (it == g_frames_of.end()) return fail("f3d_solve_sweep_phi_ksi_edges_fd: these derivatives were not made by f3d_frame_derivatives");
  return f3d_solve_sweep_phi_ksi_edges(it->second.first, it->second.second, u, v, w, du, dv, dw, phi, ksi, width, height, depth, hx, hy, hz,
                                       alpha, eps_s, eps_d, tdu, tdv, tdw, phi_next, ksi_next, slab, keep_below, keep_above);
}
int f3d_warp(f3d_devptr f0, f3d_devptr f1, f3d_devptr u, f3d_devptr v, f3d_devptr w, size_t width, size_t height, size_t depth, float hx,
             float hy, float hz, f3d_devptr output, const f3d_slab* slab)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_warp")) return 1;
  if (output == f1 || output == f0) return fail("f3d_warp: input buffer cannot serve as output buffer");
  orc_warp(P<float>(f0), P<float>(f1), P<float>(u), P<float>(v), P<float>(w), o.W, o.H, o.D, hx, hy, hz, P<float>(output), &o.g);
  return 0;
}
static int resample(int axis, f3d_devptr in, f3d_devptr out, size_t ow, size_t oh, size_t od, size_t in_n, const f3d_slab* slab_in,
                    const f3d_slab* slab, const char* who)
{
  if (in == out) return fail("%s: input buffer cannot serve as output buffer", who);
  Geo o;
  // the output extent may exceed the level being written along the axes not yet resampled; only W/H of the container bound it
  if (!make_geo(&o, std::min(ow, g_container.width), std::min(oh, g_container.height), od, slab, who)) return 1;
  orc_geom gin = o.g;
  if (axis == 2) {
    gin.z_base = slab_in ? slab_in->z_base : 0;
    gin.z_lo = slab_in ? slab_in->z_lo : 0;
    gin.z_hi = slab_in ? slab_in->z_hi : static_cast<int>(in_n);
  }
  orc_resample_axis(P<float>(in), P<float>(out), static_cast<int>(ow), static_cast<int>(oh), static_cast<int>(od), static_cast<int>(in_n),
                    axis, &gin, &o.g);
  return 0;
}
int f3d_resample_x(f3d_devptr in, f3d_devptr out, size_t ow, size_t oh, size_t od, size_t in_w, const f3d_slab* slab)
{
  return resample(0, in, out, ow, oh, od, in_w, nullptr, slab, "f3d_resample_x");
}
int f3d_resample_y(f3d_devptr in, f3d_devptr out, size_t ow, size_t oh, size_t od, size_t in_h, const f3d_slab* slab)
{
  return resample(1, in, out, ow, oh, od, in_h, nullptr, slab, "f3d_resample_y");
}
int f3d_resample_z(f3d_devptr in, f3d_devptr out, size_t ow, size_t oh, size_t od, size_t in_d, const f3d_slab* slab_in, const f3d_slab* slab)
{
  return resample(2, in, out, ow, oh, od, in_d, slab_in, slab, "f3d_resample_z");
}
int f3d_add(f3d_devptr a, f3d_devptr b, size_t width, size_t height, size_t depth, const f3d_slab* slab)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_add")) return 1;
  orc_add(P<float>(a), P<float>(b), o.W, o.H, o.D, &o.g);
  return 0;
}
int f3d_median(f3d_devptr in, size_t width, size_t height, size_t depth, size_t radius, f3d_devptr out, const f3d_slab* slab)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_median")) return 1;
  if (in == out) return fail("f3d_median: input buffer cannot serve as output buffer");
  if (radius != 3 && radius != 5 && radius != 7) return fail("f3d_median: window %zu not in {3, 5, 7}", radius);
  orc_median(P<float>(in), P<float>(out), o.W, o.H, o.D, static_cast<int>(radius), &o.g);
  return 0;
}
// the batched entries: the single-volume ones, one after the other
static bool batch_ok(const void* a, const void* b, size_t count, const char* who)
{
  if (!a || !b) return fail("%s: null argument", who), false;
  if (count == 0 || count > 3) return fail("%s: %zu volumes (one launch takes 1 .. 3)", who, count), false;
  return true;
}
int f3d_resample_x_n(const f3d_devptr* in, const f3d_devptr* out, size_t count, size_t ow, size_t oh, size_t od, size_t in_w,
                     const f3d_slab* slab)
{
  if (!batch_ok(in, out, count, "f3d_resample_x_n")) return 1;
  for (size_t i = 0; i < count; ++i)
    if (int e = resample(0, in[i], out[i], ow, oh, od, in_w, nullptr, slab, "f3d_resample_x_n")) return e;
  return 0;
}
int f3d_resample_y_n(const f3d_devptr* in, const f3d_devptr* out, size_t count, size_t ow, size_t oh, size_t od, size_t in_h,
                     const f3d_slab* slab)
{
  if (!batch_ok(in, out, count, "f3d_resample_y_n")) return 1;
  for (size_t i = 0; i < count; ++i)
    if (int e = resample(1, in[i], out[i], ow, oh, od, in_h, nullptr, slab, "f3d_resample_y_n")) return e;
  return 0;
}
int f3d_resample_z_n(const f3d_devptr* in, const f3d_devptr* out, size_t count, size_t ow, size_t oh, size_t od, size_t in_d,
                     const f3d_slab* slab_in, const f3d_slab* slab)
{
  if (!batch_ok(in, out, count, "f3d_resample_z_n")) return 1;
  for (size_t i = 0; i < count; ++i)
    if (int e = resample(2, in[i], out[i], ow, oh, od, in_d, slab_in, slab, "f3d_resample_z_n")) return e;
  return 0;
}
int f3d_add_n(const f3d_devptr* a, const f3d_devptr* b, size_t count, size_t width, size_t height, size_t depth, const f3d_slab* slab)
{
  if (!batch_ok(a, b, count, "f3d_add_n")) return 1;
  for (size_t i = 0; i < count; ++i)
    if (int e = f3d_add(a[i], b[i], width, height, depth, slab)) return e;
  return 0;
}
int f3d_median_n(const f3d_devptr* in, size_t count, size_t width, size_t height, size_t depth, size_t radius, const f3d_devptr* out,
                 const f3d_slab* slab)
{
  if (!batch_ok(in, out, count, "f3d_median_n")) return 1;
  for (size_t i = 0; i < count; ++i)
    for (size_t j = 0; j < count; ++j)
      if (in[i] == out[j]) return fail("f3d_median_n: input buffer cannot serve as output buffer");
  for (size_t i = 0; i < count; ++i)
    if (int e = f3d_median(in[i], width, height, depth, radius, out[i], slab)) return e;
  return 0;
}
int f3d_clear_box_n(const f3d_devptr* volumes, size_t count, size_t width, size_t height, size_t depth, const f3d_slab* slab)
{
  if (!batch_ok(volumes, volumes, count, "f3d_clear_box_n")) return 1;
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_clear_box_n")) return 1;
  for (size_t i = 0; i < count; ++i) {
    float* p = P<float>(volumes[i]);
    for (int z = o.g.z_lo; z < o.g.z_hi; ++z)
      for (int y = 0; y < o.H; ++y)
        std::memset(p + (static_cast<size_t>(z - o.g.z_base) * o.g.Hc + y) * o.g.pitch_f, 0, static_cast<size_t>(o.W) * sizeof(float));
  }
  return 0;
}
int f3d_set_conv_taps(const float* taps, size_t count)
{
  if (!taps || count == 0 || count > 51 || count % 2 == 0) return fail("f3d_set_conv_taps: bad tap count %zu", count);
  std::memcpy(g_taps, taps, count * sizeof(float));
  g_tap_count = static_cast<int>(count);
  return 0;
}
static int conv(int axis, f3d_devptr dst, f3d_devptr src, size_t w, size_t h, size_t d, size_t radius, const f3d_slab* slab, const char* who)
{
  if (dst == src) return fail("%s: input buffer cannot serve as output buffer", who);
  if (g_tap_count != static_cast<int>(2 * radius + 1)) return fail("%s: radius %zu does not match the %d taps", who, radius, g_tap_count);
  Geo o;
  if (!make_geo(&o, w, h, d, slab, who)) return 1;
  orc_conv_axis(P<float>(dst), P<float>(src), o.W, o.H, o.D, static_cast<int>(radius), g_taps, axis, &o.g);
  return 0;
}
int f3d_conv_rows(f3d_devptr dst, f3d_devptr src, size_t w, size_t h, size_t d, size_t r, const f3d_slab* s) { return conv(0, dst, src, w, h, d, r, s, "f3d_conv_rows"); }
int f3d_conv_cols(f3d_devptr dst, f3d_devptr src, size_t w, size_t h, size_t d, size_t r, const f3d_slab* s) { return conv(1, dst, src, w, h, d, r, s, "f3d_conv_cols"); }
int f3d_conv_slices(f3d_devptr dst, f3d_devptr src, size_t w, size_t h, size_t d, size_t r, const f3d_slab* s) { return conv(2, dst, src, w, h, d, r, s, "f3d_conv_slices"); }
int f3d_conv_rows_cols(f3d_devptr dst, f3d_devptr src, size_t w, size_t h, size_t d, size_t r, const f3d_slab* s)
{
  if (dst == src) return fail("f3d_conv_rows_cols: input buffer cannot serve as output buffer");
  std::vector<float> tmp = scratch();
  const f3d_devptr t = static_cast<f3d_devptr>(reinterpret_cast<uintptr_t>(tmp.data()));
  return conv(0, t, src, w, h, d, r, s, "f3d_conv_rows_cols") || conv(1, dst, t, w, h, d, r, s, "f3d_conv_rows_cols");
}

int f3d_range_push(const char*) { return 0; }
int f3d_range_pop(void) { return 0; }
int f3d_prof_enable(int) { return 0; }
int f3d_prof_reset(void) { return 0; }
int f3d_prof_select(unsigned) { return 0; }
int f3d_prof_read(int, size_t, double* ms, uint64_t* n, double* vox) { *ms = 0; *n = 0; *vox = 0; return 0; }

// lanes: the host-memory backend is synchronous and single-threaded; a private lane is the default one
struct f3d_lane_s { int unused; };
int f3d_lane_create(f3d_lane* lane) { *lane = new f3d_lane_s(); return 0; }
int f3d_lane_make_current(f3d_lane) { return 0; }
int f3d_lane_is_private(void) { return 0; }
int f3d_lane_get_current(f3d_lane* lane) { *lane = nullptr; return 0; }
int f3d_lane_destroy(f3d_lane lane) { delete lane; return 0; }
int f3d_crash_maps_enable(const char*) { return 0; }
int f3d_selftest_weights(unsigned, unsigned, unsigned long long* checked, unsigned long long* excluded, unsigned long long* mismatches, unsigned* first)
{
  if (checked) *checked = 0;   // the host-memory backend computes the weights with the IEEE expression itself: nothing to check
  if (excluded) *excluded = 0;
  if (mismatches) *mismatches = 0;
  if (first) *first = 0;
  return 0;
}
int f3d_comm_unique_id(void* id128) { std::memset(id128, 0, 128); return 0; }
int f3d_comm_init(const void*, int rank, int n_ranks) { return (rank == 0 && n_ranks == 1) ? 0 : fail("the host-memory backend serves one rank"); }
int f3d_comm_destroy(void) { return 0; }
int f3d_comm_rank(int* rank, int* n_ranks) { if (rank) *rank = 0; if (n_ranks) *n_ranks = 1; return 0; }
int f3d_comm_info(int* backend, int* comm_ranks, int* comm_rank, int* comm_device, unsigned long long* sent_bytes, unsigned long long* exchanges)
{
  if (backend) *backend = 0;
  if (comm_ranks) *comm_ranks = 0;
  if (comm_rank) *comm_rank = -1;
  if (comm_device) *comm_device = -1;
  if (sent_bytes) *sent_bytes = 0;
  if (exchanges) *exchanges = 0;
  return 0;
}
static void plane_copy(float* field, int plane0, int count, size_t width, size_t height, float* staging, bool pack)
{
  const size_t pitch_f = g_container.pitch / sizeof(float);
  for (int z = 0; z < count; ++z)
    for (size_t y = 0; y < height; ++y) {
      float* c = field + (static_cast<size_t>(plane0 + z) * g_container.height + y) * pitch_f;
      float* s = staging + (static_cast<size_t>(z) * height + y) * width;
      std::memcpy(pack ? s : c, pack ? c : s, width * sizeof(float));
    }
}
int f3d_pack_planes(f3d_devptr field, int plane0, int count, size_t width, size_t height, f3d_devptr staging, size_t offset_floats)
{
  plane_copy(P<float>(field), plane0, count, width, height, P<float>(staging) + offset_floats, true);
  return 0;
}
int f3d_unpack_planes(f3d_devptr field, int plane0, int count, size_t width, size_t height, f3d_devptr staging, size_t offset_floats)
{
  plane_copy(P<float>(field), plane0, count, width, height, P<float>(staging) + offset_floats, false);
  return 0;
}
int f3d_pack_segments(const f3d_devptr* fields, const int* plane0, const int* count, const size_t* off, int n, size_t width, size_t height,
                      f3d_devptr staging)
{
  for (int i = 0; i < n; ++i) plane_copy(P<float>(fields[i]), plane0[i], count[i], width, height, P<float>(staging) + off[i], true);
  return 0;
}
int f3d_unpack_segments(const f3d_devptr* fields, const int* plane0, const int* count, const size_t* off, int n, size_t width, size_t height,
                        f3d_devptr staging)
{
  for (int i = 0; i < n; ++i) plane_copy(P<float>(fields[i]), plane0[i], count[i], width, height, P<float>(staging) + off[i], false);
  return 0;
}
int f3d_copy_planes(f3d_devptr dst, int dst_plane0, f3d_devptr src, int src_plane0, int count, size_t width, size_t height);
int f3d_copy_plane_segments(const f3d_devptr* dst, const int* dst_plane0, const f3d_devptr* src, const int* src_plane0, const int* count,
                            int n_segments, size_t width, size_t height)
{
  // NOTE: segments may alias (a rank's planes are another rank's source): the device kernel reads and writes disjoint planes, and so
  // does this loop as long as every source plane is an OWNED plane and every destination a halo plane, which is what the driver asks for
  for (int i = 0; i < n_segments; ++i)
    if (f3d_copy_planes(dst[i], dst_plane0[i], src[i], src_plane0[i], count[i], width, height)) return 1;
  return 0;
}
int f3d_copy_planes(f3d_devptr dst, int dst_plane0, f3d_devptr src, int src_plane0, int count, size_t width, size_t height)
{
  const size_t pitch_f = g_container.pitch / sizeof(float);
  if (dst_plane0 < 0 || src_plane0 < 0 || count < 0 || static_cast<size_t>(dst_plane0 + count) > g_container.depth ||
      static_cast<size_t>(src_plane0 + count) > g_container.depth)
    return fail("f3d_copy_planes: planes outside the container");
  for (int z = 0; z < count; ++z)
    for (size_t y = 0; y < height; ++y)
      std::memcpy(P<float>(dst) + (static_cast<size_t>(dst_plane0 + z) * g_container.height + y) * pitch_f,
                  P<float>(src) + (static_cast<size_t>(src_plane0 + z) * g_container.height + y) * pitch_f, width * sizeof(float));
  return 0;
}
int f3d_comm_sendrecv(f3d_devptr, const size_t*, const size_t*, f3d_devptr, const size_t*, const size_t*, const int*, int n)
{
  return n == 0 ? 0 : fail("the host-memory backend has no peers");
}
int f3d_comm_sendrecv_begin(f3d_devptr a, const size_t* b, const size_t* c, f3d_devptr d, const size_t* e, const size_t* f, const int* g, int n)
{
  return f3d_comm_sendrecv(a, b, c, d, e, f, g, n);
}
int f3d_comm_sendrecv_end(void) { return 0; }
int f3d_comm_allreduce_max_f32(float*) { return 0; }
int f3d_comm_timing(int) { return 0; }
int f3d_comm_mark(int, int) { return 0; }
int f3d_comm_timing_read(int, double* us, unsigned long long* n, double* mn, double* mx, unsigned long long* bytes)
{
  if (us) *us = 0; if (n) *n = 0; if (mn) *mn = 0; if (mx) *mx = 0; if (bytes) *bytes = 0;
  return 0;
}
int f3d_abs_max(f3d_devptr field, size_t width, size_t height, size_t depth, const f3d_slab* slab, float* result)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_abs_max")) return 1;
  const size_t plane = static_cast<size_t>(o.g.Hc) * o.g.pitch_f;
  float m = 0.f;
  for (int z = o.g.z_lo; z < o.g.z_hi; ++z)
    for (int y = 0; y < o.H; ++y)
      for (int x = 0; x < o.W; ++x) {
        const float v = std::fabs(P<float>(field)[(z - o.g.z_base) * plane + static_cast<size_t>(y) * o.g.pitch_f + x]);
        if (v < INFINITY) m = std::max(m, v);
      }
  *result = m;
  return 0;
}
int f3d_flow_stats(f3d_devptr u, f3d_devptr v, f3d_devptr w, size_t width, size_t height, size_t depth, const f3d_slab* slab, float* mn,
                   float* mx, double* sum)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_flow_stats")) return 1;
  float avg;
  orc_flow_stats(P<float>(u), P<float>(v), P<float>(w), o.W, o.H, o.D, &o.g, mn, mx, &avg, sum);
  // the oracle starts its maximum at numeric_limits<float>::min() like the reference's host loop; the device starts at 0
  if (*mx == 1.175494351e-38f) *mx = 0.f;
  return 0;
}
int f3d_residual_stats(f3d_devptr f0, f3d_devptr fw, size_t width, size_t height, size_t depth, const f3d_slab* slab, double* ssq,
                       double* sab, float* mx)
{
  Geo o;
  if (!make_geo(&o, width, height, depth, slab, "f3d_residual_stats")) return 1;
  orc_residual_stats(P<float>(f0), P<float>(fw), o.W, o.H, o.D, &o.g, ssq, sab, mx);
  return 0;
}

}  // extern "C"
