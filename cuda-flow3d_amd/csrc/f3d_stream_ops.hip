// Streaming / gather kernels of the pyramid loop for gfx950: trilinear backward warp, separable area
// resampling, flow update, separable Gaussian pre-blur, |field| maximum.
//
// Replaces src/kernels/registration_3d.cu, resample_3d.cu, add_3d.cu and convolution_3d.cu of the reference
// (SURVEY.md Appendix A.2, A.1, A.6).  Same expression trees, contraction off.  All of them are single-pass
// HBM-bound maps: a wave64 covers 64 consecutive x of one row (256 B coalesced), a workgroup 4 rows; the
// few neighbouring reads each output needs (<= 13 taps, <= 9 box cells, 8 trilinear corners) hit L1/L2.
#include <cstring>

#include "f3d_internal.h"

#include <cstdint>

namespace {

constexpr int kBX = 64;
constexpr int kBY = 4;

inline dim3 grid_for(int w, int h, int planes) { return dim3((w + kBX - 1) / kBX, (h + kBY - 1) / kBY, planes); }

// Up to three volumes of ONE box per launch (f3d_*_n): u, v, w of a level -- or the two frames -- go through "+=", the median and
// the three resampling passes independently of each other, and below ~128^3 each of those launches is a few microseconds of
// fixed cost around a microsecond of work.  The kernels are the single-volume ones; a workgroup picks its volume from the
// grid (resample: blockIdx.x, the others: blockIdx.z / planes).
constexpr int kMaxBatch = 3;
struct Vols {
  const float* in[kMaxBatch];
  float* out[kMaxBatch];
};

// ---- A.2 warp: src/kernels/registration_3d.cu:28-82 ------------------------------------------------------
__global__ __launch_bounds__(kBX* kBY) void k_warp(const float* __restrict__ f0, const float* __restrict__ f1,
                                                   const float* __restrict__ u, const float* __restrict__ v,
                                                   const float* __restrict__ w, float* __restrict__ out, F3dGeo g,
                                                   float hx, float hy, float hz)
{
  const int x = blockIdx.x * kBX + threadIdx.x;
  const int y = blockIdx.y * kBY + threadIdx.y;
  const int z = g.z_lo + blockIdx.z;
  if (x >= g.W || y >= g.H) return;
  const size_t c = f3d_row(g, y, z) + x;
  const float x_f = static_cast<float>(x) + (u[c] * (1.f / hx));
  const float y_f = static_cast<float>(y) + (v[c] * (1.f / hy));
  const float z_f = static_cast<float>(z) + (w[c] * (1.f / hz));
  if ((x_f < 0.f) || (x_f > static_cast<float>(g.W - 1)) || (y_f < 0.f) || (y_f > static_cast<float>(g.H - 1)) ||
      (z_f < 0.f) || (z_f > static_cast<float>(g.D - 1)) || isnan(x_f) || isnan(y_f) || isnan(z_f)) {
    out[c] = f0[c];
    return;
  }
  const int xi = static_cast<int>(floorf(x_f));
  const int yi = static_cast<int>(floorf(y_f));
  const int zi = static_cast<int>(floorf(z_f));
  const float dx = x_f - static_cast<float>(xi);
  const float dy = y_f - static_cast<float>(yi);
  const float dz = z_f - static_cast<float>(zi);
  const int x1 = min(g.W - 1, xi + 1);
  const int y1 = min(g.H - 1, yi + 1);
  const int z1 = min(g.D - 1, zi + 1);
  const size_t r00 = f3d_row(g, yi, zi), r10 = f3d_row(g, y1, zi);
  const size_t r01 = f3d_row(g, yi, z1), r11 = f3d_row(g, y1, z1);
  const float v0 = (1.f - dx) * (1.f - dy) * f1[r00 + xi] + (dx) * (1.f - dy) * f1[r00 + x1] +
                   (1.f - dx) * (dy)*f1[r10 + xi] + (dx) * (dy)*f1[r10 + x1];
  const float v1 = (1.f - dx) * (1.f - dy) * f1[r01 + xi] + (dx) * (1.f - dy) * f1[r01 + x1] +
                   (1.f - dx) * (dy)*f1[r11 + xi] + (dx) * (dy)*f1[r11 + x1];
  out[c] = (1.f - dz) * v0 + dz * v1;
}

// ---- A.1 resample: src/kernels/resample_3d.cu:28-161 -------------------------------------------------------
// AXIS 0/1/2 = x/y/z.  gi addresses the input container, g the output planes.  A wave owns ONE output row (y, z) and walks
// along x in steps of 64: delta and the normalisation are the reference's two float divisions made once on the host (the same
// IEEE operation), and for the y and z axes the source window and its end fractions belong to the row, not to the voxel, so
// they are formed once per wave instead of once per voxel (one voxel per lane with everything inside was bound by that
// arithmetic: 2.4 TB/s for a 512^3 -> 512^3 pass).
template <int AXIS>
__global__ __launch_bounds__(kBX* kBY) void k_resample(Vols v, F3dGeo gi, F3dGeo g, int in_n, float delta, float normalization)
{
  const float* __restrict__ in = v.in[blockIdx.x];
  float* __restrict__ out = v.out[blockIdx.x];
  const int y = blockIdx.y * kBY + threadIdx.y;
  const int z = g.z_lo + blockIdx.z;
  if (y >= g.H) return;
  float left_f = 0.f, right_f = 0.f;
  int left_i = 0, cnt = 0;
  auto window = [&](int i) {
    left_f = static_cast<float>(i) * delta;
    right_f = static_cast<float>(i + 1) * delta;
    left_i = static_cast<int>(floorf(left_f));
    const int right_i = static_cast<int>(fminf(static_cast<float>(in_n), ceilf(right_f)));
    cnt = right_i - left_i;
  };
  if (AXIS != 0) window(AXIS == 1 ? y : z);
  const size_t out_row = f3d_row(g, y, z);
  const size_t in_row = AXIS == 0 ? f3d_row(gi, y, z) : 0;
  for (int x = threadIdx.x; x < g.W; x += kBX) {
    if (AXIS == 0) window(x);
    float value = 0.f;
    for (int j = 0; j < cnt; ++j) {
      float frac = 1.f;
      if (j == 0) frac = static_cast<float>(left_i + 1) - left_f;
      if (j == cnt - 1) frac = right_f - static_cast<float>(left_i + j);
      if (cnt == 1) frac = delta;
      const int s = left_i + j;
      const size_t a = AXIS == 0 ? in_row + s : (AXIS == 1 ? f3d_row(gi, s, z) + x : f3d_row(gi, y, s) + x);
      value = value + in[a] * frac;
    }
    out[out_row + x] = value * normalization;
  }
}

// The x pass through LDS: a wave streams its source row in with 16 bytes per lane (every level reads the full-size original
// of both frames, so this pass moves the most bytes of the three) and gathers the windows of its outputs from there.
constexpr int kResampleRowMax = 2048;  // floats of a source row a wave can stage (longer rows: k_resample<0>)
__global__ __launch_bounds__(kBX* kBY) void k_resample_x_lds(Vols v, F3dGeo gi, F3dGeo g, int in_n, float delta, float normalization)
{
  __shared__ __attribute__((aligned(16))) float rowbuf[kBY][kResampleRowMax];
  const float* __restrict__ in = v.in[blockIdx.x];
  float* __restrict__ out = v.out[blockIdx.x];
  const int y = blockIdx.y * kBY + threadIdx.y;
  const int z = g.z_lo + blockIdx.z;
  if (y >= g.H) return;
  float* row = rowbuf[threadIdx.y];
  const float* src = in + f3d_row(gi, y, z);
  for (int k = threadIdx.x * 4; k < in_n; k += kBX * 4) *reinterpret_cast<float4*>(row + k) = *reinterpret_cast<const float4*>(src + k);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the wave reads what its own lanes wrote
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const size_t out_row = f3d_row(g, y, z);
  for (int x = threadIdx.x; x < g.W; x += kBX) {
    const float left_f = static_cast<float>(x) * delta;
    const float right_f = static_cast<float>(x + 1) * delta;
    const int left_i = static_cast<int>(floorf(left_f));
    const int right_i = static_cast<int>(fminf(static_cast<float>(in_n), ceilf(right_f)));
    const int cnt = right_i - left_i;
    float value = 0.f;
    for (int j = 0; j < cnt; ++j) {
      float frac = 1.f;
      if (j == 0) frac = static_cast<float>(left_i + 1) - left_f;
      if (j == cnt - 1) frac = right_f - static_cast<float>(left_i + j);
      if (cnt == 1) frac = delta;
      value = value + row[left_i + j] * frac;
    }
    out[out_row + x] = value * normalization;
  }
}

// The y and z passes with four x per lane: the source window belongs to the row, x is contiguous, so a lane moves 16 bytes per
// load.  Rows start on 256-byte boundaries (f3d_alloc_pitched) and the pitch is a multiple of four floats, so the last piece of
// a row may read padding; only the columns inside the box are stored.
template <int AXIS>
__global__ __launch_bounds__(kBX* kBY) void k_resample_x4(Vols v, F3dGeo gi, F3dGeo g, int in_n, float delta, float normalization)
{
  static_assert(AXIS == 1 || AXIS == 2, "x is gathered, not streamed");
  const float* __restrict__ in = v.in[blockIdx.x];
  float* __restrict__ out = v.out[blockIdx.x];
  const int y = blockIdx.y * kBY + threadIdx.y;
  const int z = g.z_lo + blockIdx.z;
  if (y >= g.H) return;
  const int i = AXIS == 1 ? y : z;
  const float left_f = static_cast<float>(i) * delta;
  const float right_f = static_cast<float>(i + 1) * delta;
  const int left_i = static_cast<int>(floorf(left_f));
  const int right_i = static_cast<int>(fminf(static_cast<float>(in_n), ceilf(right_f)));
  const int cnt = right_i - left_i;
  const size_t out_row = f3d_row(g, y, z);
  for (int x = threadIdx.x * 4; x < g.W; x += kBX * 4) {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    for (int j = 0; j < cnt; ++j) {
      float frac = 1.f;
      if (j == 0) frac = static_cast<float>(left_i + 1) - left_f;
      if (j == cnt - 1) frac = right_f - static_cast<float>(left_i + j);
      if (cnt == 1) frac = delta;
      const int s = left_i + j;
      const float4 q = *reinterpret_cast<const float4*>(in + (AXIS == 1 ? f3d_row(gi, s, z) : f3d_row(gi, y, s)) + x);
      v0 = v0 + q.x * frac;
      v1 = v1 + q.y * frac;
      v2 = v2 + q.z * frac;
      v3 = v3 + q.w * frac;
    }
    float* o = out + out_row + x;
    if (x + 3 < g.W) {
      *reinterpret_cast<float4*>(o) = make_float4(v0 * normalization, v1 * normalization, v2 * normalization, v3 * normalization);
    } else {
      o[0] = v0 * normalization;
      if (x + 1 < g.W) o[1] = v1 * normalization;
      if (x + 2 < g.W) o[2] = v2 * normalization;
    }
  }
}

// ---- add: src/kernels/add_3d.cu:26-41 ------------------------------------------------------------------------
// v.out = operand_0 (read and written), v.in = operand_1; grid.z = planes x volumes
__global__ __launch_bounds__(kBX* kBY) void k_add(Vols v, F3dGeo g)
{
  const int planes = g.z_hi - g.z_lo;
  const int vol = static_cast<int>(blockIdx.z) / planes;
  float* __restrict__ a = v.out[vol];
  const float* __restrict__ b = v.in[vol];
  const int x = blockIdx.x * kBX + threadIdx.x;
  const int y = blockIdx.y * kBY + threadIdx.y;
  const int z = g.z_lo + (static_cast<int>(blockIdx.z) - vol * planes);
  if (x >= g.W || y >= g.H) return;
  const size_t c = f3d_row(g, y, z) + x;
  a[c] = a[c] + b[c];
}

// the box [0, W) x [0, H) x [z_lo, z_hi) of up to three volumes set to +0: 16 bytes per lane where a row allows it (rows start
// 16-byte aligned when the pitch is a multiple of four floats and the base is); the increments of a level, cleared in one launch
__global__ __launch_bounds__(256) void k_clear_box(Vols v, F3dGeo g, int chunks_per_row, int rows, int vec)
{
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  const unsigned total = static_cast<unsigned>(chunks_per_row) * static_cast<unsigned>(rows);
  if (t >= total) return;
  float* __restrict__ p = v.out[blockIdx.y];
  const int r = static_cast<int>(t / static_cast<unsigned>(chunks_per_row));
  const int c = static_cast<int>(t - static_cast<unsigned>(r) * static_cast<unsigned>(chunks_per_row));
  const int zi = r / g.H, y = r - zi * g.H;
  float* row = p + f3d_row(g, y, g.z_lo + zi);
  if (vec) {
    const int x = c * 4;
    if (x + 3 < g.W) {
      *reinterpret_cast<float4*>(row + x) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (int k = x; k < g.W; ++k) row[k] = 0.f;
    }
  } else {
    row[c] = 0.f;
  }
}

// ---- A.6 Gaussian passes: src/kernels/convolution_3d.cu:75-172,186-271,284-372 ---------------------------------
// Clean spec: zero padding, sum = sum + k[R - j] * s[i + j] for j = -R..R ascending, sum starting at 0.
template <int AXIS>
__global__ __launch_bounds__(kBX* kBY) void k_conv(float* __restrict__ dst, const float* __restrict__ src, F3dGeo g,
                                                   f3d::ConvTaps taps, int radius)
{
  const int x = blockIdx.x * kBX + threadIdx.x;
  const int y = blockIdx.y * kBY + threadIdx.y;
  const int z = g.z_lo + blockIdx.z;
  if (x >= g.W || y >= g.H) return;
  const int n = AXIS == 0 ? g.W : (AXIS == 1 ? g.H : g.D);
  const int i = AXIS == 0 ? x : (AXIS == 1 ? y : z);
  float sum = 0.f;
  for (int j = -radius; j <= radius; ++j) {
    const int s = i + j;
    float val = 0.f;
    if (s >= 0 && s < n) {
      const size_t a = AXIS == 0 ? f3d_row(g, y, z) + s : (AXIS == 1 ? f3d_row(g, s, z) + x : f3d_row(g, y, s) + x);
      val = src[a];
    }
    sum = sum + taps.k[radius - j] * val;
  }
  dst[f3d_row(g, y, z) + x] = sum;
}

// ---- max |field| over a slab (feeds the warp halo depth of the z-slab decomposition) ---------------------------
__global__ __launch_bounds__(256) void k_abs_max(const float* __restrict__ f, F3dGeo g, unsigned* result)
{
  // Largest FINITE |x|.  The result sizes the z reach of the warp, and the warp sends a voxel whose flow is NaN or infinite
  // to frame_0 without looking anywhere (registration_3d.cu:60-64): such voxels need no reach, and leaving them out keeps
  // the result finite whatever the field holds (the callers convert it to a plane count).
  const int z = g.z_lo + blockIdx.z;
  float m = 0.f;
  for (int y = blockIdx.y; y < g.H; y += gridDim.y) {
    const size_t r = f3d_row(g, y, z);
    for (int x = threadIdx.x; x < g.W; x += blockDim.x) {
      const float v = fabsf(f[r + x]);
      if (v < __builtin_inff()) m = fmaxf(m, v);  // false for NaN and for Inf
    }
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
  // one atomic per workgroup, not per wave: thousands of atomics on one address serialise in the L2 (85 us per call on a
  // 512 x 512 x 74 slab before)
  __shared__ float wave_max[4];
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
    atomicMax(result, __float_as_uint(m));  // non-negative finite floats order like uints
  }
}

// ---- flow statistics: min / max / sum of |(u, v, w)| over a slab -------------------------------------------------------
// cuda_operation_stat_p.cpp:85-104 of the reference does this on the host after a download; here the flow stays on the
// device.  Magnitudes are non-negative, so their bit patterns order like unsigned integers (atomicMin / atomicMax);
// the sum is accumulated in double (the reference adds floats in scan order, which no parallel reduction reproduces).
struct FlowStats {
  unsigned min_bits, max_bits;
  double sum;
};
__global__ __launch_bounds__(256) void k_flow_stats(const float* __restrict__ u, const float* __restrict__ v,
                                                    const float* __restrict__ w, F3dGeo g, FlowStats* out)
{
  const int z = g.z_lo + blockIdx.z;
  float lo = __uint_as_float(0x7f7fffffu), hi = 0.f;
  double sum = 0.0;
  for (int y = blockIdx.y; y < g.H; y += gridDim.y) {
    const size_t r = f3d_row(g, y, z);
    for (int x = threadIdx.x; x < g.W; x += blockDim.x) {
      const float a = u[r + x], b = v[r + x], c = w[r + x];
      const float m = sqrtf(a * a + b * b + c * c);
      lo = fminf(lo, m);
      hi = fmaxf(hi, m);
      sum += static_cast<double>(m);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_down(lo, off));
    hi = fmaxf(hi, __shfl_down(hi, off));
    sum += __shfl_down(sum, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&out->min_bits, __float_as_uint(lo));
    atomicMax(&out->max_bits, __float_as_uint(hi));
    atomicAdd(&out->sum, sum);
  }
}

// ---- registration residual: how well frame_1 warped by the flow matches frame_0 -----------------------------------------
// The reference's only diagnostic of a result is a disabled debug block that registers frame_1 with the final flow and dumps
// the volume (optical_flow_e.cpp:536-571).  Here the comparison itself runs on the device: sum of squares, sum of absolute
// values (both in double) and the maximum of |warped - frame_0| over a slab.
struct ResidualStats {
  double sum_sq, sum_abs;
  unsigned max_bits;
  unsigned pad;
};
__global__ __launch_bounds__(256) void k_residual_stats(const float* __restrict__ f0, const float* __restrict__ fw, F3dGeo g,
                                                        ResidualStats* out)
{
  const int z = g.z_lo + blockIdx.z;
  double ssq = 0.0, sab = 0.0;
  float hi = 0.f;
  for (int y = blockIdx.y; y < g.H; y += gridDim.y) {
    const size_t r = f3d_row(g, y, z);
    for (int x = threadIdx.x; x < g.W; x += blockDim.x) {
      const float d = fw[r + x] - f0[r + x];
      const float a = fabsf(d);
      hi = fmaxf(hi, a);
      sab += static_cast<double>(a);
      ssq += static_cast<double>(d) * static_cast<double>(d);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    hi = fmaxf(hi, __shfl_down(hi, off));
    ssq += __shfl_down(ssq, off);
    sab += __shfl_down(sab, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&out->max_bits, __float_as_uint(hi));
    atomicAdd(&out->sum_sq, ssq);
    atomicAdd(&out->sum_abs, sab);
  }
}

bool same_buffer(f3d_devptr a, f3d_devptr b, const char* who)
{
  if (a == b) {
    f3d::fail("%s: input buffer cannot serve as output buffer", who);
    return true;
  }
  return false;
}

bool planes_inside(const F3dGeo& g, int lo, int hi, const char* who)
{
  const int dc = static_cast<int>(f3d::container().depth);
  if (lo < g.z_base || hi - g.z_base > dc) {
    f3d::fail("%s: planes [%d,%d) needed but the container holds [%d,%d)", who, lo, hi, g.z_base, g.z_base + dc);
    return false;
  }
  return true;
}

}  // namespace

extern "C" {

int f3d_warp(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
             size_t width, size_t height, size_t depth, float hx, float hy, float hz, f3d_devptr output,
             const f3d_slab* slab)
{
  F3D_REQUIRE_READY("f3d_warp");
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, "f3d_warp")) return 1;
  if (same_buffer(frame_1, output, "f3d_warp")) return 1;
  if (g.z_lo == g.z_hi) return 0;
  hipLaunchKernelGGL(k_warp, grid_for(g.W, g.H, g.z_hi - g.z_lo), dim3(kBX, kBY, 1), 0, f3d::stream(),
                     f3d_ptr<const float>(frame_0), f3d_ptr<const float>(frame_1), f3d_ptr<const float>(flow_u),
                     f3d_ptr<const float>(flow_v), f3d_ptr<const float>(flow_w), f3d_ptr<float>(output), g, hx, hy, hz);
  F3D_HIP(hipGetLastError());
  return 0;
}

static bool batch_ok(size_t count, const char* who)
{
  if (count == 0 || count > static_cast<size_t>(kMaxBatch)) {
    f3d::fail("%s: %zu volumes (one launch takes 1 .. %d)", who, count, kMaxBatch);
    return false;
  }
  return true;
}

static int resample_launch(int axis, const f3d_devptr* inputs, const f3d_devptr* outputs, size_t count, size_t ow, size_t oh,
                           size_t od, size_t in_n, const f3d_slab* slab_in, const f3d_slab* slab, const char* who)
{
  F3D_REQUIRE_READY(who);
  if (!inputs || !outputs) return f3d::fail("%s: null argument", who);
  if (!batch_ok(count, who)) return 1;
  // no volume of the batch may be read and written: the passes stream rows, every output row of a volume depends on other rows
  for (size_t i = 0; i < count; ++i)
    for (size_t j = 0; j < count; ++j)
      if (same_buffer(inputs[i], outputs[j], who)) return 1;
  if (in_n == 0) return f3d::fail("%s: empty input axis", who);
  F3dGeo g, gi;
  if (!f3d::make_geo(&g, ow, oh, od, slab, who)) return 1;
  // the input box: same as the output except along the resampled axis
  const size_t iw = axis == 0 ? in_n : ow, ih = axis == 1 ? in_n : oh, id = axis == 2 ? in_n : od;
  if (!f3d::make_geo(&gi, iw, ih, id, axis == 2 ? slab_in : slab, who)) return 1;
  if (g.z_lo == g.z_hi) return 0;
  if (axis == 2) {
    // output plane z reads input planes [floor(z*delta), ceil((z+1)*delta)) -- check the container holds them
    const float delta = static_cast<float>(in_n) / static_cast<float>(od);
    const int lo = static_cast<int>(floorf(static_cast<float>(g.z_lo) * delta));
    int hi = static_cast<int>(ceilf(static_cast<float>(g.z_hi) * delta));
    if (hi > static_cast<int>(in_n)) hi = static_cast<int>(in_n);
    if (!planes_inside(gi, lo, hi, who)) return 1;
  }
  // one block of 4 waves per 4 rows; a wave walks its row (small levels: still enough blocks, rows x planes / 4); grid.x = volume
  const dim3 grid(static_cast<unsigned>(count), (g.H + kBY - 1) / kBY, g.z_hi - g.z_lo), block(kBX, kBY, 1);
  Vols v = {};
  // 16 bytes per lane where every row of both containers starts 16-byte aligned
  bool x4 = g.pitch % 4 == 0 && gi.pitch % 4 == 0;
  for (size_t i = 0; i < count; ++i) {
    v.in[i] = f3d_ptr<const float>(inputs[i]);
    v.out[i] = f3d_ptr<float>(outputs[i]);
    x4 = x4 && reinterpret_cast<uintptr_t>(v.in[i]) % 16 == 0 && reinterpret_cast<uintptr_t>(v.out[i]) % 16 == 0;
  }
  const int n = static_cast<int>(in_n);
  const int out_n = axis == 0 ? g.W : (axis == 1 ? g.H : g.D);
  const float delta = static_cast<float>(n) / static_cast<float>(out_n);          // resample_3d.cu: the kernels' own divisions
  const float normalization = static_cast<float>(out_n) / static_cast<float>(n);
  const bool staged = x4 && n <= kResampleRowMax && (n + 3) / 4 * 4 <= gi.pitch;
  if (axis == 0 && staged) hipLaunchKernelGGL(k_resample_x_lds, grid, block, 0, f3d::stream(), v, gi, g, n, delta, normalization);
  if (axis == 0 && !staged) hipLaunchKernelGGL(k_resample<0>, grid, block, 0, f3d::stream(), v, gi, g, n, delta, normalization);
  if (axis == 1 && x4) hipLaunchKernelGGL(k_resample_x4<1>, grid, block, 0, f3d::stream(), v, gi, g, n, delta, normalization);
  if (axis == 2 && x4) hipLaunchKernelGGL(k_resample_x4<2>, grid, block, 0, f3d::stream(), v, gi, g, n, delta, normalization);
  if (axis == 1 && !x4) hipLaunchKernelGGL(k_resample<1>, grid, block, 0, f3d::stream(), v, gi, g, n, delta, normalization);
  if (axis == 2 && !x4) hipLaunchKernelGGL(k_resample<2>, grid, block, 0, f3d::stream(), v, gi, g, n, delta, normalization);
  F3D_HIP(hipGetLastError());
  return 0;
}

int f3d_resample_x(f3d_devptr input, f3d_devptr output, size_t out_width, size_t out_height, size_t out_depth,
                   size_t in_width, const f3d_slab* slab)
{
  return resample_launch(0, &input, &output, 1, out_width, out_height, out_depth, in_width, nullptr, slab, "f3d_resample_x");
}

int f3d_resample_y(f3d_devptr input, f3d_devptr output, size_t out_width, size_t out_height, size_t out_depth,
                   size_t in_height, const f3d_slab* slab)
{
  return resample_launch(1, &input, &output, 1, out_width, out_height, out_depth, in_height, nullptr, slab, "f3d_resample_y");
}

int f3d_resample_z(f3d_devptr input, f3d_devptr output, size_t out_width, size_t out_height, size_t out_depth,
                   size_t in_depth, const f3d_slab* slab_in, const f3d_slab* slab)
{
  return resample_launch(2, &input, &output, 1, out_width, out_height, out_depth, in_depth, slab_in, slab, "f3d_resample_z");
}

int f3d_resample_x_n(const f3d_devptr* inputs, const f3d_devptr* outputs, size_t count, size_t out_width, size_t out_height,
                     size_t out_depth, size_t in_width, const f3d_slab* slab)
{
  return resample_launch(0, inputs, outputs, count, out_width, out_height, out_depth, in_width, nullptr, slab, "f3d_resample_x_n");
}

int f3d_resample_y_n(const f3d_devptr* inputs, const f3d_devptr* outputs, size_t count, size_t out_width, size_t out_height,
                     size_t out_depth, size_t in_height, const f3d_slab* slab)
{
  return resample_launch(1, inputs, outputs, count, out_width, out_height, out_depth, in_height, nullptr, slab, "f3d_resample_y_n");
}

int f3d_resample_z_n(const f3d_devptr* inputs, const f3d_devptr* outputs, size_t count, size_t out_width, size_t out_height,
                     size_t out_depth, size_t in_depth, const f3d_slab* slab_in, const f3d_slab* slab)
{
  return resample_launch(2, inputs, outputs, count, out_width, out_height, out_depth, in_depth, slab_in, slab, "f3d_resample_z_n");
}

static int add_launch(const f3d_devptr* operand_0, const f3d_devptr* operand_1, size_t count, size_t width, size_t height,
                      size_t depth, const f3d_slab* slab, const char* who)
{
  F3D_REQUIRE_READY(who);
  if (!operand_0 || !operand_1) return f3d::fail("%s: null argument", who);
  if (!batch_ok(count, who)) return 1;
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, who)) return 1;
  if (g.z_lo == g.z_hi) return 0;
  Vols v = {};
  for (size_t i = 0; i < count; ++i) {
    v.out[i] = f3d_ptr<float>(operand_0[i]);
    v.in[i] = f3d_ptr<const float>(operand_1[i]);
  }
  const int planes = g.z_hi - g.z_lo;
  if (static_cast<long>(planes) * static_cast<long>(count) > 65535L) {  // grid.z: a window that deep goes volume by volume
    for (size_t i = 0; i < count; ++i) {
      Vols one = {};
      one.in[0] = v.in[i];
      one.out[0] = v.out[i];
      hipLaunchKernelGGL(k_add, grid_for(g.W, g.H, planes), dim3(kBX, kBY, 1), 0, f3d::stream(), one, g);
    }
  } else {
    hipLaunchKernelGGL(k_add, grid_for(g.W, g.H, planes * static_cast<int>(count)), dim3(kBX, kBY, 1), 0, f3d::stream(), v, g);
  }
  F3D_HIP(hipGetLastError());
  return 0;
}

int f3d_add(f3d_devptr operand_0, f3d_devptr operand_1, size_t width, size_t height, size_t depth, const f3d_slab* slab)
{
  return add_launch(&operand_0, &operand_1, 1, width, height, depth, slab, "f3d_add");
}

int f3d_add_n(const f3d_devptr* operand_0, const f3d_devptr* operand_1, size_t count, size_t width, size_t height, size_t depth,
              const f3d_slab* slab)
{
  return add_launch(operand_0, operand_1, count, width, height, depth, slab, "f3d_add_n");
}

int f3d_clear_box_n(const f3d_devptr* volumes, size_t count, size_t width, size_t height, size_t depth, const f3d_slab* slab)
{
  F3D_REQUIRE_READY("f3d_clear_box_n");
  if (!volumes) return f3d::fail("f3d_clear_box_n: null argument");
  if (!batch_ok(count, "f3d_clear_box_n")) return 1;
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, "f3d_clear_box_n")) return 1;
  if (g.z_lo == g.z_hi) return 0;
  Vols v = {};
  bool vec = g.pitch % 4 == 0;
  for (size_t i = 0; i < count; ++i) {
    v.out[i] = f3d_ptr<float>(volumes[i]);
    vec = vec && reinterpret_cast<uintptr_t>(v.out[i]) % 16 == 0;
  }
  const int chunks = vec ? (g.W + 3) / 4 : g.W;
  const long rows = static_cast<long>(g.H) * (g.z_hi - g.z_lo);
  const long total = rows * chunks;
  if (total >= (1L << 32)) return f3d::fail("f3d_clear_box_n: box too large for one launch");
  const dim3 grid(static_cast<unsigned>((total + 255) / 256), static_cast<unsigned>(count), 1);
  hipLaunchKernelGGL(k_clear_box, grid, dim3(256, 1, 1), 0, f3d::stream(), v, g, chunks, static_cast<int>(rows), vec ? 1 : 0);
  F3D_HIP(hipGetLastError());
  return 0;
}

static int conv_launch(int axis, f3d_devptr dst, f3d_devptr src, size_t w, size_t h, size_t d, size_t radius,
                       const f3d_slab* slab, const char* who)
{
  F3D_REQUIRE_READY(who);
  if (same_buffer(dst, src, who)) return 1;
  const f3d::ConvTaps& taps = f3d::conv_taps();
  if (taps.count != static_cast<int>(2 * radius + 1))
    return f3d::fail("%s: radius %zu does not match the %d taps uploaded with f3d_set_conv_taps", who, radius, taps.count);
  F3dGeo g;
  if (!f3d::make_geo(&g, w, h, d, slab, who)) return 1;
  if (g.z_lo == g.z_hi) return 0;
  const dim3 grid = grid_for(g.W, g.H, g.z_hi - g.z_lo), block(kBX, kBY, 1);
  float* o = f3d_ptr<float>(dst);
  const float* s = f3d_ptr<const float>(src);
  const int r = static_cast<int>(radius);
  if (axis == 0) hipLaunchKernelGGL(k_conv<0>, grid, block, 0, f3d::stream(), o, s, g, taps, r);
  if (axis == 1) hipLaunchKernelGGL(k_conv<1>, grid, block, 0, f3d::stream(), o, s, g, taps, r);
  F3D_HIP(hipGetLastError());
  return 0;
}

int f3d_conv_rows(f3d_devptr dst, f3d_devptr src, size_t width, size_t height, size_t depth, size_t kernel_radius,
                  const f3d_slab* slab)
{
  return conv_launch(0, dst, src, width, height, depth, kernel_radius, slab, "f3d_conv_rows");
}

int f3d_conv_cols(f3d_devptr dst, f3d_devptr src, size_t width, size_t height, size_t depth, size_t kernel_radius,
                  const f3d_slab* slab)
{
  return conv_launch(1, dst, src, width, height, depth, kernel_radius, slab, "f3d_conv_cols");
}

int f3d_abs_max(f3d_devptr field, size_t width, size_t height, size_t depth, const f3d_slab* slab, float* result)
{
  F3D_REQUIRE_READY("f3d_abs_max");
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, "f3d_abs_max")) return 1;
  static thread_local unsigned* d_result = nullptr;   // per thread: two lanes may ask at once
  if (!d_result) F3D_HIP(hipMalloc(reinterpret_cast<void**>(&d_result), sizeof(unsigned)));
  F3D_HIP(hipMemsetAsync(d_result, 0, sizeof(unsigned), f3d::stream()));
  if (g.z_hi > g.z_lo) {
    const int gy = g.H < 16 ? g.H : 16;
    hipLaunchKernelGGL(k_abs_max, dim3(1, gy, g.z_hi - g.z_lo), dim3(256, 1, 1), 0, f3d::stream(),
                       f3d_ptr<const float>(field), g, d_result);
    F3D_HIP(hipGetLastError());
  }
  unsigned bits = 0;
  F3D_HIP(hipMemcpyAsync(&bits, d_result, sizeof(unsigned), hipMemcpyDeviceToHost, f3d::stream()));
  F3D_HIP(hipStreamSynchronize(f3d::stream()));
  std::memcpy(result, &bits, sizeof(float));
  return 0;
}

int f3d_flow_stats(f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w, size_t width, size_t height, size_t depth,
                   const f3d_slab* slab, float* min_magnitude, float* max_magnitude, double* sum_magnitude)
{
  F3D_REQUIRE_READY("f3d_flow_stats");
  if (!min_magnitude || !max_magnitude || !sum_magnitude) return f3d::fail("f3d_flow_stats: null argument");
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, "f3d_flow_stats")) return 1;
  static thread_local FlowStats* d_stats = nullptr;
  if (!d_stats) F3D_HIP(hipMalloc(reinterpret_cast<void**>(&d_stats), sizeof(FlowStats)));
  const FlowStats init = {0x7f7fffffu, 0u, 0.0};  // FLT_MAX, 0, 0
  F3D_HIP(hipMemcpyAsync(d_stats, &init, sizeof(init), hipMemcpyHostToDevice, f3d::stream()));
  if (g.z_hi > g.z_lo) {
    const int gy = g.H < 64 ? g.H : 64;
    hipLaunchKernelGGL(k_flow_stats, dim3(1, gy, g.z_hi - g.z_lo), dim3(256, 1, 1), 0, f3d::stream(), f3d_ptr<const float>(flow_u),
                       f3d_ptr<const float>(flow_v), f3d_ptr<const float>(flow_w), g, d_stats);
    F3D_HIP(hipGetLastError());
  }
  FlowStats h;
  F3D_HIP(hipMemcpyAsync(&h, d_stats, sizeof(h), hipMemcpyDeviceToHost, f3d::stream()));
  F3D_HIP(hipStreamSynchronize(f3d::stream()));
  std::memcpy(min_magnitude, &h.min_bits, sizeof(float));
  std::memcpy(max_magnitude, &h.max_bits, sizeof(float));
  *sum_magnitude = h.sum;
  return 0;
}

int f3d_residual_stats(f3d_devptr frame_0, f3d_devptr frame_1_warped, size_t width, size_t height, size_t depth,
                       const f3d_slab* slab, double* sum_squares, double* sum_abs, float* max_abs)
{
  F3D_REQUIRE_READY("f3d_residual_stats");
  if (!sum_squares || !sum_abs || !max_abs) return f3d::fail("f3d_residual_stats: null argument");
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, "f3d_residual_stats")) return 1;
  static thread_local ResidualStats* d_stats = nullptr;
  if (!d_stats) F3D_HIP(hipMalloc(reinterpret_cast<void**>(&d_stats), sizeof(ResidualStats)));
  F3D_HIP(hipMemsetAsync(d_stats, 0, sizeof(ResidualStats), f3d::stream()));
  if (g.z_hi > g.z_lo) {
    const int gy = g.H < 64 ? g.H : 64;
    hipLaunchKernelGGL(k_residual_stats, dim3(1, gy, g.z_hi - g.z_lo), dim3(256, 1, 1), 0, f3d::stream(),
                       f3d_ptr<const float>(frame_0), f3d_ptr<const float>(frame_1_warped), g, d_stats);
    F3D_HIP(hipGetLastError());
  }
  ResidualStats h;
  F3D_HIP(hipMemcpyAsync(&h, d_stats, sizeof(h), hipMemcpyDeviceToHost, f3d::stream()));
  F3D_HIP(hipStreamSynchronize(f3d::stream()));
  *sum_squares = h.sum_sq;
  *sum_abs = h.sum_abs;
  std::memcpy(max_abs, &h.max_bits, sizeof(float));
  return 0;
}

}  // extern "C"
