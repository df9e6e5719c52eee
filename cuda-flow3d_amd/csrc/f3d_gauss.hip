// Separable Gaussian pre-blur for gfx950: rows + columns in one kernel through LDS, slices as a z march.
//
// Replaces src/kernels/convolution_3d.cu (convolutionRowsKernel :75-172, convolutionColumnsKernel :186-271,
// convolutionSlicesKernel :284-372) on the path OpticalFlowE takes (optical_flow_e.cpp:213-242); the one-pass-per-launch
// kernels of f3d_stream_ops.hip stay behind f3d_conv_rows / f3d_conv_cols for callers that want a single pass.
// Clean zero-padded spec of SURVEY.md A.6, per axis:  sum = 0;  for j = -R .. R:  sum = sum + k[R - j] * s[i + j]  (s = 0 outside the
// volume), every product and every sum rounded separately (-ffp-contract=off).  Both kernels add the taps of an output in
// exactly that order, zero terms included, so the bits are those of three separate passes.
//
//   k_gauss_xy : a workgroup owns 64 x 32 outputs of one plane.  The (64 + 2R) x (32 + 2R) input patch is fetched once into
//                LDS (zero outside the volume), the row pass runs on all 32 + 2R patch rows into a second LDS image, the column
//                pass reads that image: one HBM read and one write per voxel for two passes (the x-convolved volume never exists
//                in memory), neighbours from LDS at consecutive addresses (no bank conflicts).
//   k_gauss_z  : a thread owns one (x, y) and marches along z with the last 2R + 1 planes of its column in an LDS ring that only
//                it touches (no barriers); every plane is read once per z-chunk, chunks overlap by 2R planes.
#include "f3d_internal.h"

namespace {

constexpr int kGX = 64, kGY = 32;   // outputs per workgroup of k_gauss_xy
constexpr int kMaxR = 25;           // 51 taps (MAX_KERNEL_LENGTH, convolution_3d.cu:49)

// RT > 0: the radius as a compile-time constant (loops unrolled, taps in scalar registers, LDS offsets immediate) for the radii
// the application really uses (sigma = 2 -> R = 6); RT = 0: any radius up to 25 at run time.
template <int RT>
__global__ __launch_bounds__(256) void k_gauss_xy(float* __restrict__ dst, const float* __restrict__ src, F3dGeo g,
                                                  f3d::ConvTaps taps, int Rrt)
{
  const int R = RT > 0 ? RT : Rrt;
  extern __shared__ float lds[];
  const int PW = kGX + 2 * R, PH = kGY + 2 * R;
  float* A = lds;                // input patch  [PH][PW]
  float* B = lds + PH * PW;      // row-convolved [PH][kGX]
  const int lane = threadIdx.x, wy = threadIdx.y;  // 64 x 4
  const int x0 = blockIdx.x * kGX, y0 = blockIdx.y * kGY;
  const int z = g.z_lo + blockIdx.z;

  for (int rr = wy; rr < PH; rr += 4) {
    const int gy = y0 - R + rr;
    const bool row_in = gy >= 0 && gy < g.H;
    const size_t row = row_in ? f3d_row(g, gy, z) : 0;
    for (int cc = lane; cc < PW; cc += kGX) {
      const int gx = x0 - R + cc;
      A[rr * PW + cc] = (row_in && gx >= 0 && gx < g.W) ? src[row + gx] : 0.f;
    }
  }
  __syncthreads();
  for (int rr = wy; rr < PH; rr += 4) {
    const float* a = A + rr * PW + lane;  // a[R + j] = patch column x + j
    float sum = 0.f;
    if (RT > 0) {
#pragma unroll
      for (int j = -RT; j <= RT; ++j) sum = sum + taps.k[RT - j] * a[RT + j];
    } else {
      for (int j = -R; j <= R; ++j) sum = sum + taps.k[R - j] * a[R + j];
    }
    B[rr * kGX + lane] = sum;
  }
  __syncthreads();
  const int gx = x0 + lane;
  for (int oy = wy; oy < kGY; oy += 4) {
    const int gy = y0 + oy;
    if (gx >= g.W || gy >= g.H) continue;
    const float* b = B + (oy + R) * kGX + lane;
    float sum = 0.f;
    if (RT > 0) {
#pragma unroll
      for (int j = -RT; j <= RT; ++j) sum = sum + taps.k[RT - j] * b[j * kGX];
    } else {
      for (int j = -R; j <= R; ++j) sum = sum + taps.k[R - j] * b[j * kGX];
    }
    dst[f3d_row(g, gy, z) + gx] = sum;
  }
}

__global__ __launch_bounds__(256) void k_gauss_z(float* __restrict__ dst, const float* __restrict__ src, F3dGeo g,
                                                 f3d::ConvTaps taps, int R, int zchunk)
{
  extern __shared__ float ring[];  // [2R + 1][256]: thread t owns column t
  const int K = 2 * R + 1;
  const int t = threadIdx.y * 64 + threadIdx.x;
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int z0 = g.z_lo + blockIdx.z * zchunk;
  const int z1 = min(z0 + zchunk, g.z_hi);
  if (x >= g.W || y >= g.H) return;  // no barrier below: a thread only ever reads what it wrote itself
  float* col = ring + t;
  int slot = 0;  // ring slot of the plane that arrives next; plane p sits in slot (p - (z0 - R)) mod K
  for (int p = z0 - R; p < z1 + R; ++p) {
    const float v = (p >= 0 && p < g.D) ? src[f3d_row(g, y, p) + x] : 0.f;
    col[slot * 256] = v;
    slot = slot + 1 == K ? 0 : slot + 1;
    const int zo = p - R;  // the output whose window [zo - R, zo + R] is now complete
    if (zo >= z0) {
      // the window starts at the slot right after the one just written (the oldest plane), taps in ascending z
      int s = slot;
      float sum = 0.f;
      for (int j = -R; j <= R; ++j) {
        sum = sum + taps.k[R - j] * col[s * 256];
        s = s + 1 == K ? 0 : s + 1;
      }
      dst[f3d_row(g, y, zo) + x] = sum;
    }
  }
}

// The same march with the window in registers: 2 RT + 1 values, roles rotated by unrolling the plane loop 2 RT + 1 deep, so
// nothing is ever moved.  Per output 2 RT + 1 multiplies and adds, one load, one store, no LDS.
template <int RT>
__global__ __launch_bounds__(256) void k_gauss_z_reg(float* __restrict__ dst, const float* __restrict__ src, F3dGeo g,
                                                     f3d::ConvTaps taps, int zchunk)
{
  constexpr int K = 2 * RT + 1;
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int z0 = g.z_lo + blockIdx.z * zchunk;
  const int z1 = min(z0 + zchunk, g.z_hi);
  if (x >= g.W || y >= g.H) return;
  const size_t plane = static_cast<size_t>(g.Hc) * static_cast<size_t>(g.pitch);
  const float* in = src + f3d_row(g, y, g.z_base) + x;   // in[(p - z_base) * plane] = plane p
  float* out = dst + f3d_row(g, y, g.z_base) + x;
  float win[K];
#pragma unroll
  for (int i = 0; i < K; ++i) win[i] = 0.f;
  // plane p goes to win[(p - (z0 - RT)) mod K]; after it the window of output p - RT is complete
  for (int pb = z0 - RT; pb < z1 + RT; pb += K) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int p = pb + i;
      if (p < z1 + RT) {
        win[i] = (p >= 0 && p < g.D) ? in[static_cast<size_t>(p - g.z_base) * plane] : 0.f;
        const int zo = p - RT;
        if (zo >= z0) {
          float sum = 0.f;
#pragma unroll
          for (int j = 0; j < K; ++j) sum = sum + taps.k[K - 1 - j] * win[(i + 1 + j) % K];  // oldest plane first: ascending z
          out[static_cast<size_t>(zo - g.z_base) * plane] = sum;
        }
      }
    }
  }
}

bool planes_in_container(const F3dGeo& g, int lo, int hi, const char* who)
{
  const int dc = static_cast<int>(f3d::container().depth);
  if (lo < g.z_base || hi - g.z_base > dc) {
    f3d::fail("%s: planes [%d,%d) needed but the container holds [%d,%d)", who, lo, hi, g.z_base, g.z_base + dc);
    return false;
  }
  return true;
}

int check_taps(size_t radius, const char* who)
{
  const f3d::ConvTaps& taps = f3d::conv_taps();
  if (radius > static_cast<size_t>(kMaxR)) return f3d::fail("%s: radius %zu exceeds the %d-tap limit", who, radius, 2 * kMaxR + 1);
  if (taps.count != static_cast<int>(2 * radius + 1))
    return f3d::fail("%s: radius %zu does not match the %d taps uploaded with f3d_set_conv_taps", who, radius, taps.count);
  return 0;
}

}  // namespace

extern "C" {

int f3d_conv_rows_cols(f3d_devptr dst, f3d_devptr src, size_t width, size_t height, size_t depth, size_t kernel_radius,
                       const f3d_slab* slab)
{
  F3D_REQUIRE_READY("f3d_conv_rows_cols");
  if (dst == src) return f3d::fail("f3d_conv_rows_cols: input buffer cannot serve as output buffer");
  if (check_taps(kernel_radius, "f3d_conv_rows_cols")) return 1;
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, "f3d_conv_rows_cols")) return 1;
  if (g.z_lo == g.z_hi) return 0;
  const int R = static_cast<int>(kernel_radius);
  const size_t lds = static_cast<size_t>(kGY + 2 * R) * (kGX + 2 * R + kGX) * sizeof(float);
  const dim3 grid((g.W + kGX - 1) / kGX, (g.H + kGY - 1) / kGY, g.z_hi - g.z_lo), block(64, 4, 1);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, grid, block, lds, f3d::stream(), f3d_ptr<float>(dst), f3d_ptr<const float>(src), g, f3d::conv_taps(), R);
  };
  if (R == 6) go(k_gauss_xy<6>);
  else if (R == 3) go(k_gauss_xy<3>);
  else go(k_gauss_xy<0>);
  F3D_HIP(hipGetLastError());
  return 0;
}

int f3d_conv_slices(f3d_devptr dst, f3d_devptr src, size_t width, size_t height, size_t depth, size_t kernel_radius,
                    const f3d_slab* slab)
{
  F3D_REQUIRE_READY("f3d_conv_slices");
  if (dst == src) return f3d::fail("f3d_conv_slices: input buffer cannot serve as output buffer");
  if (check_taps(kernel_radius, "f3d_conv_slices")) return 1;
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, "f3d_conv_slices")) return 1;
  if (g.z_lo == g.z_hi) return 0;
  const int R = static_cast<int>(kernel_radius);
  if (!planes_in_container(g, g.z_lo - R < 0 ? 0 : g.z_lo - R, g.z_hi + R > g.D ? g.D : g.z_hi + R, "f3d_conv_slices")) return 1;
  // z-chunks: enough workgroups for every CU several times over, chunks long enough that the 2R planes two neighbours both
  // read stay a small share
  const int planes = g.z_hi - g.z_lo;
  const long tiles = static_cast<long>((g.W + 63) / 64) * ((g.H + 3) / 4);
  long chunks = (2048 + tiles - 1) / tiles;
  const long max_chunks = planes / (8 * R + 8) > 0 ? planes / (8 * R + 8) : 1;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  const int zchunk = static_cast<int>((planes + chunks - 1) / chunks);
  const dim3 grid((g.W + 63) / 64, (g.H + 3) / 4, (planes + zchunk - 1) / zchunk), block(64, 4, 1);
  const size_t lds = static_cast<size_t>(2 * R + 1) * 256 * sizeof(float);
  float* o = f3d_ptr<float>(dst);
  const float* i = f3d_ptr<const float>(src);
  if (R == 6) hipLaunchKernelGGL(k_gauss_z_reg<6>, grid, block, 0, f3d::stream(), o, i, g, f3d::conv_taps(), zchunk);
  else if (R == 3) hipLaunchKernelGGL(k_gauss_z_reg<3>, grid, block, 0, f3d::stream(), o, i, g, f3d::conv_taps(), zchunk);
  else hipLaunchKernelGGL(k_gauss_z, grid, block, lds, f3d::stream(), o, i, g, f3d::conv_taps(), R, zchunk);
  F3D_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
