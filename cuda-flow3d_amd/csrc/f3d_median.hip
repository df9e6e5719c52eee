// 3-D median filter of the flow components for gfx950 (window diameter 3, 5 or 7, mirror boundary).
//
// Replaces src/kernels/median_3d.cu:49-299 of the reference: the output is the
// element of rank r^3/2 of the r^3 window gathered at mirrored indices.  The reference sorts a 343-float
// per-thread local array by insertion; any exact selection gives the same value, so here
//   r = 3, 5 : the window lives in VGPRs and goes through a Batcher odd-even merge network of
//              compare-exchanges (v_min_f32 / v_max_f32); everything that does not feed rank r^3/2 is
//              pruned at compile time by dead-code elimination (1184 of 1441 exchanges remain for r = 5),
//   r = 7    : 343 values do not fit the register file; exact rank selection by bisection on the
//              order-preserving integer image of the floats (32 counting passes over the window).
// The only observable difference to a stable sort is the sign of a zero result when the window holds both
// -0 and +0 (they compare equal); values are otherwise identical.
#include "f3d_internal.h"

#include <cstdlib>

namespace {

constexpr int kBX = 64;
constexpr int kBY = 4;

__device__ __forceinline__ void cmp_exchange(float& a, float& b)
{
  const float lo = fminf(a, b);
  const float hi = fmaxf(a, b);
  a = lo;
  b = hi;
}

// Batcher's odd-even merge sort for arbitrary N, fully unrolled over a register array.  Exchanges whose results nobody
// reads afterwards disappear at compile time, so a caller that looks at a few ranks only pays for what feeds them.
template <int N>
__device__ __forceinline__ void sort_network(float (&v)[N])
{
#pragma unroll
  for (int p = 1; p < N; p <<= 1) {
#pragma unroll
    for (int k = p; k >= 1; k >>= 1) {
#pragma unroll
      for (int j = k % p; j <= N - 1 - k; j += 2 * k) {
#pragma unroll
        for (int i = 0; i <= (k - 1 < N - j - k - 1 ? k - 1 : N - j - k - 1); ++i) {
          if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) cmp_exchange(v[i + j], v[i + j + k]);
        }
      }
    }
  }
}

template <int N>
__device__ __forceinline__ float rank_middle(float (&v)[N])
{
  sort_network(v);
  return v[N / 2];
}

// The element of rank NB (0-based) of the union of two SORTED lists a[0 .. NB] (NB + 1 values) and b[0 .. NB-1]: with
// i values taken from a and NB + 1 - i from b it is min over i of max(a[i-1], b[NB-i]) -- NB max, NB min.
template <int NB>
__device__ __forceinline__ float rank_nb_of_two_sorted(const float* a, const float* b)
{
  float r = a[NB];
#pragma unroll
  for (int i = 1; i <= NB; ++i) r = fminf(r, fmaxf(a[i - 1], b[NB - i]));
  return r;
}

// Up to three volumes of one box per launch (f3d_median_n: u, v, w of a level): grid.z = z-chunks x volumes, a workgroup finds its
// volume by subtraction (scalar work; the kernels below are register-bound and must not pay a vector division for it).
constexpr int kMaxBatch = 3;
struct MedVols {
  const float* in[kMaxBatch];
  float* out[kMaxBatch];
  int zblocks;  // z-chunks (bisection kernel: planes) per volume
};
__device__ __forceinline__ int med_volume(const MedVols& v, int& zb)
{
  int vol = 0;
  zb = static_cast<int>(blockIdx.z);
  while (zb >= v.zblocks) {
    zb -= v.zblocks;
    ++vol;
  }
  return vol;
}

// LDS-staged variant: a workgroup of 64 x 4 lanes marches along z over a chunk of planes.  The (64 + 2h) x (4 + 2h)
// mirrored footprint of every plane is fetched once into a ring of R planes in LDS (a few loads per lane and step);
// the R^3 window of a voxel is then R^3 LDS reads instead of R^3 cached global loads with 64-bit address arithmetic.
template <int R>
__global__ __launch_bounds__(kBX* kBY) void k_median_net(MedVols mv, F3dGeo g, int zchunk)
{
  int zb;
  const int vol = med_volume(mv, zb);
  const float* __restrict__ in = mv.in[vol];
  float* __restrict__ out = mv.out[vol];
  constexpr int HALF = R / 2;
  constexpr int TW = kBX + 2 * HALF, TH = kBY + 2 * HALF;
  __shared__ float ring[R][TH][TW];
  const int tid = threadIdx.y * kBX + threadIdx.x;
  const int x0 = blockIdx.x * kBX, y0 = blockIdx.y * kBY;
  const int x = x0 + threadIdx.x;
  const int y = y0 + threadIdx.y;
  const int z0 = g.z_lo + zb * zchunk;
  const int z1 = min(z0 + zchunk, g.z_hi);
  const bool owner = x < g.W && y < g.H;

  // plane zz (any integer: mirrored) -> ring slot (zz mod R), cooperatively
  auto fetch = [&](int zz) {
    const int zm = f3d_clampi(f3d_mir(zz, g.D), 0, g.D - 1);
    float(*dst)[TW] = ring[((zz % R) + R) % R];
    for (int i = tid; i < TW * TH; i += kBX * kBY) {
      const int ty = i / TW, tx = i - ty * TW;
      const int xs = f3d_clampi(f3d_mir(x0 + tx - HALF, g.W), 0, g.W - 1);
      const int ys = f3d_clampi(f3d_mir(y0 + ty - HALF, g.H), 0, g.H - 1);
      dst[ty][tx] = in[f3d_row(g, ys, zm) + xs];
    }
  };
  for (int zz = z0 - HALF; zz < z0 + HALF; ++zz) fetch(zz);
  for (int z = z0; z < z1; ++z) {
    fetch(z + HALF);
    __syncthreads();
    float v[R * R * R];
#pragma unroll
    for (int iz = 0; iz < R; ++iz) {
      const float(*pl)[TW] = ring[(((z + iz - HALF) % R) + R) % R];
#pragma unroll
      for (int iy = 0; iy < R; ++iy)
#pragma unroll
        for (int ix = 0; ix < R; ++ix) v[(iz * R + iy) * R + ix] = pl[threadIdx.y + iy][threadIdx.x + ix];
    }
    const float med = rank_middle<R * R * R>(v);
    if (owner) out[f3d_row(g, y, z) + x] = med;
    __syncthreads();  // the slot of plane z - HALF is overwritten by the next fetch
  }
}

// Two outputs per lane and step.  The windows of voxels (x, y, z) and (x, y, z+1) share R-1 of their R planes: S, (R-1) R^2
// values.  With m = (R^3 - 1) / 2 the rank looked for, an element of S can be the median of S u P (P = the R^2 values of the
// one plane a window has for itself) only if its rank inside S lies in [m - R^2, m], and the median is then the element of
// rank R^2 among those R^2 + 1 candidates and the sorted P.  So a pair costs ONE pruned sorting network over S (ranks
// m - R^2 .. m), two small full sorts and two min/max chains -- 2 524 min/max for two outputs at R = 5 against 2 x 2 244 of
// the single-output network -- and holds ~110 values at a time instead of 125 + temporaries.
template <int R>
__global__ __launch_bounds__(kBX* kBY) void k_median_pair(MedVols mv, F3dGeo g, int zchunk)
{
  int zb;
  const int vol = med_volume(mv, zb);
  const float* __restrict__ in = mv.in[vol];
  float* __restrict__ out = mv.out[vol];
  constexpr int HALF = R / 2;
  constexpr int TW = kBX + 2 * HALF, TH = kBY + 2 * HALF;
  constexpr int NS = R + 1;  // ring slots: planes z-HALF .. z+HALF+1
  constexpr int RR = R * R, NSH = (R - 1) * RR, M = (R * R * R - 1) / 2;
  __shared__ float ring[NS][TH][TW];
  const int tid = threadIdx.y * kBX + threadIdx.x;
  const int x0 = blockIdx.x * kBX, y0 = blockIdx.y * kBY;
  const int x = x0 + threadIdx.x;
  const int y = y0 + threadIdx.y;
  const int z0 = g.z_lo + zb * zchunk;
  const int z1 = min(z0 + zchunk, g.z_hi);
  const bool owner = x < g.W && y < g.H;
  const int zz_max = z1 - 1 + HALF;  // the last plane this chunk may touch (a slab window holds nothing beyond it)

  auto slot_of = [&](int zz) { return ((zz % NS) + NS) % NS; };
  auto fetch = [&](int zz) {
    const int zm = f3d_clampi(f3d_mir(min(zz, zz_max), g.D), 0, g.D - 1);
    float(*dst)[TW] = ring[slot_of(zz)];
    for (int i = tid; i < TW * TH; i += kBX * kBY) {
      const int ty = i / TW, tx = i - ty * TW;
      const int xs = f3d_clampi(f3d_mir(x0 + tx - HALF, g.W), 0, g.W - 1);
      const int ys = f3d_clampi(f3d_mir(y0 + ty - HALF, g.H), 0, g.H - 1);
      dst[ty][tx] = in[f3d_row(g, ys, zm) + xs];
    }
  };
  auto gather = [&](int zz, float* v) __attribute__((always_inline)) {
    const float(*pl)[TW] = ring[slot_of(zz)];
#pragma unroll
    for (int iy = 0; iy < R; ++iy)
#pragma unroll
      for (int ix = 0; ix < R; ++ix) v[iy * R + ix] = pl[threadIdx.y + iy][threadIdx.x + ix];
  };
  for (int zz = z0 - HALF; zz < z0 + HALF; ++zz) fetch(zz);
  for (int z = z0; z < z1; z += 2) {
    fetch(z + HALF);
    fetch(z + HALF + 1);
    __syncthreads();
    float s[NSH];
#pragma unroll
    for (int iz = 0; iz < R - 1; ++iz) gather(z - HALF + 1 + iz, s + iz * RR);
    sort_network(s);
    float p[RR];
    gather(z - HALF, p);
    sort_network(p);
    const float med_a = rank_nb_of_two_sorted<RR>(s + (M - RR), p);
    gather(z + HALF + 1, p);
    sort_network(p);
    const float med_b = rank_nb_of_two_sorted<RR>(s + (M - RR), p);
    if (owner) {
      out[f3d_row(g, y, z) + x] = med_a;
      if (z + 1 < z1) out[f3d_row(g, y, z + 1) + x] = med_b;
    }
    __syncthreads();  // the slots of planes z - HALF and z - HALF + 1 are overwritten by the next fetches
  }
}

#include "f3d_median_nets.h"

// 5^3 with the SORTED planes kept: a lane holds the sorted 25-lists of the four newest planes in registers, so a pair of
// outputs costs two fresh plane sorts (2 x 280 min/max), two merges of neighbouring plane lists (2 x 238), the pruned merge
// that yields the 26 candidates (244) and the two selection chains (2 x 50): 1 380 for two outputs against 2 524 with the
// sort of the raw 100 in k_median_pair.  Planes are named Q0 .. Q5 = z-2 .. z+3; a step consumes the lists of Q0 and Q1 and
// produces those of Q4 and Q5, so six register arrays rotate by two per step and the march is unrolled three steps deep to
// make the rotation a renaming.  ~200 live values at the peak: two waves per SIMD, like the single-output network.
__global__ __launch_bounds__(kBX* kBY) void k_median_keep(MedVols mv, F3dGeo g, int zchunk)
{
  int zb;
  const int vol = med_volume(mv, zb);
  const float* __restrict__ in = mv.in[vol];
  float* __restrict__ out = mv.out[vol];
  constexpr int R = 5, HALF = 2;
  constexpr int TW = kBX + 2 * HALF, TH = kBY + 2 * HALF;
  constexpr int NS = R + 1;
  __shared__ float ring[NS][TH][TW];
  const int tid = threadIdx.y * kBX + threadIdx.x;
  const int x0 = blockIdx.x * kBX, y0 = blockIdx.y * kBY;
  const int x = x0 + threadIdx.x;
  const int y = y0 + threadIdx.y;
  const int z0 = g.z_lo + zb * zchunk;
  const int z1 = min(z0 + zchunk, g.z_hi);
  const bool owner = x < g.W && y < g.H;
  const int zz_max = z1 - 1 + HALF;

  auto slot_of = [&](int zz) { return ((zz % NS) + NS) % NS; };
  auto fetch = [&](int zz) {
    const int zm = f3d_clampi(f3d_mir(min(zz, zz_max), g.D), 0, g.D - 1);
    float(*dst)[TW] = ring[slot_of(zz)];
    for (int i = tid; i < TW * TH; i += kBX * kBY) {
      const int ty = i / TW, tx = i - ty * TW;
      const int xs = f3d_clampi(f3d_mir(x0 + tx - HALF, g.W), 0, g.W - 1);
      const int ys = f3d_clampi(f3d_mir(y0 + ty - HALF, g.H), 0, g.H - 1);
      dst[ty][tx] = in[f3d_row(g, ys, zm) + xs];
    }
  };
  auto sorted_plane = [&](int zz, float (&v)[25]) __attribute__((always_inline)) {
    const float(*pl)[TW] = ring[slot_of(zz)];
#pragma unroll
    for (int iy = 0; iy < R; ++iy)
#pragma unroll
      for (int ix = 0; ix < R; ++ix) v[iy * R + ix] = pl[threadIdx.y + iy][threadIdx.x + ix];
    sort_network(v);
  };
  // one pair of outputs: q0 .. q3 come in sorted, q4 and q5 are made here
  auto step = [&](const float (&q0)[25], const float (&q1)[25], const float (&q2)[25], const float (&q3)[25], float (&q4)[25],
                  float (&q5)[25], int z) __attribute__((always_inline)) {
    fetch(z + HALF);
    fetch(z + HALF + 1);
    __syncthreads();
    sorted_plane(z + HALF, q4);
    float c[26];
    {
      float ab[50], cd[50];
      merge25(q1, q2, ab);
      merge25(q3, q4, cd);
      candidates(ab, cd, c);
    }
    const float med_a = rank_nb_of_two_sorted<25>(c, q0);
    sorted_plane(z + HALF + 1, q5);
    const float med_b = rank_nb_of_two_sorted<25>(c, q5);
    if (owner) {
      out[f3d_row(g, y, z) + x] = med_a;
      if (z + 1 < z1) out[f3d_row(g, y, z + 1) + x] = med_b;
    }
    __syncthreads();
  };
  float L0[25], L1[25], L2[25], L3[25], L4[25], L5[25];
  for (int zz = z0 - HALF; zz < z0 + HALF; ++zz) fetch(zz);
  __syncthreads();
  sorted_plane(z0 - 2, L0);
  sorted_plane(z0 - 1, L1);
  sorted_plane(z0, L2);
  sorted_plane(z0 + 1, L3);
  for (int z = z0; z < z1; z += 6) {
    step(L0, L1, L2, L3, L4, L5, z);
    if (z + 2 >= z1) break;
    step(L2, L3, L4, L5, L0, L1, z + 2);
    if (z + 4 >= z1) break;
    step(L4, L5, L0, L1, L2, L3, z + 4);
  }
}

__device__ __forceinline__ unsigned order_key(float f)
{
  const unsigned b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// Exact rank selection without holding the window: the answer is the largest T with #(key < T) <= rank.
template <int R>
__global__ __launch_bounds__(kBX* kBY) void k_median_bisect(MedVols mv, F3dGeo g)
{
  int zb;
  const int vol = med_volume(mv, zb);
  const float* __restrict__ in = mv.in[vol];
  float* __restrict__ out = mv.out[vol];
  constexpr int HALF = R / 2;
  constexpr int RANK = (R * R * R) / 2;
  const int x = blockIdx.x * kBX + threadIdx.x;
  const int y = blockIdx.y * kBY + threadIdx.y;
  const int z = g.z_lo + zb;
  if (x >= g.W || y >= g.H) return;
  int xs[R];
#pragma unroll
  for (int i = 0; i < R; ++i) xs[i] = f3d_mir(x + i - HALF, g.W);
  unsigned prefix = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned cand = prefix | (1u << bit);
    int below = 0;
    for (int iz = 0; iz < R; ++iz) {
      const int zz = f3d_mir(z + iz - HALF, g.D);
      for (int iy = 0; iy < R; ++iy) {
        const size_t row = f3d_row(g, f3d_mir(y + iy - HALF, g.H), zz);
#pragma unroll
        for (int ix = 0; ix < R; ++ix) below += order_key(in[row + xs[ix]]) < cand ? 1 : 0;
      }
    }
    if (below <= RANK) prefix = cand;
  }
  const unsigned b = (prefix >> 31) ? (prefix ^ 0x80000000u) : ~prefix;
  out[f3d_row(g, y, z) + x] = __uint_as_float(b);
}

}  // namespace

static int median_launch(const f3d_devptr* inputs, size_t count, size_t width, size_t height, size_t depth, size_t radius,
                         const f3d_devptr* outputs, const f3d_slab* slab, const char* who)
{
  F3D_REQUIRE_READY(who);
  if (!inputs || !outputs) return f3d::fail("%s: null argument", who);
  if (count == 0 || count > static_cast<size_t>(kMaxBatch)) return f3d::fail("%s: %zu volumes (one launch takes 1 .. %d)", who, count, kMaxBatch);
  for (size_t i = 0; i < count; ++i)
    for (size_t j = 0; j < count; ++j)
      if (inputs[i] == outputs[j]) return f3d::fail("%s: input buffer cannot serve as output buffer", who);
  if (radius != 3 && radius != 5 && radius != 7)
    return f3d::fail("%s: wrong median radius (%zu). Supported values: 3, 5, 7", who, radius);
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, who)) return 1;
  const int half = static_cast<int>(radius) / 2;
  if (g.W <= half || g.H <= half || g.D <= half)
    return f3d::fail("%s: every dimension must exceed radius/2 = %d for the mirror boundary", who, half);
  if (g.z_lo == g.z_hi) return 0;
  {
    const int dc = static_cast<int>(f3d::container().depth);
    int lo = g.z_lo - half < 0 ? 0 : g.z_lo - half;
    int hi = g.z_hi + half > g.D ? g.D : g.z_hi + half;
    if (g.z_lo - half < 0 && half + 1 > hi) hi = half + 1;
    if (g.z_hi + half > g.D && g.D - 1 - half < lo) lo = g.D - 1 - half;
    if (lo < g.z_base || hi - g.z_base > dc)
      return f3d::fail("%s: planes [%d,%d) needed but the container holds [%d,%d)", who, lo, hi, g.z_base, g.z_base + dc);
  }
  const dim3 block(kBX, kBY, 1);
  MedVols mv = {};
  for (size_t i = 0; i < count; ++i) {
    mv.in[i] = f3d_ptr<const float>(inputs[i]);
    mv.out[i] = f3d_ptr<float>(outputs[i]);
  }
  const unsigned nvol = static_cast<unsigned>(count);
  if (count > 1 && static_cast<long>(g.z_hi - g.z_lo) * static_cast<long>(count) > 65535L) {  // grid.z: volume by volume
    for (size_t i = 0; i < count; ++i)
      if (int e = median_launch(inputs + i, 1, width, height, depth, radius, outputs + i, slab, who)) return e;
    return 0;
  }
  if (radius == 7) {
    mv.zblocks = g.z_hi - g.z_lo;
    const dim3 grid((g.W + kBX - 1) / kBX, (g.H + kBY - 1) / kBY, mv.zblocks * nvol);
    hipLaunchKernelGGL(k_median_bisect<7>, grid, block, 0, f3d::stream(), mv, g);
  } else {
    const int planes = g.z_hi - g.z_lo;
    // workgroups of ONE z-chunk layer of the launch: the volumes of a batch fill the rounds together
    const long tiles = static_cast<long>((g.W + kBX - 1) / kBX) * ((g.H + kBY - 1) / kBY) * static_cast<long>(count);
    // F3D_MEDIAN_PAIR: 0 = the one-output-per-step network, 1 = k_median_pair, 2 = k_median_keep (timing comparisons);
    // unset = whichever the model below prefers.  Read per call (a launch costs far more) so that tests can switch it.
    const char* forced_env = std::getenv("F3D_MEDIAN_PAIR");
    const int forced = forced_env ? std::atoi(forced_env) : -1;
    if (forced == 0) {
      // z-chunks: enough workgroups to fill the chip, long enough to amortise the ring prologue
      long nz = (8192 + tiles - 1) / tiles;
      if (nz > planes) nz = planes;
      if (nz < 1) nz = 1;
      const int zchunk = static_cast<int>((planes + nz - 1) / nz);
      mv.zblocks = (planes + zchunk - 1) / zchunk;
      const dim3 grid((g.W + kBX - 1) / kBX, (g.H + kBY - 1) / kBY, mv.zblocks * nvol);
      if (radius == 3) hipLaunchKernelGGL(k_median_net<3>, grid, block, 0, f3d::stream(), mv, g, zchunk);
      if (radius == 5) hipLaunchKernelGGL(k_median_net<5>, grid, block, 0, f3d::stream(), mv, g, zchunk);
    } else {
      // Both kernels are bound by the vector unit: a chunk of zc planes costs its min/max count (per pair of planes 2 600
      // resp. 1 430, plus the four plane sorts k_median_keep starts with and a few hundred for the ring prologue), a
      // workgroup puts one wave on each SIMD of its CU, so 256 workgroups make a round whatever the occupancy, and below
      // two waves per SIMD nothing hides the LDS latency (x 1.3, measured).  Even chunks, so that only the last chunk of
      // an odd range computes a plane for nothing.  (512^3: keep 3.6 ms, pair 5.1-5.5 ms, single-output network 9.0 ms.)
      const bool can_keep = radius == 5 && forced != 1;
      const bool can_pair = !(radius == 5 && forced == 2);
      const long pair_ops = radius == 5 ? 2600 : 330;
      int zchunk = 2;
      bool keep = false;
      long best = -1;
      for (int zc = 2; zc <= planes + 1; zc += 2) {
        const long wgs = tiles * ((planes + zc - 1) / zc);
        const long rounds = (wgs + 255) / 256, thin = wgs < 512 ? 13 : 10;
        const long cost_pair = rounds * thin * (300 + (zc / 2) * pair_ops);
        const long cost_keep = rounds * thin * (300 + 1200 + (zc / 2) * 1430);
        if (can_pair && (best < 0 || cost_pair < best)) {
          best = cost_pair;
          zchunk = zc;
          keep = false;
        }
        if (can_keep && (best < 0 || cost_keep < best)) {
          best = cost_keep;
          zchunk = zc;
          keep = true;
        }
      }
      mv.zblocks = (planes + zchunk - 1) / zchunk;
      const dim3 grid((g.W + kBX - 1) / kBX, (g.H + kBY - 1) / kBY, mv.zblocks * nvol);
      if (radius == 3) hipLaunchKernelGGL(k_median_pair<3>, grid, block, 0, f3d::stream(), mv, g, zchunk);
      if (radius == 5 && keep) hipLaunchKernelGGL(k_median_keep, grid, block, 0, f3d::stream(), mv, g, zchunk);
      if (radius == 5 && !keep) hipLaunchKernelGGL(k_median_pair<5>, grid, block, 0, f3d::stream(), mv, g, zchunk);
    }
  }
  F3D_HIP(hipGetLastError());
  return 0;
}

extern "C" int f3d_median(f3d_devptr input, size_t width, size_t height, size_t depth, size_t radius, f3d_devptr output,
                          const f3d_slab* slab)
{
  return median_launch(&input, 1, width, height, depth, radius, &output, slab, "f3d_median");
}

extern "C" int f3d_median_n(const f3d_devptr* inputs, size_t count, size_t width, size_t height, size_t depth, size_t radius,
                            const f3d_devptr* outputs, const f3d_slab* slab)
{
  return median_launch(inputs, count, width, height, depth, radius, outputs, slab, "f3d_median_n");
}
