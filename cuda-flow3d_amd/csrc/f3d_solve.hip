// Solver kernels for gfx950: robust-penalty weights (phi, ksi) and the Jacobi / in-voxel Gauss-Seidel sweep.
//
// Replaces src/kernels/solve_3d.cu (compute_phi_ksi_3d :33-262, solve_3d :264-508) of the reference.
// Same per-voxel expression trees, built with -ffp-contract=off so that every
// + - * / sqrt is one correctly rounded IEEE binary32 operation; the data movement is redesigned for CDNA4:
//
//   * one wave64 = one 64-float row segment (256 B coalesced per load).  Lanes 1..62 own an output voxel,
//     lanes 0 and 63 only carry the x-halo: x-neighbours come from DPP wave shifts, not from memory or LDS.
//   * a workgroup is F3D_TY such rows adjacent in y and marches along z with a rolling register window
//     (z-1, z, z+1, and z+2 in flight), so every plane of every input is pulled from HBM once per sweep
//     (2.5-D blocking) instead of the reference's (16+2)(8+2)(4+2) LDS tile with 2.25x halo over-fetch.
//   * the y-neighbour rows are the rows the adjacent waves of the same workgroup stream at the same
//     moment, so they are served by the CU's L1/the XCD's L2; mirror (Neumann) halos are index arithmetic.
//   * loads for the next z step are issued before the arithmetic of the current one (software pipelining).
#include "f3d_internal.h"

namespace {

constexpr int kLanes = 64;
constexpr int kOutX = 62;  // output voxels per wave row (lanes 1..62)
constexpr int kTY = 4;     // rows (waves) per workgroup

// value held by the lane to the left / right (wave-wide shift by one lane; edge lanes keep their own value)
__device__ __forceinline__ float lane_left(float v)
{
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_right(float v)
{
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

struct SolveArgs {
  const float* in[10];  // f0, f1(warped), u, v, w, du, dv, dw, phi, ksi
  float* out[3];        // sweep: temp_du, temp_dv, temp_dw; phi_ksi: phi, ksi
  float hx, hy, hz;
  float p0, p1;  // sweep: alpha, unused; phi_ksi: eps_smooth, eps_data
};

enum { F0 = 0, F1 = 1, U = 2, V = 3, Wf = 4, DU = 5, DV = 6, DW = 7, PHI = 8 };

// One 7-point neighbourhood of the NA stencilled inputs, all in registers.
template <int NA>
struct Hood {
  float c[NA], xm[NA], xp[NA], ym[NA], yp[NA], zm[NA], zp[NA];
};

// A.3: src/kernels/solve_3d.cu:177-260
__device__ __forceinline__ void phi_ksi_voxel(const Hood<8>& n, float hx, float hy, float hz, float eps_s,
                                              float eps_d, float& phi, float& ksi)
{
  const float dux = (n.xp[U] - n.xm[U] + n.xp[DU] - n.xm[DU]) / (2.f * hx);
  const float duy = (n.yp[U] - n.ym[U] + n.yp[DU] - n.ym[DU]) / (2.f * hy);
  const float duz = (n.zp[U] - n.zm[U] + n.zp[DU] - n.zm[DU]) / (2.f * hz);
  const float dvx = (n.xp[V] - n.xm[V] + n.xp[DV] - n.xm[DV]) / (2.f * hx);
  const float dvy = (n.yp[V] - n.ym[V] + n.yp[DV] - n.ym[DV]) / (2.f * hy);
  const float dvz = (n.zp[V] - n.zm[V] + n.zp[DV] - n.zm[DV]) / (2.f * hz);
  const float dwx = (n.xp[Wf] - n.xm[Wf] + n.xp[DW] - n.xm[DW]) / (2.f * hx);
  const float dwy = (n.yp[Wf] - n.ym[Wf] + n.yp[DW] - n.ym[DW]) / (2.f * hy);
  const float dwz = (n.zp[Wf] - n.zm[Wf] + n.zp[DW] - n.zm[DW]) / (2.f * hz);

  phi = 1.f / (2.f * sqrtf(dux * dux + duy * duy + duz * duz + dvx * dvx + dvy * dvy + dvz * dvz + dwx * dwx +
                           dwy * dwy + dwz * dwz + eps_s * eps_s));

  const float fx = (n.xp[F0] - n.xm[F0] + n.xp[F1] - n.xm[F1]) / (4.f * hx);
  const float fy = (n.yp[F0] - n.ym[F0] + n.yp[F1] - n.ym[F1]) / (4.f * hy);
  const float fz = (n.zp[F0] - n.zm[F0] + n.zp[F1] - n.zm[F1]) / (4.f * hz);
  const float ft = n.c[F1] - n.c[F0];

  const float J11 = fx * fx, J22 = fy * fy, J33 = fz * fz;
  const float J12 = fx * fy, J13 = fx * fz, J23 = fy * fz;
  const float J14 = fx * ft, J24 = fy * ft, J34 = fz * ft, J44 = ft * ft;

  const float du = n.c[DU], dv = n.c[DV], dw = n.c[DW];
  float s = (J11 * du + J12 * dv + J13 * dw + J14) * du + (J12 * du + J22 * dv + J23 * dw + J24) * dv +
            (J13 * du + J23 * dv + J33 * dw + J34) * dw + (J14 * du + J24 * dv + J34 * dw + J44);
  s = static_cast<float>(s > 0) * s;
  ksi = 1.f / (2.f * sqrtf(s + eps_d * eps_d));
}

// A.4: src/kernels/solve_3d.cu:425-506
__device__ __forceinline__ void sweep_voxel(const Hood<9>& n, float ksi, float hx, float hy, float hz, float alpha,
                                            bool has_xp, bool has_xm, bool has_yp, bool has_ym, bool has_zp,
                                            bool has_zm, float& r_du, float& r_dv, float& r_dw)
{
  const float fx = (n.xp[F0] - n.xm[F0] + n.xp[F1] - n.xm[F1]) / (4.f * hx);
  const float fy = (n.yp[F0] - n.ym[F0] + n.yp[F1] - n.ym[F1]) / (4.f * hy);
  const float fz = (n.zp[F0] - n.zm[F0] + n.zp[F1] - n.zm[F1]) / (4.f * hz);
  const float ft = n.c[F1] - n.c[F0];

  const float J11 = fx * fx, J22 = fy * fy, J33 = fz * fz;
  const float J12 = fx * fy, J13 = fx * fz, J23 = fy * fz;
  const float J14 = fx * ft, J24 = fy * ft, J34 = fz * ft;

  const float hx_2 = alpha / (hx * hx);
  const float hy_2 = alpha / (hy * hy);
  const float hz_2 = alpha / (hz * hz);

  const float xp = static_cast<float>(has_xp) * hx_2;
  const float xm = static_cast<float>(has_xm) * hx_2;
  const float yp = static_cast<float>(has_yp) * hy_2;
  const float ym = static_cast<float>(has_ym) * hy_2;
  const float zp = static_cast<float>(has_zp) * hz_2;
  const float zm = static_cast<float>(has_zm) * hz_2;

  const float phi_xp = (n.xp[PHI] + n.c[PHI]) / 2.f;
  const float phi_xm = (n.xm[PHI] + n.c[PHI]) / 2.f;
  const float phi_yp = (n.yp[PHI] + n.c[PHI]) / 2.f;
  const float phi_ym = (n.ym[PHI] + n.c[PHI]) / 2.f;
  const float phi_zp = (n.zp[PHI] + n.c[PHI]) / 2.f;
  const float phi_zm = (n.zm[PHI] + n.c[PHI]) / 2.f;

  const float sumH = (xp * phi_xp + xm * phi_xm + yp * phi_yp + ym * phi_ym + zp * phi_zp + zm * phi_zm);
  const float sumU = phi_xp * xp * (n.xp[U] + n.xp[DU] - n.c[U]) + phi_xm * xm * (n.xm[U] + n.xm[DU] - n.c[U]) +
                     phi_yp * yp * (n.yp[U] + n.yp[DU] - n.c[U]) + phi_ym * ym * (n.ym[U] + n.ym[DU] - n.c[U]) +
                     phi_zp * zp * (n.zp[U] + n.zp[DU] - n.c[U]) + phi_zm * zm * (n.zm[U] + n.zm[DU] - n.c[U]);
  const float sumV = phi_xp * xp * (n.xp[V] + n.xp[DV] - n.c[V]) + phi_xm * xm * (n.xm[V] + n.xm[DV] - n.c[V]) +
                     phi_yp * yp * (n.yp[V] + n.yp[DV] - n.c[V]) + phi_ym * ym * (n.ym[V] + n.ym[DV] - n.c[V]) +
                     phi_zp * zp * (n.zp[V] + n.zp[DV] - n.c[V]) + phi_zm * zm * (n.zm[V] + n.zm[DV] - n.c[V]);
  const float sumW = phi_xp * xp * (n.xp[Wf] + n.xp[DW] - n.c[Wf]) + phi_xm * xm * (n.xm[Wf] + n.xm[DW] - n.c[Wf]) +
                     phi_yp * yp * (n.yp[Wf] + n.yp[DW] - n.c[Wf]) + phi_ym * ym * (n.ym[Wf] + n.ym[DW] - n.c[Wf]) +
                     phi_zp * zp * (n.zp[Wf] + n.zp[DW] - n.c[Wf]) + phi_zm * zm * (n.zm[Wf] + n.zm[DW] - n.c[Wf]);

  r_du = (ksi * (-J14 - J12 * n.c[DV] - J13 * n.c[DW]) + sumU) / (ksi * J11 + sumH);
  r_dv = (ksi * (-J24 - J12 * r_du - J23 * n.c[DW]) + sumV) / (ksi * J22 + sumH);
  r_dw = (ksi * (-J34 - J13 * r_du - J23 * r_dv) + sumW) / (ksi * J33 + sumH);
}

// SWEEP = true: solve sweep (9 stencilled inputs + ksi, 3 outputs); false: phi/ksi (8 inputs, 2 outputs).
template <bool SWEEP>
__global__ __launch_bounds__(kLanes* kTY) void k_solver(SolveArgs a, F3dGeo g, int zchunk)
{
  constexpr int NA = SWEEP ? 9 : 8;
  const int lane = threadIdx.x;
  const int y = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.y) * kTY + static_cast<int>(threadIdx.y));
  if (y >= g.H) return;
  const int z0 = g.z_lo + static_cast<int>(blockIdx.z) * zchunk;
  const int z1 = min(z0 + zchunk, g.z_hi);
  if (z0 >= z1) return;

  const int x = static_cast<int>(blockIdx.x) * kOutX - 1 + lane;
  const int xi = f3d_clampi(f3d_mir(x, g.W), 0, g.W - 1);  // x-halo lanes read the mirrored column
  const int ylo = f3d_mir(y - 1, g.H);
  const int yhi = f3d_mir(y + 1, g.H);
  const bool owner = lane >= 1 && lane <= kOutX && x < g.W;

  // rolling window: m = plane z-1, c = z, p = z+1, q = z+2 (in flight); yl/yh = rows y-1/y+1 of plane z
  float m[NA], c[NA], p[NA], q[NA], yl[NA], yh[NA], nyl[NA], nyh[NA];
  float kc = 0.f, kn = 0.f;
  {
    const size_t rm = f3d_row(g, y, f3d_mir(z0 - 1, g.D)) + xi;
    const size_t rc = f3d_row(g, y, z0) + xi;
    const size_t rp = f3d_row(g, y, f3d_mir(z0 + 1, g.D)) + xi;
    const size_t rl = f3d_row(g, ylo, z0) + xi;
    const size_t rh = f3d_row(g, yhi, z0) + xi;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      m[i] = a.in[i][rm];
      c[i] = a.in[i][rc];
      p[i] = a.in[i][rp];
      yl[i] = a.in[i][rl];
      yh[i] = a.in[i][rh];
    }
    if (SWEEP) kc = a.in[9][rc];
  }

  for (int z = z0; z < z1; ++z) {
    const bool more = z + 1 < z1;
    if (more) {
      const size_t rq = f3d_row(g, y, f3d_mir(z + 2, g.D)) + xi;
      const size_t rl = f3d_row(g, ylo, z + 1) + xi;
      const size_t rh = f3d_row(g, yhi, z + 1) + xi;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        q[i] = a.in[i][rq];
        nyl[i] = a.in[i][rl];
        nyh[i] = a.in[i][rh];
      }
      if (SWEEP) kn = a.in[9][f3d_row(g, y, z + 1) + xi];
    }

    Hood<NA> n;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      n.c[i] = c[i];
      n.xm[i] = lane_left(c[i]);
      n.xp[i] = lane_right(c[i]);
      n.ym[i] = yl[i];
      n.yp[i] = yh[i];
      n.zm[i] = m[i];
      n.zp[i] = p[i];
    }

    const size_t o = f3d_row(g, y, z) + xi;
    if constexpr (SWEEP) {
      float r_du, r_dv, r_dw;
      sweep_voxel(n, kc, a.hx, a.hy, a.hz, a.p0, x < g.W - 1, x > 0, y < g.H - 1, y > 0, z < g.D - 1, z > 0, r_du,
                  r_dv, r_dw);
      if (owner) {
        a.out[0][o] = r_du;
        a.out[1][o] = r_dv;
        a.out[2][o] = r_dw;
      }
    } else {
      float phi, ksi;
      phi_ksi_voxel(n, a.hx, a.hy, a.hz, a.p0, a.p1, phi, ksi);
      if (owner) {
        a.out[0][o] = phi;
        a.out[1][o] = ksi;
      }
    }

    if (more) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        m[i] = c[i];
        c[i] = p[i];
        p[i] = q[i];
        yl[i] = nyl[i];
        yh[i] = nyh[i];
      }
      kc = kn;
    }
  }
}

// z-chunks so that even a coarse pyramid level spreads over all 256 CUs
int pick_zchunk(const F3dGeo& g, dim3* grid)
{
  const int ntx = (g.W + kOutX - 1) / kOutX;
  const int nty = (g.H + kTY - 1) / kTY;
  const int planes = g.z_hi - g.z_lo;
  const long want_wg = 2048;
  long nzc = (want_wg + static_cast<long>(ntx) * nty - 1) / (static_cast<long>(ntx) * nty);
  if (nzc < 1) nzc = 1;
  long max_chunks = planes / 4 > 0 ? planes / 4 : 1;
  if (nzc > max_chunks) nzc = max_chunks;
  int zchunk = static_cast<int>((planes + nzc - 1) / nzc);
  int nz = (planes + zchunk - 1) / zchunk;
  *grid = dim3(ntx, nty, nz);
  return zchunk;
}

bool slab_reach_ok(const F3dGeo& g, int reach, const char* who)
{
  const int dc = static_cast<int>(f3d::container().depth);
  const int lo = g.z_lo - reach < 0 ? 0 : g.z_lo - reach;
  const int hi = g.z_hi + reach > g.D ? g.D : g.z_hi + reach;  // one past
  int need_lo = lo, need_hi = hi;
  if (g.z_lo - reach < 0 && reach + 1 > need_hi) need_hi = reach + 1 > g.D ? g.D : reach + 1;  // mirror targets 1..reach
  if (g.z_hi + reach > g.D && g.D - 1 - reach < need_lo) need_lo = g.D - 1 - reach < 0 ? 0 : g.D - 1 - reach;
  if (need_lo < g.z_base || need_hi - g.z_base > dc) {
    f3d::fail("%s: planes [%d,%d) needed but the container holds [%d,%d)", who, need_lo, need_hi, g.z_base,
              g.z_base + dc);
    return false;
  }
  return true;
}

}  // namespace

extern "C" {

int f3d_phi_ksi(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, size_t width, size_t height,
                size_t depth, float hx, float hy, float hz, float equation_smoothness, float equation_data,
                f3d_devptr phi, f3d_devptr ksi, const f3d_slab* slab)
{
  F3D_REQUIRE_READY("f3d_phi_ksi");
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, "f3d_phi_ksi")) return 1;
  if (g.W < 2 || g.H < 2 || g.D < 2) return f3d::fail("f3d_phi_ksi: every dimension must be at least 2");
  if (g.z_lo == g.z_hi) return 0;
  if (!slab_reach_ok(g, 1, "f3d_phi_ksi")) return 1;
  SolveArgs a;
  const f3d_devptr in[8] = {frame_0, frame_1, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw};
  for (int i = 0; i < 8; ++i) a.in[i] = f3d_ptr<const float>(in[i]);
  a.in[8] = a.in[9] = nullptr;
  a.out[0] = f3d_ptr<float>(phi);
  a.out[1] = f3d_ptr<float>(ksi);
  a.out[2] = nullptr;
  a.hx = hx; a.hy = hy; a.hz = hz;
  a.p0 = equation_smoothness;
  a.p1 = equation_data;
  dim3 grid;
  const int zchunk = pick_zchunk(g, &grid);
  f3d::prof_begin(F3D_K_PHI_KSI, static_cast<size_t>(g.W) * g.H * (g.z_hi - g.z_lo));
  hipLaunchKernelGGL(k_solver<false>, grid, dim3(kLanes, kTY, 1), 0, f3d::stream(), a, g, zchunk);
  f3d::prof_end(F3D_K_PHI_KSI);
  F3D_HIP(hipGetLastError());
  return 0;
}

int f3d_solve_sweep(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                    f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi,
                    size_t width, size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha,
                    f3d_devptr temp_du, f3d_devptr temp_dv, f3d_devptr temp_dw, const f3d_slab* slab)
{
  F3D_REQUIRE_READY("f3d_solve_sweep");
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, "f3d_solve_sweep")) return 1;
  if (g.W < 2 || g.H < 2 || g.D < 2) return f3d::fail("f3d_solve_sweep: every dimension must be at least 2");
  if (g.z_lo == g.z_hi) return 0;
  if (!slab_reach_ok(g, 1, "f3d_solve_sweep")) return 1;
  SolveArgs a;
  const f3d_devptr in[10] = {frame_0, frame_1, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw, phi, ksi};
  for (int i = 0; i < 10; ++i) a.in[i] = f3d_ptr<const float>(in[i]);
  a.out[0] = f3d_ptr<float>(temp_du);
  a.out[1] = f3d_ptr<float>(temp_dv);
  a.out[2] = f3d_ptr<float>(temp_dw);
  a.hx = hx; a.hy = hy; a.hz = hz;
  a.p0 = equation_alpha;
  a.p1 = 0.f;
  dim3 grid;
  const int zchunk = pick_zchunk(g, &grid);
  f3d::prof_begin(F3D_K_SWEEP, static_cast<size_t>(g.W) * g.H * (g.z_hi - g.z_lo));
  hipLaunchKernelGGL(k_solver<true>, grid, dim3(kLanes, kTY, 1), 0, f3d::stream(), a, g, zchunk);
  f3d::prof_end(F3D_K_SWEEP);
  F3D_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
