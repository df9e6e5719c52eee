// Solver kernels for gfx950: robust-penalty weights (phi, ksi) and the Jacobi / in-voxel Gauss-Seidel sweep.
//
// Replaces src/kernels/solve_3d.cu (compute_phi_ksi_3d :33-262, solve_3d :264-508) of the reference.
// Same per-voxel expression trees, built with -ffp-contract=off so that every
// + - * / sqrt is one correctly rounded IEEE binary32 operation; the data movement is redesigned for CDNA4:
//
//   * one wave64 = one 64-float row segment of a 64-aligned x tile (256 B per load instruction); x-neighbours come
//     from DPP wave shifts with the tile's halo column merged in as the value the edge lane receives;
//   * a workgroup is 8 (one sweep, phi/ksi) or 9+2 (two fused sweeps) such rows adjacent in y and marches along z with
//     rotating plane register sets (z-1, z, z+1 and the planes in flight), so every plane of every input is pulled
//     from HBM once per launch (2.5-D blocking) instead of the reference's (16+2)(8+2)(4+2) LDS tile with 2.25x halo
//     over-fetch; y-neighbours go through a double-buffered LDS image of the current plane, one barrier per z step;
//   * y-halo rows and x-halo columns are fetched by LDS-DMA (`global_load_lds_dword`) into small rings, at the moment
//     the owning neighbour tile streams the same plane, so they hit the XCD's L2; tiles are dealt to the 8 XCDs in
//     contiguous runs; mirror (Neumann) halos are index arithmetic;
//   * every memory instruction of the hot loops is issued by hand (inline assembly) with counted `s_waitcnt vmcnt(N)`,
//     so loads for later planes stay in flight across the arithmetic of the current one.
// Kernels: k_phiksi6 (A.3), k_sweep6 (one sweep, A.4), k_sweep7 (two consecutive sweeps in one launch).  The ladder
// of earlier variants (register rows, LDS rows, buffer loads, halos two planes ahead, ...) is in DESIGN.md section 3
// and in the git history.
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "f3d_internal.h"

namespace {

constexpr int kLanes = 64;

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
  unsigned long long* probe = nullptr;  // timing experiments (k_sweep7 with ABL bit 3): [wave][phase] cycle sums
  int plain_division = 0;               // timing experiments (F3D_UDIV=0): every division the ordinary IEEE sequence
  // k_phiksi6 on TWO windows of the same container in one launch (f3d_phi_ksi_zones): z-chunks nz_first and up belong to
  // [z2_lo, z2_hi); none when z2_hi <= z2_lo
  int z2_lo = 0, z2_hi = 0, nz_first = 0;
};

enum { F0 = 0, F1 = 1, U = 2, V = 3, Wf = 4, DU = 5, DV = 6, DW = 7, PHI = 8 };

// One 7-point neighbourhood of the NA stencilled inputs, all in registers.
template <int NA>
struct Hood {
  float c[NA], xm[NA], xp[NA], ym[NA], yp[NA], zm[NA], zp[NA];
};

// ---- exact division by a wave-uniform divisor ----------------------------------------------------------------------------
// The derivatives of A.3 / A.4 divide per-voxel numerators by 2h or 4h.  IEEE binary32 division is an 11-instruction
// sequence with a quarter-rate reciprocal (19.5 ns per wave and SIMD, tools/lab/valu_rate); phi/ksi does twelve of them.
// For a divisor d shared by the wave, q = (float)((double)x * R), R = RN64(1 / (double)d), is the SAME float:
//   * x * R differs from x / d by at most 2^-52 relatively (two roundings to binary64);
//   * a quotient of two binary32 numbers is never a rounding boundary of binary32 (a midpoint m has an odd 25-bit
//     significand; x = m * d would need more than 24 significant bits), and it stays at least 2^-49 * |q| away from
//     every boundary (x - m * d is a non-zero multiple of 2^(min exponent), |x| * 2^-49 at least);
//   so no boundary lies between the exact quotient and the computed double, and both round to the same binary32.
// The argument needs a normal result: a quotient in the subnormal range CAN be a tie.  Numerators with 0 < |x| < 2^-100
// (and divisors outside 2^-20 .. 2^20) therefore send the whole wave through the ordinary division; +-0, Inf and NaN are
// fine.  Checked on 2 x 10^8 random and adversarial (near-midpoint) numerators against binary32 division: no difference.
// Cost 7.5 ns instead of 19.5 ns, plus two integer instructions per numerator for the guard.
struct UDiv {
  double r;  // RN64(1 / d)
  float d;
};
__device__ __forceinline__ UDiv make_udiv(float d)
{
  UDiv u;
  u.d = d;
  const double r = 1.0 / static_cast<double>(d);
  // the divisor is wave-uniform: keep the reciprocal in scalar registers
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(__double_as_longlong(r)));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(__double_as_longlong(r) >> 32));
  u.r = __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
  return u;
}
__device__ __forceinline__ bool udiv_divisor_ok(float d) { return d >= 0x1p-20f && d <= 0x1p20f; }
__device__ __forceinline__ float udiv(float x, const UDiv& u) { return static_cast<float>(static_cast<double>(x) * u.r); }
// key(x) = (bits << 1) - 1 as unsigned: +-0 -> 0xffffffff, tiny non-zero -> small, anything >= 2^-100 -> >= kUdivSafe
constexpr unsigned kUdivSafe = (0x0d800000u << 1) - 1u;
__device__ __forceinline__ unsigned udiv_key(float x) { return (__float_as_uint(x) << 1) - 1u; }
template <int N>
__device__ __forceinline__ bool udiv_all_safe(const float (&num)[N])
{
  unsigned k = 0xffffffffu;
#pragma unroll
  for (int i = 0; i < N; ++i) k = min(k, udiv_key(num[i]));
  return __builtin_amdgcn_ballot_w64(k < kUdivSafe) == 0;  // wave-uniform
}

// fx, fy, fz of A.3 / A.4 from their numerators
struct FDivs {
  UDiv x4, y4, z4;
  bool ok;
};
__device__ __forceinline__ FDivs make_f_divs(float hx, float hy, float hz)
{
  FDivs s;
  s.x4 = make_udiv(4.f * hx); s.y4 = make_udiv(4.f * hy); s.z4 = make_udiv(4.f * hz);
  s.ok = udiv_divisor_ok(s.x4.d) && udiv_divisor_ok(s.y4.d) && udiv_divisor_ok(s.z4.d);
  return s;
}
__device__ __forceinline__ void f_derivatives(float (&q)[3], const FDivs& dv)
{
  if (__builtin_expect(dv.ok && udiv_all_safe(q), 1)) {
    q[0] = udiv(q[0], dv.x4);
    q[1] = udiv(q[1], dv.y4);
    q[2] = udiv(q[2], dv.z4);
  } else {
    q[0] = q[0] / dv.x4.d;
    q[1] = q[1] / dv.y4.d;
    q[2] = q[2] / dv.z4.d;
  }
}

// the six uniform divisors of the solver kernels
struct SolveDivs {
  UDiv x2, y2, z2, x4, y4, z4;
  bool ok;
};
__device__ __forceinline__ SolveDivs make_solve_divs(float hx, float hy, float hz)
{
  SolveDivs s;
  s.x2 = make_udiv(2.f * hx); s.y2 = make_udiv(2.f * hy); s.z2 = make_udiv(2.f * hz);
  s.x4 = make_udiv(4.f * hx); s.y4 = make_udiv(4.f * hy); s.z4 = make_udiv(4.f * hz);
  s.ok = udiv_divisor_ok(s.x2.d) && udiv_divisor_ok(s.y2.d) && udiv_divisor_ok(s.z2.d) && udiv_divisor_ok(s.x4.d) &&
         udiv_divisor_ok(s.y4.d) && udiv_divisor_ok(s.z4.d);
  return s;
}

// A.3: src/kernels/solve_3d.cu:177-260
__device__ __forceinline__ void phi_ksi_voxel(const Hood<8>& n, const SolveDivs& dv, float eps_s, float eps_d, float& phi,
                                              float& ksi)
{
  // numerators in the order x, y, z of u, v, w, then of f (the reference's left-to-right sums)
  float q[12] = {n.xp[U] - n.xm[U] + n.xp[DU] - n.xm[DU],     n.yp[U] - n.ym[U] + n.yp[DU] - n.ym[DU],
                 n.zp[U] - n.zm[U] + n.zp[DU] - n.zm[DU],     n.xp[V] - n.xm[V] + n.xp[DV] - n.xm[DV],
                 n.yp[V] - n.ym[V] + n.yp[DV] - n.ym[DV],     n.zp[V] - n.zm[V] + n.zp[DV] - n.zm[DV],
                 n.xp[Wf] - n.xm[Wf] + n.xp[DW] - n.xm[DW],   n.yp[Wf] - n.ym[Wf] + n.yp[DW] - n.ym[DW],
                 n.zp[Wf] - n.zm[Wf] + n.zp[DW] - n.zm[DW],   n.xp[F0] - n.xm[F0] + n.xp[F1] - n.xm[F1],
                 n.yp[F0] - n.ym[F0] + n.yp[F1] - n.ym[F1],   n.zp[F0] - n.zm[F0] + n.zp[F1] - n.zm[F1]};
  if (dv.ok && udiv_all_safe(q)) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      q[3 * c + 0] = udiv(q[3 * c + 0], dv.x2);
      q[3 * c + 1] = udiv(q[3 * c + 1], dv.y2);
      q[3 * c + 2] = udiv(q[3 * c + 2], dv.z2);
    }
    q[9] = udiv(q[9], dv.x4);
    q[10] = udiv(q[10], dv.y4);
    q[11] = udiv(q[11], dv.z4);
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      q[3 * c + 0] = q[3 * c + 0] / dv.x2.d;
      q[3 * c + 1] = q[3 * c + 1] / dv.y2.d;
      q[3 * c + 2] = q[3 * c + 2] / dv.z2.d;
    }
    q[9] = q[9] / dv.x4.d;
    q[10] = q[10] / dv.y4.d;
    q[11] = q[11] / dv.z4.d;
  }
  const float dux = q[0], duy = q[1], duz = q[2], dvx = q[3], dvy = q[4], dvz = q[5], dwx = q[6], dwy = q[7], dwz = q[8];

  phi = 1.f / (2.f * sqrtf(dux * dux + duy * duy + duz * duz + dvx * dvx + dvy * dvy + dvz * dvz + dwx * dwx +
                           dwy * dwy + dwz * dwz + eps_s * eps_s));

  const float fx = q[9], fy = q[10], fz = q[11];
  const float ft = n.c[F1] - n.c[F0];

  const float J11 = fx * fx, J22 = fy * fy, J33 = fz * fz;
  const float J12 = fx * fy, J13 = fx * fz, J23 = fy * fz;
  const float J14 = fx * ft, J24 = fy * ft, J34 = fz * ft, J44 = ft * ft;

  const float du = n.c[DU], dv_c = n.c[DV], dw = n.c[DW];
  float s = (J11 * du + J12 * dv_c + J13 * dw + J14) * du + (J12 * du + J22 * dv_c + J23 * dw + J24) * dv_c +
            (J13 * du + J23 * dv_c + J33 * dw + J34) * dw + (J14 * du + J24 * dv_c + J34 * dw + J44);
  s = static_cast<float>(s > 0) * s;
  ksi = 1.f / (2.f * sqrtf(s + eps_d * eps_d));
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


__device__ __forceinline__ float lane_left_or(float v, float edge)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v),
                                                               0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_right_or(float v, float edge)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v),
                                                               0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// ---- the sweep's per-voxel arithmetic ------------------------------------------------------------------------------
// The neighbour sums use S = U + dU formed once per voxel: the reference's (U[nb] + dU[nb]) - U[c] reads the same two
// operands, so S[nb] - U[c] is bit-identical and saves 15 adds, 6 lane shifts and 3 of the 9 LDS images per voxel.
constexpr int kTY3 = 8;  // rows (waves) per workgroup of k_sweep6 and k_phiksi6
enum { LF0 = 0, LF1 = 1, LPHI = 2, LSU = 3, LSV = 4, LSW = 5, kNL = 6 };

struct Face6 {  // the six stencilled quantities of one neighbour
  float v[kNL];
};

__device__ __forceinline__ void sweep_voxel_s(const Face6& xm, const Face6& xp, const Face6& ym, const Face6& yp,
                                              const Face6& zm, const Face6& zp, const float (&c)[kNL], float Uc, float Vc,
                                              float Wc, float dVc, float dWc, float ksi, float hx, float hy, float hz,
                                              float alpha, bool has_xp, bool has_xm, bool has_yp, bool has_ym, bool has_zp,
                                              bool has_zm, float& r_du, float& r_dv, float& r_dw)
{
  // plain IEEE divisions here: k_sweep6 is bound by memory, not by the vector unit (the uniform-divisor form pays in
  // k_phiksi6 and k_sweep7)
  const float fx = (xp.v[LF0] - xm.v[LF0] + xp.v[LF1] - xm.v[LF1]) / (4.f * hx);
  const float fy = (yp.v[LF0] - ym.v[LF0] + yp.v[LF1] - ym.v[LF1]) / (4.f * hy);
  const float fz = (zp.v[LF0] - zm.v[LF0] + zp.v[LF1] - zm.v[LF1]) / (4.f * hz);
  const float ft = c[LF1] - c[LF0];

  const float J11 = fx * fx, J22 = fy * fy, J33 = fz * fz;
  const float J12 = fx * fy, J13 = fx * fz, J23 = fy * fz;
  const float J14 = fx * ft, J24 = fy * ft, J34 = fz * ft;

  const float hx_2 = alpha / (hx * hx);
  const float hy_2 = alpha / (hy * hy);
  const float hz_2 = alpha / (hz * hz);
  const float wxp = static_cast<float>(has_xp) * hx_2;
  const float wxm = static_cast<float>(has_xm) * hx_2;
  const float wyp = static_cast<float>(has_yp) * hy_2;
  const float wym = static_cast<float>(has_ym) * hy_2;
  const float wzp = static_cast<float>(has_zp) * hz_2;
  const float wzm = static_cast<float>(has_zm) * hz_2;

  const float phi_xp = (xp.v[LPHI] + c[LPHI]) / 2.f;
  const float phi_xm = (xm.v[LPHI] + c[LPHI]) / 2.f;
  const float phi_yp = (yp.v[LPHI] + c[LPHI]) / 2.f;
  const float phi_ym = (ym.v[LPHI] + c[LPHI]) / 2.f;
  const float phi_zp = (zp.v[LPHI] + c[LPHI]) / 2.f;
  const float phi_zm = (zm.v[LPHI] + c[LPHI]) / 2.f;

  const float sumH = (wxp * phi_xp + wxm * phi_xm + wyp * phi_yp + wym * phi_ym + wzp * phi_zp + wzm * phi_zm);
  const float sumU = phi_xp * wxp * (xp.v[LSU] - Uc) + phi_xm * wxm * (xm.v[LSU] - Uc) + phi_yp * wyp * (yp.v[LSU] - Uc) +
                     phi_ym * wym * (ym.v[LSU] - Uc) + phi_zp * wzp * (zp.v[LSU] - Uc) + phi_zm * wzm * (zm.v[LSU] - Uc);
  const float sumV = phi_xp * wxp * (xp.v[LSV] - Vc) + phi_xm * wxm * (xm.v[LSV] - Vc) + phi_yp * wyp * (yp.v[LSV] - Vc) +
                     phi_ym * wym * (ym.v[LSV] - Vc) + phi_zp * wzp * (zp.v[LSV] - Vc) + phi_zm * wzm * (zm.v[LSV] - Vc);
  const float sumW = phi_xp * wxp * (xp.v[LSW] - Wc) + phi_xm * wxm * (xm.v[LSW] - Wc) + phi_yp * wyp * (yp.v[LSW] - Wc) +
                     phi_ym * wym * (ym.v[LSW] - Wc) + phi_zp * wzp * (zp.v[LSW] - Wc) + phi_zm * wzm * (zm.v[LSW] - Wc);

  r_du = (ksi * (-J14 - J12 * dVc - J13 * dWc) + sumU) / (ksi * J11 + sumH);
  r_dv = (ksi * (-J24 - J12 * r_du - J23 * dWc) + sumV) / (ksi * J22 + sumH);
  r_dw = (ksi * (-J34 - J13 * r_du - J23 * r_dv) + sumW) / (ksi * J33 + sumH);
}

// (du becomes Su = u + du; Sv, Sw are kept beside dv, dw because the in-voxel Gauss-Seidel step still needs those).
struct PlaneRegs {
  float f0, f1, phi, u, v, w, su, dv, dw, sv, sw, ksi;
  float fz, ft;  // k_pair8 on precomputed frame derivatives: f0, f1 then hold fx, fy (dead code elsewhere)
};

__device__ __forceinline__ void plane_finish(PlaneRegs& p)
{
  p.su = p.u + p.su;  // su held the raw du
  p.sv = p.v + p.dv;
  p.sw = p.w + p.dw;
}

__device__ __forceinline__ Face6 plane_face(const PlaneRegs& p)
{
  Face6 f;
  f.v[LF0] = p.f0; f.v[LF1] = p.f1; f.v[LPHI] = p.phi; f.v[LSU] = p.su; f.v[LSV] = p.sv; f.v[LSW] = p.sw;
  return f;
}

constexpr int kRing = 4;  // LDS ring of the DMA-fed halos in k_sweep6 / k_phiksi6: planes z .. z+3
typedef __attribute__((address_space(3))) float LdsFloat;

// ---- one sweep per launch: k_sweep6 ------------------------------------------------------------------------------------
// Every load is a `global_load_dword vdst, voff, s[base:base+1]` (two SGPRs per array, no descriptor, one VALU add per
// plane for the shared row offset); the 18 x-halo values of a row are fetched by ONE instruction (lane i / 32+i reads
// array i, landing in LDS by DMA: `global_load_lds_dword` writes lane L's dword to M0 + 4 L, masked lanes write nothing,
// tools/lab/dma_probe.hip); the edge waves' halo rows land in LDS the same way.  Own rows are requested TWO planes ahead
// (five rotating register sets).  Issued through inline assembly on purpose: hipcc makes every later ds_read wait for
// vmcnt(0) once it knows of an LDS-DMA in flight, and it cannot count loads it does not see, so the two
// `s_waitcnt vmcnt(N)` per step below are the only ones, tied to the registers they guard by "+v" operands.
__device__ __forceinline__ float gld(const float* base, unsigned byte_off)
{
  float v;
  // s_nop 4: if the register allocator ever restores the base pair from spilled lanes (v_readlane) right before this
  // statement, gfx9 wants 5 wait states between a VALU write of an SGPR and a VMEM read of it, and the hazard
  // recogniser does not look inside inline assembly
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(base) : "memory");
  return v;
}
__device__ __forceinline__ void gst(float* base, unsigned byte_off, float v)
{
  // `nt`: the results of a launch are read again by the NEXT launch at the earliest, a whole volume later, so they need not
  // displace the planes the neighbouring tiles are about to re-read in L2 / the memory-side cache.  Measured, paired, on the
  // 512^3 solve: -0.7 ... -2.4 % (profiles/r04_store_hint_ab.txt, r04_store_hints.txt; sc1 / sc0 sc1 / nt sc1 do not beat it,
  // and `nt` on the centre-only DMA LOADS costs 5 %).
  asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2 nt" ::"v"(byte_off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void gld_lds(const float* base, unsigned byte_off, float* lds_dst)
{
  const unsigned m0v = static_cast<unsigned>(reinterpret_cast<unsigned long>((LdsFloat*)lds_dst));
  asm volatile("s_nop 4\n\tglobal_load_lds_dword %0, %1" ::"v"(byte_off), "s"(base), "{m0}"(m0v) : "memory");
}
__device__ __forceinline__ void gld_lds_lane(const float* lane_addr, float* lds_dst)
{
  const unsigned m0v = static_cast<unsigned>(reinterpret_cast<unsigned long>((LdsFloat*)lds_dst));
  // s_nop 0: the compiler's write of M0 may sit right in front of this statement, and an LDS-DMA instruction must not
  // read M0 in the cycle after a scalar write of it (one wait state; nobody inserts it inside inline assembly)
  asm volatile("s_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(lane_addr), "{m0}"(m0v) : "memory");
}
// The registers a hand-issued load writes are, to the compiler, defined at the load statement: nothing stops it from
// copying them (a v_mov for a tied asm operand, a control-flow merge, a loop back edge) while the data is still on its way.
// So the wait is a statement of its own with NO register operands, and only the empty statement after it hands the
// registers back through "+v": whatever copies the compiler makes for those operands then come after the s_waitcnt.  The
// hand-back leaves a comment naming its registers in the generated code ("; f3d_handback v12 v13 ..."), which is what
// tests/test_isa_hazards.py uses to walk the control-flow graph of every kernel here for any read of a load's destination
// between the load and the wait that covers it, so a copy inserted elsewhere fails the build check instead of a parity test
// once in a few hundred launches.
#define F3D_WAIT_PLANE(N, P)                                                                                           \
  do {                                                                                                                 \
    asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");                                                              \
    asm volatile("; f3d_handback %0 %1 %2 %3 %4 %5 %6 %7 %8 %9"                                                        \
                 : "+v"((P).f0), "+v"((P).f1), "+v"((P).phi), "+v"((P).u), "+v"((P).v), "+v"((P).w), "+v"((P).su),     \
                   "+v"((P).dv), "+v"((P).dw), "+v"((P).ksi)::"memory");                                              \
  } while (0)

// ABLATE (timing experiments only, results are wrong): 1 = no arithmetic, 2 = no halo traffic, 3 = no LDS exchange
template <int ABLATE, int TY>
__global__ __launch_bounds__(kLanes* TY, 4) void k_sweep6(SolveArgs a, F3dGeo g, int zchunk, int ntx, int nty, int n_tiles,
                                                            int xcd_remap)
{
  __shared__ float img[2][kNL][TY + 2][kLanes];  // face image of the current plane, double buffered
  __shared__ float hrow[kRing][2][9][kLanes];      // raw y-halo rows (edge waves), by LDS-DMA
  __shared__ float hcol[kRing][TY][kLanes];      // raw x-halo values of a row: [array] left, [32 + array] right

  int tile = static_cast<int>(blockIdx.x);
  if (xcd_remap) {
    const int per_xcd = (n_tiles + 7) / 8;
    tile = (tile % 8) * per_xcd + tile / 8;
  }
  if (tile >= n_tiles) return;
  const int tx = tile % ntx;
  const int ty = (tile / ntx) % nty;
  const int tz = tile / (ntx * nty);

  const int lane = threadIdx.x;
  const int r = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  const int z0 = g.z_lo + tz * zchunk;
  const int z1 = min(z0 + zchunk, g.z_hi);
  const int y0 = ty * TY;
  const int y = y0 + r;
  const int yy = f3d_clampi(f3d_mir(y, g.H), 0, g.H - 1);
  const int x = tx * kLanes + lane;
  const int xi = f3d_clampi(f3d_mir(x, g.W), 0, g.W - 1);
  const unsigned xb = static_cast<unsigned>(xi) * 4u;
  const bool owner = x < g.W && y < g.H;
  const int side = lane < 32 ? 0 : 1;
  const int xh = f3d_clampi(f3d_mir(side == 0 ? tx * kLanes - 1 : tx * kLanes + kLanes, g.W), 0, g.W - 1);
  const bool edge = (r == 0) || (r == TY - 1);
  const int which = r == 0 ? 0 : 1;
  const int yh_row = f3d_clampi(f3d_mir(r == 0 ? y0 - 1 : y0 + TY, g.H), 0, g.H - 1);
  const int lds_halo = r == 0 ? 0 : TY + 1;

  // array bases moved to the first plane this chunk touches: every byte offset below is small and positive
  const int zb = z0 > 0 ? z0 - 1 : 0;
  const size_t base_off = f3d_row(g, 0, zb);
  const unsigned plane_b = static_cast<unsigned>(g.Hc) * static_cast<unsigned>(g.pitch) * 4u;
  const unsigned row_b = static_cast<unsigned>(g.pitch) * 4u;
  constexpr int kOrder[9] = {F0, F1, U, V, Wf, DU, DV, DW, PHI};  // order inside the halo rings
  const float* base[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) base[i] = a.in[i] + base_off;
  float* obase[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) obase[i] = a.out[i] + base_off;
  // lane i (and 32 + i), i < 9, gathers the x-halo of array kOrder[i]: its own 64-bit base
  const bool col_lane = (lane & 31) < 9;
  const float* lane_base = base[0];
#pragma unroll
  for (int i = 1; i < 9; ++i)
    if ((lane & 31) == i) lane_base = base[kOrder[i]];
  lane_base += xh;

  auto rowoff = [&](int yrow, int zz) {
    return static_cast<unsigned>(__builtin_amdgcn_readfirstlane(
        static_cast<int>(static_cast<unsigned>(zz - zb) * plane_b + static_cast<unsigned>(yrow) * row_b)));
  };
  auto load_plane = [&](PlaneRegs& p, int zz) {
    const unsigned off = xb + rowoff(yy, zz);
    p.f0 = gld(base[F0], off);
    p.f1 = gld(base[F1], off);
    p.u = gld(base[U], off);
    p.v = gld(base[V], off);
    p.w = gld(base[Wf], off);
    p.su = gld(base[DU], off);
    p.dv = gld(base[DV], off);
    p.dw = gld(base[DW], off);
    p.phi = gld(base[PHI], off);
    p.ksi = gld(base[9], off);
  };
  auto dma_halos = [&](int zz) {  // 1 instruction per wave + 9 for the two edge waves
    const int slot = zz & (kRing - 1);
    if (col_lane) gld_lds_lane(lane_base + (rowoff(yy, zz) >> 2), &hcol[slot][r][0]);
    if (edge) {
      const unsigned off = xb + rowoff(yh_row, zz);
#pragma unroll
      for (int i = 0; i < 9; ++i) gld_lds(base[kOrder[i]], off, &hrow[slot][which][i][0]);
    }
  };
  auto ring_row = [&](PlaneRegs& p, int slot) {
    const float* d = &hrow[slot][which][0][lane];
    p.f0 = d[0 * kLanes]; p.f1 = d[1 * kLanes]; p.u = d[2 * kLanes]; p.v = d[3 * kLanes]; p.w = d[4 * kLanes];
    p.su = d[5 * kLanes]; p.dv = d[6 * kLanes]; p.dw = d[7 * kLanes]; p.phi = d[8 * kLanes];
  };
  auto ring_col = [&](PlaneRegs& p, int slot) {
    const float* d = &hcol[slot][r][side * 32];
    p.f0 = d[0]; p.f1 = d[1]; p.u = d[2]; p.v = d[3]; p.w = d[4]; p.su = d[5]; p.dv = d[6]; p.dw = d[7]; p.phi = d[8];
  };

  // write the face image of a finished plane (own row, and the halo row an edge wave keeps in its ring) into buffer nb
  auto publish = [&](const PlaneRegs& pl, int nb, int ring_slot) {
    const Face6 f = plane_face(pl);
#pragma unroll
    for (int i = 0; i < kNL; ++i) img[nb][i][r + 1][lane] = f.v[i];
    if (edge) {
      PlaneRegs Hc;
      ring_row(Hc, ring_slot);
      plane_finish(Hc);
      const Face6 hf = plane_face(Hc);
#pragma unroll
      for (int i = 0; i < kNL; ++i) img[nb][i][lds_halo][lane] = hf.v[i];
    }
  };

  // M, C, P: finished planes z-1, z, z+1.  Q1: raw plane z+2, requested one step ago.  Q2: receives plane z+3.
  auto step = [&](auto full, const PlaneRegs& M, const PlaneRegs& C, const PlaneRegs& P, PlaneRegs& Q1, PlaneRegs& Q2, int z) {
    constexpr bool FULL = decltype(full)::value;
    const bool row3 = FULL || z + 3 <= z1;   // plane z+3 is somebody's z-neighbour
    const bool halo3 = FULL || z + 3 < z1;   // plane z+3 is computed by this chunk
    if (row3) load_plane(Q2, f3d_mir(z + 3, g.D));
    if (halo3 && ABLATE != 2) dma_halos(z + 3);

    const int b = z & 1;
    const int slot = z & (kRing - 1);
    const Face6 cf = plane_face(C);
    if (ABLATE != 3) __syncthreads();  // the image of plane z is complete: it was written during step z-1 (or the prologue)

    Face6 ym, yp, xm, xp;
#pragma unroll
    for (int i = 0; i < kNL; ++i) {
      ym.v[i] = img[b][i][r][lane];
      yp.v[i] = img[b][i][r + 2][lane];
    }
    PlaneRegs X;
    ring_col(X, slot);
    // Publish the NEXT plane now, off the critical path of the next barrier: buffer b^1 was last read during step z-1,
    // i.e. before the barrier every wave has just passed.
    if ((FULL || z + 1 < z1) && ABLATE != 3) publish(P, b ^ 1, (z + 1) & (kRing - 1));
    plane_finish(X);
    const Face6 xf = plane_face(X);
#pragma unroll
    for (int i = 0; i < kNL; ++i) {
      xm.v[i] = lane_left_or(cf.v[i], xf.v[i]);
      xp.v[i] = lane_right_or(cf.v[i], xf.v[i]);
    }
    float r_du, r_dv, r_dw;
    if (ABLATE == 1) {
      r_du = xm.v[0] + xp.v[1] + ym.v[2] + yp.v[3] + M.su + P.sv + C.ksi;
      r_dv = xm.v[4] + xp.v[5] + ym.v[0] + yp.v[1] + M.f0 + P.f1 + C.u;
      r_dw = xm.v[2] + xp.v[3] + ym.v[4] + yp.v[5] + M.phi + P.phi + C.dv + C.dw + C.v + C.w;
    } else {
      sweep_voxel_s(xm, xp, ym, yp, plane_face(M), plane_face(P), cf.v, C.u, C.v, C.w, C.dv, C.dw, C.ksi, a.hx, a.hy, a.hz,
                    a.p0, x < g.W - 1, x > 0, y < g.H - 1, y > 0, z < g.D - 1, z > 0, r_du, r_dv, r_dw);
    }
    asm volatile("" ::"v"(r_du), "v"(r_dv), "v"(r_dw));
    __builtin_amdgcn_sched_barrier(0);
    // All that was requested BEFORE this step must have landed (plane z+2, its halos, the last stores); what this step
    // requested stays in flight: the counter retires in order, so allow exactly this step's loads.
    if (FULL && ABLATE == 2) {
      F3D_WAIT_PLANE(10, Q1);
    } else if (FULL) {
      if (edge) F3D_WAIT_PLANE(20, Q1);  // 10 row + 1 column gather + 9 halo-row loads
      else F3D_WAIT_PLANE(11, Q1);
    } else {
      F3D_WAIT_PLANE(0, Q1);
    }
    if (FULL || z + 2 <= z1) plane_finish(Q1);
    __builtin_amdgcn_sched_barrier(0);
    if (owner) {
      const unsigned off = xb + rowoff(yy, z);
      gst(obase[0], off, r_du);
      gst(obase[1], off, r_dv);
      gst(obase[2], off, r_dw);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  PlaneRegs A, B, C, D, E;
  E = PlaneRegs{};
  D = PlaneRegs{};
  load_plane(A, f3d_mir(z0 - 1, g.D));
  load_plane(B, z0);
  load_plane(C, f3d_mir(z0 + 1, g.D));
  if (z0 + 2 <= z1) load_plane(D, f3d_mir(z0 + 2, g.D));
  dma_halos(z0);
  if (z0 + 1 < z1) dma_halos(z0 + 1);
  if (z0 + 2 < z1) dma_halos(z0 + 2);
  F3D_WAIT_PLANE(0, A);
  F3D_WAIT_PLANE(0, B);
  F3D_WAIT_PLANE(0, C);
  F3D_WAIT_PLANE(0, D);
  plane_finish(A);
  plane_finish(B);
  plane_finish(C);
  __syncthreads();  // DMA-written rings are visible
  publish(B, z0 & 1, z0 & (kRing - 1));
  __builtin_amdgcn_sched_barrier(0);
  int z = z0;
  for (; z + 7 < z1; z += 5) {
    step(std::true_type{}, A, B, C, D, E, z);
    step(std::true_type{}, B, C, D, E, A, z + 1);
    step(std::true_type{}, C, D, E, A, B, z + 2);
    step(std::true_type{}, D, E, A, B, C, z + 3);
    step(std::true_type{}, E, A, B, C, D, z + 4);
  }
  for (; z < z1; z += 5) {
    step(std::false_type{}, A, B, C, D, E, z);
    if (z + 1 < z1) step(std::false_type{}, B, C, D, E, A, z + 1);
    if (z + 2 < z1) step(std::false_type{}, C, D, E, A, B, z + 2);
    if (z + 3 < z1) step(std::false_type{}, D, E, A, B, C, z + 3);
    if (z + 4 < z1) step(std::false_type{}, E, A, B, C, D, z + 4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores issued by hand
}

struct Plane8 {
  float v[8];
};

// ---- phi/ksi: k_phiksi6 -----------------------------------------------------------------------------------------------
// The load path of k_sweep6 (hand-issued global loads, one-instruction x-halo gather, LDS-DMA halo rows, rows requested
// two steps ahead, next plane published right after the barrier) for the eight inputs of A.3, all of them stencilled
// (the central differences of A.3 do not factor, there is nothing to pre-combine).
#define F3D_WAIT_PLANE8(N, P)                                                                                          \
  do {                                                                                                                 \
    asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");                                                              \
    asm volatile("; f3d_handback %0 %1 %2 %3 %4 %5 %6 %7"                                                              \
                 : "+v"((P).v[0]), "+v"((P).v[1]), "+v"((P).v[2]), "+v"((P).v[3]), "+v"((P).v[4]), "+v"((P).v[5]),     \
                   "+v"((P).v[6]), "+v"((P).v[7])::"memory");                                                         \
  } while (0)

__global__ __launch_bounds__(kLanes* kTY3, 4) void k_phiksi6(SolveArgs a, F3dGeo g, int zchunk, int ntx, int nty, int n_tiles,
                                                             int xcd_remap)
{
  constexpr int NA = 8;
  __shared__ float img[2][NA][kTY3 + 2][kLanes];
  __shared__ float hrow[kRing][2][NA][kLanes];
  __shared__ float hcol[kRing][kTY3][kLanes];
  SolveDivs divs = make_solve_divs(a.hx, a.hy, a.hz);
  divs.ok = divs.ok && !a.plain_division;

  int tile = static_cast<int>(blockIdx.x);
  if (xcd_remap) {
    const int per_xcd = (n_tiles + 7) / 8;
    tile = (tile % 8) * per_xcd + tile / 8;
  }
  if (tile >= n_tiles) return;
  const int tx = tile % ntx;
  const int ty = (tile / ntx) % nty;
  const int tz = tile / (ntx * nty);

  const int lane = threadIdx.x;
  const int r = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  const bool second = a.z2_hi > a.z2_lo && tz >= a.nz_first;  // wave-uniform
  const int z0 = second ? a.z2_lo + (tz - a.nz_first) * zchunk : g.z_lo + tz * zchunk;
  const int z1 = min(z0 + zchunk, second ? a.z2_hi : g.z_hi);
  const int y0 = ty * kTY3;
  const int y = y0 + r;
  const int yy = f3d_clampi(f3d_mir(y, g.H), 0, g.H - 1);
  const int x = tx * kLanes + lane;
  const int xi = f3d_clampi(f3d_mir(x, g.W), 0, g.W - 1);
  const unsigned xb = static_cast<unsigned>(xi) * 4u;
  const bool owner = x < g.W && y < g.H;
  const int side = lane < 32 ? 0 : 1;
  const int xh = f3d_clampi(f3d_mir(side == 0 ? tx * kLanes - 1 : tx * kLanes + kLanes, g.W), 0, g.W - 1);
  const bool edge = (r == 0) || (r == kTY3 - 1);
  const int which = r == 0 ? 0 : 1;
  const int yh_row = f3d_clampi(f3d_mir(r == 0 ? y0 - 1 : y0 + kTY3, g.H), 0, g.H - 1);
  const int lds_halo = r == 0 ? 0 : kTY3 + 1;

  const int zb = z0 > 0 ? z0 - 1 : 0;
  const size_t base_off = f3d_row(g, 0, zb);
  const unsigned plane_b = static_cast<unsigned>(g.Hc) * static_cast<unsigned>(g.pitch) * 4u;
  const unsigned row_b = static_cast<unsigned>(g.pitch) * 4u;
  const float* base[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) base[i] = a.in[i] + base_off;
  float* obase[2] = {a.out[0] + base_off, a.out[1] + base_off};
  const bool col_lane = (lane & 31) < NA;
  const float* lane_base = base[0];
#pragma unroll
  for (int i = 1; i < NA; ++i)
    if ((lane & 31) == i) lane_base = base[i];
  lane_base += xh;

  auto rowoff = [&](int yrow, int zz) {
    return static_cast<unsigned>(__builtin_amdgcn_readfirstlane(
        static_cast<int>(static_cast<unsigned>(zz - zb) * plane_b + static_cast<unsigned>(yrow) * row_b)));
  };
  auto load_plane = [&](Plane8& p, int zz) {
    const unsigned off = xb + rowoff(yy, zz);
#pragma unroll
    for (int i = 0; i < NA; ++i) p.v[i] = gld(base[i], off);
  };
  auto dma_halos = [&](int zz) {
    const int slot = zz & (kRing - 1);
    if (col_lane) gld_lds_lane(lane_base + (rowoff(yy, zz) >> 2), &hcol[slot][r][0]);
    if (edge) {
      const unsigned off = xb + rowoff(yh_row, zz);
#pragma unroll
      for (int i = 0; i < NA; ++i) gld_lds(base[i], off, &hrow[slot][which][i][0]);
    }
  };
  auto publish = [&](const Plane8& pl, int nb, int ring_slot) {
#pragma unroll
    for (int i = 0; i < NA; ++i) img[nb][i][r + 1][lane] = pl.v[i];
    if (edge) {
#pragma unroll
      for (int i = 0; i < NA; ++i) img[nb][i][lds_halo][lane] = hrow[ring_slot][which][i][lane];
    }
  };

  auto step = [&](auto full, const Plane8& M, const Plane8& C, const Plane8& P, Plane8& Q1, Plane8& Q2, int z) {
    constexpr bool FULL = decltype(full)::value;
    const bool row3 = FULL || z + 3 <= z1;
    const bool halo3 = FULL || z + 3 < z1;
    if (row3) load_plane(Q2, f3d_mir(z + 3, g.D));
    if (halo3) dma_halos(z + 3);

    const int b = z & 1;
    const int slot = z & (kRing - 1);
    __syncthreads();

    Hood<8> n;
    float xcol[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      n.c[i] = C.v[i];
      n.ym[i] = img[b][i][r][lane];
      n.yp[i] = img[b][i][r + 2][lane];
      n.zm[i] = M.v[i];
      n.zp[i] = P.v[i];
      xcol[i] = hcol[slot][r][side * 32 + i];
    }
    if (FULL || z + 1 < z1) publish(P, b ^ 1, (z + 1) & (kRing - 1));
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      n.xm[i] = lane_left_or(C.v[i], xcol[i]);
      n.xp[i] = lane_right_or(C.v[i], xcol[i]);
    }
    float phi, ksi;
    phi_ksi_voxel(n, divs, a.p0, a.p1, phi, ksi);
    asm volatile("" ::"v"(phi), "v"(ksi));
    __builtin_amdgcn_sched_barrier(0);
    if (FULL) {
      if (edge) F3D_WAIT_PLANE8(17, Q1);  // 8 row loads + 1 column gather + 8 halo-row loads of this step stay in flight
      else F3D_WAIT_PLANE8(9, Q1);
    } else {
      F3D_WAIT_PLANE8(0, Q1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (owner) {
      const unsigned off = xb + rowoff(yy, z);
      gst(obase[0], off, phi);
      gst(obase[1], off, ksi);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  Plane8 A, B, C, D, E;
  D = Plane8{};
  E = Plane8{};
  load_plane(A, f3d_mir(z0 - 1, g.D));
  load_plane(B, z0);
  load_plane(C, f3d_mir(z0 + 1, g.D));
  if (z0 + 2 <= z1) load_plane(D, f3d_mir(z0 + 2, g.D));
  dma_halos(z0);
  if (z0 + 1 < z1) dma_halos(z0 + 1);
  if (z0 + 2 < z1) dma_halos(z0 + 2);
  F3D_WAIT_PLANE8(0, A);
  F3D_WAIT_PLANE8(0, B);
  F3D_WAIT_PLANE8(0, C);
  F3D_WAIT_PLANE8(0, D);
  __syncthreads();
  publish(B, z0 & 1, z0 & (kRing - 1));
  __builtin_amdgcn_sched_barrier(0);
  int z = z0;
  for (; z + 7 < z1; z += 5) {
    step(std::true_type{}, A, B, C, D, E, z);
    step(std::true_type{}, B, C, D, E, A, z + 1);
    step(std::true_type{}, C, D, E, A, B, z + 2);
    step(std::true_type{}, D, E, A, B, C, z + 3);
    step(std::true_type{}, E, A, B, C, D, z + 4);
  }
  for (; z < z1; z += 5) {
    step(std::false_type{}, A, B, C, D, E, z);
    if (z + 1 < z1) step(std::false_type{}, B, C, D, E, A, z + 1);
    if (z + 2 < z1) step(std::false_type{}, C, D, E, A, B, z + 2);
    if (z + 3 < z1) step(std::false_type{}, D, E, A, B, C, z + 3);
    if (z + 4 < z1) step(std::false_type{}, E, A, B, C, D, z + 4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- two sweeps per launch: k_sweep7 (temporal blocking) ----------------------------------------------------------------
// A sweep streams 52 B per voxel for ~150 flops, and the 13-stream ceiling of the memory system (tools/lab/stream_lab)
// is ~5.0-5.6 TB/s, so a one-sweep kernel cannot pass ~65 % of the 8 TB/s peak.  Two consecutive sweeps read the same
// f0, f1, u, v, w, phi, ksi; only du, dv, dw change in between.  This kernel keeps the intermediate field on chip:
//
//   * the workgroup owns TY core rows of an aligned 64-column tile and marches along z as k_sweep6 does, but with
//     TY + 2 row waves (rows y0-1 .. y0+TY): every row wave computes sweep 1 ("stage 1") of its row for plane q; the TY
//     core waves then compute sweep 2 ("stage 2") for plane q-1 from the stage-1 results of planes q-2, q-1 (kept in
//     registers), q (just computed), of the rows above and below (LDS image img1) and of the lanes left and right (DPP);
//   * one more wave, the column wave, computes stage 1 for the 2 x TY voxels just left and right of the tile (columns
//     x0-1 and x0+64), which the tile's edge lanes need in stage 2.  It works from LDS only: the row waves gather TWO
//     columns on each side (and ksi) with their one x-halo instruction, and those raw values are all it needs;
//   * stage 2 reuses everything of stage 1 that does not depend on du, dv, dw (the J terms, the six face weights, the three
//     denominators), carried in registers from one z step to the next;
//   * at the faces of the volume the reference's mirror rule makes the missing neighbour equal the opposite one
//     (index -1 -> 1, n -> n-2), so stage 2 substitutes xm := xp etc. there and never looks at stage-1 values of
//     voxels outside the volume (which the halo waves compute from mirrored data, i.e. with the operands in another
//     order).
// HBM traffic for two sweeps: 10 reads + 3 writes per voxel instead of 20 + 6.
struct Carry {
  float J12, J13, J23, J14, J24, J34, d1, d2, d3, ksi, pw[6], U, V, W;
  float fx, fy, fz, ft;  // handed on to a phi/ksi second stage (k_pair8, PAIR_SP); dead code elsewhere
};
struct S3 {
  float u, v, w;
};

// sweep_voxel_s, additionally handing out what stage 2 of the same voxel reuses (same operations, same order)
// FD: the frame derivatives of the voxel are handed in (gfx, gfy, gfz, gft: computed once per level by k_frame_derivatives with
// the same expressions) instead of being formed from the neighbours' frame values
// XSEL: the x-face weights are applied by selection (k_pair8): has_xp / has_xm are looked at only where `at_x_face` (wave-
// uniform) says the tile touches a face of the volume, so an interior tile multiplies by alpha / hx^2 from a scalar register
// instead of carrying two per-lane weights through the march.  Same bits: the weight of a missing neighbour is 0, and a phi
// average (positive, finite) times +0 is the +0 the selection writes.
template <bool FD = false, bool XSEL = false>
__device__ __forceinline__ void sweep_stage1(const Face6& xm, const Face6& xp, const Face6& ym, const Face6& yp,
                                             const Face6& zm, const Face6& zp, const float (&c)[kNL], float Uc, float Vc,
                                             float Wc, float dVc, float dWc, float ksi, float hx, float hy, float hz,
                                             const FDivs& fd, float alpha, bool has_xp, bool has_xm, bool has_yp, bool has_ym,
                                             bool has_zp, bool has_zm, float& r_du, float& r_dv, float& r_dw, Carry& k,
                                             float gfx = 0.f, float gfy = 0.f, float gfz = 0.f, float gft = 0.f,
                                             bool at_x_face = true, float w_x = 0.f, float w_y = 0.f, float w_z = 0.f)
{
  float fq[3] = {gfx, gfy, gfz};
  if (!FD) {
    fq[0] = xp.v[LF0] - xm.v[LF0] + xp.v[LF1] - xm.v[LF1];
    fq[1] = yp.v[LF0] - ym.v[LF0] + yp.v[LF1] - ym.v[LF1];
    fq[2] = zp.v[LF0] - zm.v[LF0] + zp.v[LF1] - zm.v[LF1];
    f_derivatives(fq, fd);
  }
  const float fx = fq[0], fy = fq[1], fz = fq[2];
  const float ft = FD ? gft : c[LF1] - c[LF0];
  k.fx = fx; k.fy = fy; k.fz = fz; k.ft = ft;

  const float J11 = fx * fx, J22 = fy * fy, J33 = fz * fz;
  k.J12 = fx * fy; k.J13 = fx * fz; k.J23 = fy * fz;
  k.J14 = fx * ft; k.J24 = fy * ft; k.J34 = fz * ft;

  // XSEL callers hand in alpha / h^2 made on the host (pair_consts: the same float operations)
  const float hx_2 = XSEL ? w_x : alpha / (hx * hx);
  const float hy_2 = XSEL ? w_y : alpha / (hy * hy);
  const float hz_2 = XSEL ? w_z : alpha / (hz * hz);
  const float wxp = XSEL ? hx_2 : static_cast<float>(has_xp) * hx_2;
  const float wxm = XSEL ? hx_2 : static_cast<float>(has_xm) * hx_2;
  // XSEL (k_pair8): the y / z face flags are the same for every lane of a row wave, and alpha / h^2 arrives in a scalar register:
  // (float)(flag) * w is w or +0 for the finite positive w the host checked (pair_consts), i.e. a SCALAR select instead of a
  // vector select and a vector multiply per face (the column wave, whose rows differ by lane, makes one vector select)
  const float wyp = XSEL ? (has_yp ? hy_2 : 0.f) : static_cast<float>(has_yp) * hy_2;
  const float wym = XSEL ? (has_ym ? hy_2 : 0.f) : static_cast<float>(has_ym) * hy_2;
  const float wzp = XSEL ? (has_zp ? hz_2 : 0.f) : static_cast<float>(has_zp) * hz_2;
  const float wzm = XSEL ? (has_zm ? hz_2 : 0.f) : static_cast<float>(has_zm) * hz_2;

  // phi_f * w_f: the product the reference forms first in every term of sumU/V/W and (commuted) in sumH
  k.pw[0] = (xp.v[LPHI] + c[LPHI]) / 2.f * wxp;
  k.pw[1] = (xm.v[LPHI] + c[LPHI]) / 2.f * wxm;
  k.pw[2] = (yp.v[LPHI] + c[LPHI]) / 2.f * wyp;
  k.pw[3] = (ym.v[LPHI] + c[LPHI]) / 2.f * wym;
  k.pw[4] = (zp.v[LPHI] + c[LPHI]) / 2.f * wzp;
  k.pw[5] = (zm.v[LPHI] + c[LPHI]) / 2.f * wzm;
  if (XSEL && at_x_face) {
    k.pw[0] = has_xp ? k.pw[0] : 0.f;
    k.pw[1] = has_xm ? k.pw[1] : 0.f;
  }
  const float sumH = (k.pw[0] + k.pw[1] + k.pw[2] + k.pw[3] + k.pw[4] + k.pw[5]);
  const float sumU = k.pw[0] * (xp.v[LSU] - Uc) + k.pw[1] * (xm.v[LSU] - Uc) + k.pw[2] * (yp.v[LSU] - Uc) +
                     k.pw[3] * (ym.v[LSU] - Uc) + k.pw[4] * (zp.v[LSU] - Uc) + k.pw[5] * (zm.v[LSU] - Uc);
  const float sumV = k.pw[0] * (xp.v[LSV] - Vc) + k.pw[1] * (xm.v[LSV] - Vc) + k.pw[2] * (yp.v[LSV] - Vc) +
                     k.pw[3] * (ym.v[LSV] - Vc) + k.pw[4] * (zp.v[LSV] - Vc) + k.pw[5] * (zm.v[LSV] - Vc);
  const float sumW = k.pw[0] * (xp.v[LSW] - Wc) + k.pw[1] * (xm.v[LSW] - Wc) + k.pw[2] * (yp.v[LSW] - Wc) +
                     k.pw[3] * (ym.v[LSW] - Wc) + k.pw[4] * (zp.v[LSW] - Wc) + k.pw[5] * (zm.v[LSW] - Wc);
  k.d1 = ksi * J11 + sumH;
  k.d2 = ksi * J22 + sumH;
  k.d3 = ksi * J33 + sumH;
  k.ksi = ksi;
  k.U = Uc; k.V = Vc; k.W = Wc;
  r_du = (ksi * (-k.J14 - k.J12 * dVc - k.J13 * dWc) + sumU) / k.d1;
  r_dv = (ksi * (-k.J24 - k.J12 * r_du - k.J23 * dWc) + sumV) / k.d2;
  r_dw = (ksi * (-k.J34 - k.J13 * r_du - k.J23 * r_dv) + sumW) / k.d3;
}

// the second sweep of a voxel: neighbours' S = U + dU after sweep 1, own dv, dw after sweep 1
__device__ __forceinline__ void sweep_stage2(const Carry& k, const S3& xm, const S3& xp, const S3& ym, const S3& yp,
                                             const S3& zm, const S3& zp, float dVc, float dWc, float& r_du, float& r_dv,
                                             float& r_dw)
{
  const float sumU = k.pw[0] * (xp.u - k.U) + k.pw[1] * (xm.u - k.U) + k.pw[2] * (yp.u - k.U) + k.pw[3] * (ym.u - k.U) +
                     k.pw[4] * (zp.u - k.U) + k.pw[5] * (zm.u - k.U);
  const float sumV = k.pw[0] * (xp.v - k.V) + k.pw[1] * (xm.v - k.V) + k.pw[2] * (yp.v - k.V) + k.pw[3] * (ym.v - k.V) +
                     k.pw[4] * (zp.v - k.V) + k.pw[5] * (zm.v - k.V);
  const float sumW = k.pw[0] * (xp.w - k.W) + k.pw[1] * (xm.w - k.W) + k.pw[2] * (yp.w - k.W) + k.pw[3] * (ym.w - k.W) +
                     k.pw[4] * (zp.w - k.W) + k.pw[5] * (zm.w - k.W);
  r_du = (k.ksi * (-k.J14 - k.J12 * dVc - k.J13 * dWc) + sumU) / k.d1;
  r_dv = (k.ksi * (-k.J24 - k.J12 * r_du - k.J23 * dWc) + sumV) / k.d2;
  r_dw = (k.ksi * (-k.J34 - k.J13 * r_du - k.J23 * r_dv) + sumW) / k.d3;
}

// ABL (timing experiments only, wrong results): bit 0 = no stage-1 arithmetic, bit 1 = no stage-2 arithmetic,
// bit 2 = no global loads after the prologue, bit 3 = s_memtime stamps around the phases of a step, summed per wave into
// SolveArgs::probe (the stamps share lgkmcnt with the LDS traffic and cost SGPRs, so the build is slower than the real one)
template <int TY, int ABL = 0>
__global__ __launch_bounds__(kLanes*(TY + 3)) void k_sweep7(SolveArgs a, F3dGeo g, int zchunk, int ntx, int nty, int n_tiles,
                                                             int xcd_remap)
{
  constexpr int NR = TY + 2;  // row waves: rows y0-1 .. y0+TY
  constexpr int kR7 = 8;      // ring depth of the DMA-fed halo buffers: a slot is rewritten 8 steps after it was read
  static_assert(TY <= 32, "the column wave holds one halo voxel per lane: 2 x TY <= 64");
  __shared__ float img0[2][kNL][NR + 2][kLanes];  // faces of a plane before sweep 1; rows 0 and NR+1: halo rows y0-2, y0+TY+1
  __shared__ float img1[2][3][NR][kLanes];        // S = U + dU after sweep 1, rows of the row waves
  __shared__ float hrow[kR7][2][9][kLanes];       // raw halo rows of the edge waves, by LDS-DMA
  __shared__ float hcol[kR7][NR][kLanes];         // raw x-halo of a row: [side*32 + near*16 + array], near = adjacent column
  __shared__ float hc1[2][3][2][32];              // S after sweep 1 in the two halo columns: [component][side][core row]
  FDivs fdivs = make_f_divs(a.hx, a.hy, a.hz);
  fdivs.ok = fdivs.ok && !a.plain_division;

  int tile = static_cast<int>(blockIdx.x);
  if (xcd_remap) {
    const int per_xcd = (n_tiles + 7) / 8;
    tile = (tile % 8) * per_xcd + tile / 8;
  }
  if (tile >= n_tiles) return;
  const int tx = tile % ntx;
  const int ty = (tile / ntx) % nty;
  const int tz = tile / (ntx * nty);

  const int lane = threadIdx.x;
  const int r = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  const bool colw = r == NR;
  const int z0 = g.z_lo + tz * zchunk;
  const int z1 = min(z0 + zchunk, g.z_hi);
  const int qs = z0 > 0 ? z0 - 1 : 0;        // first and last plane of stage 1
  const int qe = z1 < g.D ? z1 : g.D - 1;
  const int q_end = z1 < g.D ? qe : qe + 1;  // the top chunk takes one more step: stage 2 of plane D-1 alone
  const int x0 = tx * kLanes;
  const int y0 = ty * TY;
  const bool tile_at_x_face = __builtin_amdgcn_readfirstlane(static_cast<int>(tx == 0 || x0 + kLanes >= g.W)) != 0;

  // row waves
  const int y = y0 - 1 + r;
  const int yy = f3d_clampi(f3d_mir(y, g.H), 0, g.H - 1);
  const int x = x0 + lane;
  const int xi = f3d_clampi(f3d_mir(x, g.W), 0, g.W - 1);
  const unsigned xb = static_cast<unsigned>(xi) * 4u;
  const bool core = r >= 1 && r <= TY;
  const bool owner = core && x < g.W && y < g.H;
  const int side = lane < 32 ? 0 : 1;
  const bool edge = (r == 0) || (r == NR - 1);
  const int which = r == 0 ? 0 : 1;
  const int yh_row = f3d_clampi(f3d_mir(r == 0 ? y0 - 2 : y0 + TY + 1, g.H), 0, g.H - 1);
  const int lds_halo = r == 0 ? 0 : NR + 1;
  // column wave: lane = side * 32 + core row (lanes beyond TY rows repeat the last row and publish nothing)
  const bool cactive = (lane & 31) < TY;
  const int crow = cactive ? (lane & 31) : TY - 1;
  const int cy = y0 + crow;
  const int cx = side == 0 ? x0 - 1 : x0 + kLanes;

  const int zb = qs > 0 ? qs - 1 : 0;
  const size_t base_off = f3d_row(g, 0, zb);
  const unsigned plane_b = static_cast<unsigned>(g.Hc) * static_cast<unsigned>(g.pitch) * 4u;
  const unsigned row_b = static_cast<unsigned>(g.pitch) * 4u;
  constexpr int kOrder[10] = {F0, F1, U, V, Wf, DU, DV, DW, PHI, 9};  // order inside the halo rings
  const float* base[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) base[i] = a.in[i] + base_off;
  float* obase[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) obase[i] = a.out[i] + base_off;
  // x-halo gather: lane = side*32 + near*16 + array reads one value; columns x0-2, x0-1 | x0+64, x0+65
  const int garr = lane & 15;
  const int near = (lane >> 4) & 1;
  const bool col_lane = garr < 10;
  const float* lane_base = base[0];
#pragma unroll
  for (int i = 1; i < 10; ++i)
    if (garr == i) lane_base = base[kOrder[i]];
  lane_base += f3d_clampi(f3d_mir(side == 0 ? x0 - 2 + near : x0 + kLanes + 1 - near, g.W), 0, g.W - 1);

  auto rowoff = [&](int yrow, int zz) {
    return static_cast<unsigned>(__builtin_amdgcn_readfirstlane(
        static_cast<int>(static_cast<unsigned>(zz - zb) * plane_b + static_cast<unsigned>(yrow) * row_b)));
  };
  auto load_plane = [&](PlaneRegs& p, int zz) {
    const unsigned off = xb + rowoff(yy, zz);
    p.f0 = gld(base[F0], off);
    p.f1 = gld(base[F1], off);
    p.u = gld(base[U], off);
    p.v = gld(base[V], off);
    p.w = gld(base[Wf], off);
    p.su = gld(base[DU], off);
    p.dv = gld(base[DV], off);
    p.dw = gld(base[DW], off);
    p.phi = gld(base[PHI], off);
    p.ksi = gld(base[9], off);
  };
  auto dma_halos = [&](int q) {  // plane q (mirrored for the address), ring slot q mod 8
    const int slot = q & (kR7 - 1);
    const int zz = f3d_mir(q, g.D);
    if (col_lane) gld_lds_lane(lane_base + (rowoff(yy, zz) >> 2), &hcol[slot][r][0]);
    if (edge) {
      const unsigned off = xb + rowoff(yh_row, zz);
#pragma unroll
      for (int i = 0; i < 9; ++i) gld_lds(base[kOrder[i]], off, &hrow[slot][which][i][0]);
    }
  };
  auto from9 = [&](PlaneRegs& p, const float* d, int stride) {
    p.f0 = d[0 * stride]; p.f1 = d[1 * stride]; p.u = d[2 * stride]; p.v = d[3 * stride]; p.w = d[4 * stride];
    p.su = d[5 * stride]; p.dv = d[6 * stride]; p.dw = d[7 * stride]; p.phi = d[8 * stride];
  };
  // column wave: raw values of its voxel's column (near = 1) or of the column beyond (near = 0), row wave rw, plane q
  auto col_raw = [&](PlaneRegs& p, int q, int rw, int nr) {
    const float* d = &hcol[q & (kR7 - 1)][rw][side * 32 + nr * 16];
    from9(p, d, 1);
    p.ksi = d[9];
  };
  auto publish = [&](const PlaneRegs& pl, int q) {  // faces of plane q before sweep 1 -> img0[q & 1]
    const int nb = q & 1;
    const Face6 f = plane_face(pl);
#pragma unroll
    for (int i = 0; i < kNL; ++i) img0[nb][i][r + 1][lane] = f.v[i];
    if (edge) {
      PlaneRegs Hc;
      from9(Hc, &hrow[q & (kR7 - 1)][which][0][lane], kLanes);
      plane_finish(Hc);
      const Face6 hf = plane_face(Hc);
#pragma unroll
      for (int i = 0; i < kNL; ++i) img0[nb][i][lds_halo][lane] = hf.v[i];
    }
  };

  S3 hM = {0.f, 0.f, 0.f}, hC = {0.f, 0.f, 0.f};  // S after sweep 1 of planes q-2 and q-1
  float hC_dv = 0.f, hC_dw = 0.f;
  Carry kC = {};

  // M, C, P: finished planes q-1, q, q+1 (the column wave fills P at the start of the step).  Q: receives plane q+2,
  // requested at the top of the step and awaited at its end.  ONE plane in flight per wave, not two: with a dozen waves
  // per CU a second plane only parks the waves in the issue stage (tools/lab/issue_lab: 12 waves x 10 loads issue in
  // 10 cycles each, 12 x 20 in 40-95) and costs 10 registers.
  unsigned long long pr[6] = {0, 0, 0, 0, 0, 0};
  auto step = [&](PlaneRegs& M, PlaneRegs& C, PlaneRegs& P, PlaneRegs& Q, int q) {
    const bool more = q + 2 <= qe + 1;  // plane q+2 is still somebody's z-neighbour
    unsigned long long tk[7] = {0, 0, 0, 0, 0, 0, 0};
    if (ABL & 8) tk[0] = __builtin_amdgcn_s_memtime();
    if (!colw && more && !(ABL & 4)) {
      load_plane(Q, f3d_mir(q + 2, g.D));
      dma_halos(q + 2);
    }
    if (ABL & 8) tk[1] = __builtin_amdgcn_s_memtime();
    __syncthreads();  // img0 of plane q, img1 / hc1 of plane q-1 and the DMA rings up to plane q+1 are complete
    if (ABL & 8) tk[2] = __builtin_amdgcn_s_memtime();
    const bool do1 = q <= qe;
    const int b = q & 1;

    float r_du = 0.f, r_dv = 0.f, r_dw = 0.f;
    Carry kN = kC;
    if (do1) {
      Face6 ym, yp, xm, xp;
      const Face6 cf = plane_face(C);
      int vx, vy;
      if (colw) {
        col_raw(P, q + 1, crow + 1, 1);
        plane_finish(P);
        PlaneRegs T;
        col_raw(T, q, crow, 1);
        plane_finish(T);
        ym = plane_face(T);
        col_raw(T, q, crow + 2, 1);
        plane_finish(T);
        yp = plane_face(T);
        col_raw(T, q, crow + 1, 0);
        plane_finish(T);
        const Face6 outer = plane_face(T);
        Face6 inner;
#pragma unroll
        for (int i = 0; i < kNL; ++i) inner.v[i] = img0[b][i][crow + 2][side ? kLanes - 1 : 0];
#pragma unroll
        for (int i = 0; i < kNL; ++i) {
          xm.v[i] = side ? inner.v[i] : outer.v[i];
          xp.v[i] = side ? outer.v[i] : inner.v[i];
        }
        vx = cx;
        vy = cy;
      } else {
#pragma unroll
        for (int i = 0; i < kNL; ++i) {
          ym.v[i] = img0[b][i][r][lane];
          yp.v[i] = img0[b][i][r + 2][lane];
        }
        PlaneRegs X;
        from9(X, &hcol[q & (kR7 - 1)][r][side * 32 + 16], 1);
        if (q + 1 <= qe) publish(P, q + 1);
        plane_finish(X);
        const Face6 xf = plane_face(X);
#pragma unroll
        for (int i = 0; i < kNL; ++i) {
          xm.v[i] = lane_left_or(cf.v[i], xf.v[i]);
          xp.v[i] = lane_right_or(cf.v[i], xf.v[i]);
        }
        vx = x;
        vy = y;
      }
      if (ABL & 1) {
        r_du = xm.v[0] + xp.v[1] + ym.v[2] + yp.v[3] + M.su + P.sv + C.ksi;
        r_dv = xm.v[4] + xp.v[5] + ym.v[0] + yp.v[1] + M.f0 + P.f1 + C.u;
        r_dw = xm.v[2] + xp.v[3] + ym.v[4] + yp.v[5] + M.phi + P.phi + C.dv + C.dw + C.v + C.w;
        kN.J12 = r_du; kN.d1 = r_dv; kN.pw[0] = r_dw;
      } else
      sweep_stage1(xm, xp, ym, yp, plane_face(M), plane_face(P), cf.v, C.u, C.v, C.w, C.dv, C.dw, C.ksi, a.hx, a.hy, a.hz,
                   fdivs, a.p0, vx < g.W - 1, vx > 0, vy < g.H - 1, vy > 0, q < g.D - 1, q > 0, r_du, r_dv, r_dw, kN);
    }
    const S3 sN = {C.u + r_du, C.v + r_dv, C.w + r_dw};  // what a neighbour reads after sweep 1: U + dU'
    if (do1) {
      if (colw) {
        if (cactive) {
          hc1[b][0][side][crow] = sN.u;
          hc1[b][1][side][crow] = sN.v;
          hc1[b][2][side][crow] = sN.w;
        }
      } else {
        img1[b][0][r][lane] = sN.u;
        img1[b][1][r][lane] = sN.v;
        img1[b][2][r][lane] = sN.w;
      }
    }

    if (ABL & 8) {
      asm volatile("" ::"v"(r_du), "v"(r_dv), "v"(r_dw));
      __builtin_amdgcn_sched_barrier(0);
      tk[3] = __builtin_amdgcn_s_memtime();
    }
    // stage 2 of plane t = q - 1
    const int t = q - 1;
    const bool do2 = core && t >= z0;
    float o_du = 0.f, o_dv = 0.f, o_dw = 0.f;
    if (do2) {
      const int pb = t & 1;
      S3 ym, yp, xm, xp, zm, zp;
      ym.u = img1[pb][0][r - 1][lane]; ym.v = img1[pb][1][r - 1][lane]; ym.w = img1[pb][2][r - 1][lane];
      yp.u = img1[pb][0][r + 1][lane]; yp.v = img1[pb][1][r + 1][lane]; yp.w = img1[pb][2][r + 1][lane];
      const float eu = hc1[pb][0][side][r - 1], ev = hc1[pb][1][side][r - 1], ew = hc1[pb][2][side][r - 1];
      xm.u = lane_left_or(hC.u, eu); xm.v = lane_left_or(hC.v, ev); xm.w = lane_left_or(hC.w, ew);
      xp.u = lane_right_or(hC.u, eu); xp.v = lane_right_or(hC.v, ev); xp.w = lane_right_or(hC.w, ew);
      zm = hM;
      zp = sN;
      // mirror rule at the faces of the volume: the missing neighbour is the opposite one.  Only tiles that touch a face
      // pay for the selects (wave-uniform branches).
      if (tile_at_x_face) {
        if (x == 0) xm = xp;
        if (x == g.W - 1) xp = xm;
      }
      if (y == 0) ym = yp;
      if (y == g.H - 1) yp = ym;
      if (t == 0) zm = zp;
      if (t == g.D - 1) zp = zm;
      if (ABL & 2) {
        o_du = xm.u + xp.v + ym.w + kC.J12;
        o_dv = yp.u + zm.v + zp.w + kC.d1;
        o_dw = xm.w + yp.v + zp.u + kC.pw[0] + hC_dv + hC_dw;
      } else
      sweep_stage2(kC, xm, xp, ym, yp, zm, zp, hC_dv, hC_dw, o_du, o_dv, o_dw);
    }
    asm volatile("" ::"v"(o_du), "v"(o_dv), "v"(o_dw), "v"(sN.u), "v"(sN.v), "v"(sN.w));
    hM = hC;
    hC = sN;
    hC_dv = r_dv;
    hC_dw = r_dw;
    kC = kN;
    __builtin_amdgcn_sched_barrier(0);
    if (ABL & 8) tk[4] = __builtin_amdgcn_s_memtime();
    // plane q+2 and its halos (requested at the top of this step) and the stores of the last step have landed
    if (!colw) {
      F3D_WAIT_PLANE(0, Q);
      plane_finish(Q);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ABL & 8) tk[5] = __builtin_amdgcn_s_memtime();
    if (do2 && owner) {
      const unsigned off = xb + rowoff(yy, t);
      gst(obase[0], off, o_du);
      gst(obase[1], off, o_dv);
      gst(obase[2], off, o_dw);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ABL & 8) {
      tk[6] = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int i = 0; i < 6; ++i) pr[i] += tk[i + 1] - tk[i];
    }
  };

  PlaneRegs A = {}, B = {}, C = {}, D = {};
  if (!colw) {
    load_plane(A, f3d_mir(qs - 1, g.D));
    load_plane(B, qs);
    load_plane(C, f3d_mir(qs + 1, g.D));
    dma_halos(qs - 1);
    dma_halos(qs);
    dma_halos(qs + 1);
    F3D_WAIT_PLANE(0, A);
    F3D_WAIT_PLANE(0, B);
    F3D_WAIT_PLANE(0, C);
    plane_finish(A);
    plane_finish(B);
    plane_finish(C);
  }
  __syncthreads();  // DMA-written rings are visible
  if (colw) {
    col_raw(A, qs - 1, crow + 1, 1);
    col_raw(B, qs, crow + 1, 1);
    plane_finish(A);
    plane_finish(B);
  } else {
    publish(B, qs);
  }
  __builtin_amdgcn_sched_barrier(0);
  int q = qs;
  for (; q + 3 <= q_end; q += 4) {
    step(A, B, C, D, q);
    step(B, C, D, A, q + 1);
    step(C, D, A, B, q + 2);
    step(D, A, B, C, q + 3);
  }
  if (q <= q_end) step(A, B, C, D, q);
  if (q + 1 <= q_end) step(B, C, D, A, q + 1);
  if (q + 2 <= q_end) step(C, D, A, B, q + 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores issued by hand
  if ((ABL & 8) && a.probe && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) atomicAdd(&a.probe[r * 8 + i], pr[i]);
    atomicAdd(&a.probe[r * 8 + 6], static_cast<unsigned long long>(q_end - qs + 1));
  }
}

struct Tuning {
  int xcd_remap;   // deal tiles to the 8 XCDs in contiguous runs (1) or leave the hardware's round robin (0)
  int zchunk;      // planes per z-chunk, 0 = automatic
  long want_wg;    // k_sweep6 / k_phiksi6: workgroups to aim at when cutting z-chunks
};

const Tuning& tuning()
{
  static const Tuning t = [] {
    Tuning v = {1, 0, 4096};
    if (const char* e = std::getenv("F3D_XCD_REMAP")) v.xcd_remap = std::atoi(e);
    if (const char* e = std::getenv("F3D_ZCHUNK")) v.zchunk = std::atoi(e);
    if (const char* e = std::getenv("F3D_WANT_WG")) v.want_wg = std::atol(e);
    return v;
  }();
  return t;
}

// The kernels address a z-chunk with 32-bit byte offsets from the chunk's first plane: a chunk, with the halo planes it
// reads on either side, must stay below 4 GiB of one array.  Planes of a container larger than 1024 x 1024 floats leave room
// for fewer than 1024 of them.
inline int max_planes_per_chunk(const F3dGeo& g)
{
  static const bool off = std::getenv("F3D_CHUNK_LIMIT") && std::atoi(std::getenv("F3D_CHUNK_LIMIT")) == 0;
  if (off) return 0x3fffffff;  // only to show that tests/test_gpu_big.py catches the overflow
  const unsigned long long plane_bytes = static_cast<unsigned long long>(g.Hc) * static_cast<unsigned long long>(g.pitch) * 4ull;
  const long long planes = static_cast<long long>(0xffffffffull / plane_bytes) - 8;  // 3-4 halo planes each way and slack
  return planes < 1 ? 1 : (planes > 0x3fffffff ? 0x3fffffff : static_cast<int>(planes));
}

// Frame derivatives of a level, once: fx, fy, fz as A.3 / A.4 form them (the numerator left to right, one IEEE division by 4h)
// and ft = F1 - F0.  They depend on the two frames of the level only, which no solver launch changes, so the 120 launches of a
// level that would each recompute them read them instead (k_pair8 with FD).
__global__ __launch_bounds__(256) void k_frame_derivatives(const float* __restrict__ f0, const float* __restrict__ f1, F3dGeo g, float hx,
                                                           float hy, float hz, float* __restrict__ fx, float* __restrict__ fy,
                                                           float* __restrict__ fz, float* __restrict__ ft)
{
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int z = g.z_lo + blockIdx.z;
  if (x >= g.W || y >= g.H) return;
  const size_t c = f3d_row(g, y, z) + x;
  const size_t xm = f3d_row(g, y, z) + f3d_mir(x - 1, g.W), xp = f3d_row(g, y, z) + f3d_mir(x + 1, g.W);
  const size_t ym = f3d_row(g, f3d_mir(y - 1, g.H), z) + x, yp = f3d_row(g, f3d_mir(y + 1, g.H), z) + x;
  const size_t zm = f3d_row(g, y, f3d_mir(z - 1, g.D)) + x, zp = f3d_row(g, y, f3d_mir(z + 1, g.D)) + x;
  fx[c] = (f0[xp] - f0[xm] + f1[xp] - f1[xm]) / (4.f * hx);
  fy[c] = (f0[yp] - f0[ym] + f1[yp] - f1[ym]) / (4.f * hy);
  fz[c] = (f0[zp] - f0[zm] + f1[zp] - f1[zm]) / (4.f * hz);
  ft[c] = f1[c] - f0[c];
}

#include "f3d_solve_pair8.h"
#include "f3d_solve_tri.h"

// k_sweep6 (SWEEP) or k_phiksi6: 64 x 8 tiles, z cut into chunks so that even a coarse pyramid level spreads over all CUs
template <bool SWEEP>
void launch_solver(const SolveArgs& args, const F3dGeo& g)
{
  const Tuning& t = tuning();
  const int planes = g.z_hi - g.z_lo;
  const int ntx = (g.W + kLanes - 1) / kLanes;
  const int nty = (g.H + kTY3 - 1) / kTY3;
  long nzc = (t.want_wg + static_cast<long>(ntx) * nty - 1) / (static_cast<long>(ntx) * nty);
  const long max_chunks = planes / 4 > 0 ? planes / 4 : 1;
  if (nzc > max_chunks) nzc = max_chunks;
  if (nzc < 1) nzc = 1;
  int zchunk = static_cast<int>((planes + nzc - 1) / nzc);
  // Small launches (up to 128^3) are bound by the latency of the serial march, ~1.35 us per plane step after ~6 us, not by
  // bandwidth: there the chunk is the one that minimises rounds x (planes per chunk + 4.5), 512 workgroups running per
  // round (64^3: 2-plane chunks, 11.3 -> 8.8 us per sweep; measured with tools/kbench.py, F3D_ZCHUNK sweeps).
  const long tiles_xy = static_cast<long>(ntx) * nty;
  if (tiles_xy * planes <= 4096) {
    double best = -1.0;
    for (int zc = 1; zc <= planes; ++zc) {
      const long wgs = tiles_xy * ((planes + zc - 1) / zc);
      const double cost = static_cast<double>((wgs + 511) / 512) * (zc + 4.5);
      if (best < 0.0 || cost < best) {
        best = cost;
        zchunk = zc;
      }
    }
  }
  else if (planes <= 128) {
    // A window of few planes over a wide level -- a z-slab of the multi-GPU driver: cutting it into 4096 workgroups' worth
    // of chunks makes every chunk re-read its two halo planes for a handful of its own (10-plane chunks on a 74-plane slab
    // of 512 x 512: +20 % planes).  One round of resident workgroups (512) is enough here: 512 x 512 x 74 phi/ksi
    // 236 -> 206 us, sweep 256 -> 223 us with a single chunk.
    const long chunks = std::max<long>(1, (512 + tiles_xy - 1) / tiles_xy);
    zchunk = static_cast<int>((planes + chunks - 1) / chunks);
  }
  if (t.zchunk > 0) zchunk = t.zchunk;
  zchunk = std::min(zchunk, max_planes_per_chunk(g));
  SolveArgs a = args;
  int nz = (planes + zchunk - 1) / zchunk;
  if (!SWEEP && a.z2_hi > a.z2_lo) {  // the chunks of the second window follow those of the first
    a.nz_first = nz;
    nz += (a.z2_hi - a.z2_lo + zchunk - 1) / zchunk;
  }
  const int n_tiles = ntx * nty * nz;
  const int per_xcd = (n_tiles + 7) / 8;
  const int blocks = t.xcd_remap ? per_xcd * 8 : n_tiles;
  const dim3 grid(blocks, 1, 1), block(kLanes, kTY3, 1);
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, block, 0, f3d::stream(), a, g, zchunk, ntx, nty, n_tiles, t.xcd_remap); };
  if constexpr (SWEEP) {
#ifdef F3D_LAB  // timing builds that skip parts of the work (WRONG results): only in lib/lab/libf3d_hip.so (make lab), never in the product
    static const int ablate = std::getenv("F3D_ABLATE") ? std::atoi(std::getenv("F3D_ABLATE")) : 0;
    if (ablate == 1) return go(k_sweep6<1, kTY3>);
    if (ablate == 2) return go(k_sweep6<2, kTY3>);
    if (ablate == 3) return go(k_sweep6<3, kTY3>);
#endif
    go(k_sweep6<0, kTY3>);
  } else {
    go(k_phiksi6);
  }
}

template <int TY>
void launch_sweep2_ty(const SolveArgs& a, const F3dGeo& g, long want_wg, int force_zchunk, int xcd_remap)
{
  const int planes = g.z_hi - g.z_lo;
  const int ntx = (g.W + kLanes - 1) / kLanes;
  const int nty = (g.H + TY - 1) / TY;
  // One workgroup per CU at a time (LDS, registers), so the chunking is chosen by a round model: a z-chunk costs its
  // planes plus ~6 steps of prologue and repeated stage-1 planes, and 256 workgroups run per round.
  const long tiles = static_cast<long>(ntx) * nty;
  // chunks of at least two planes: a coarse level is latency-bound and prefers many short marches (64^3: 2-plane chunks in
  // one round, 28 -> 16 us per launch), a fine one never gets near the limit
  const int max_chunks = planes / 2 > 0 ? planes / 2 : 1;
  const int zc_limit = max_planes_per_chunk(g);
  int zchunk = std::min(planes, zc_limit);
  if (want_wg > 0) {  // experiments: aim at a workgroup count
    long nzc = (want_wg + tiles - 1) / tiles;
    if (nzc > max_chunks) nzc = max_chunks;
    if (nzc < 1) nzc = 1;
    zchunk = static_cast<int>((planes + nzc - 1) / nzc);
  } else {
    long best = -1;
    for (int nzc = 1; nzc <= max_chunks; ++nzc) {
      const int zc = (planes + nzc - 1) / nzc;
      if (zc > zc_limit) continue;
      const long wgs = tiles * ((planes + zc - 1) / zc);
      const long cost = ((wgs + 255) / 256) * (zc + 6);
      if (best < 0 || cost < best) {
        best = cost;
        zchunk = zc;
      }
    }
  }
  if (force_zchunk > 0) zchunk = force_zchunk;
  zchunk = std::min(zchunk, zc_limit);
  const int nz = (planes + zchunk - 1) / zchunk;
  const int n_tiles = ntx * nty * nz;
  const int per_xcd = (n_tiles + 7) / 8;
  const int blocks = xcd_remap ? per_xcd * 8 : n_tiles;
  const dim3 grid(blocks, 1, 1), block(kLanes, TY + 3, 1);
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, block, 0, f3d::stream(), a, g, zchunk, ntx, nty, n_tiles, xcd_remap); };
#ifdef F3D_LAB  // timing builds (WRONG results): lab library only
  static const int abl = std::getenv("F3D_ABLATE7") ? std::atoi(std::getenv("F3D_ABLATE7")) : 0;
  if constexpr (TY == 9) {
    if (abl == 1) return go(k_sweep7<TY, 1>);
    if (abl == 2) return go(k_sweep7<TY, 2>);
    if (abl == 3) return go(k_sweep7<TY, 3>);
    if (abl == 4) return go(k_sweep7<TY, 4>);
    if (abl == 7) return go(k_sweep7<TY, 7>);
    if (abl == 8) {
      static unsigned long long* probe = nullptr;
      if (!probe) (void)hipMalloc(reinterpret_cast<void**>(&probe), 16 * 8 * sizeof(unsigned long long));
      (void)hipMemsetAsync(probe, 0, 16 * 8 * sizeof(unsigned long long), f3d::stream());
      SolveArgs ap = a;
      ap.probe = probe;
      hipLaunchKernelGGL((k_sweep7<TY, 8>), grid, block, 0, f3d::stream(), ap, g, zchunk, ntx, nty, n_tiles, xcd_remap);
      unsigned long long h[16 * 8];
      (void)hipMemcpyAsync(h, probe, sizeof(h), hipMemcpyDeviceToHost, f3d::stream());
      (void)hipStreamSynchronize(f3d::stream());
      static int shown = 0;
      if (shown++ < 1) {
        std::fprintf(stderr, "k_sweep7 probe (avg cycles per step): wave  issue  barrier  stage1  stage2  vmwait  store\n");
        for (int w = 0; w < TY + 3; ++w) {
          const double n = static_cast<double>(h[w * 8 + 6]);
          std::fprintf(stderr, "  %2d  %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f   (%.0f steps)\n", w, h[w * 8] / n, h[w * 8 + 1] / n,
                       h[w * 8 + 2] / n, h[w * 8 + 3] / n, h[w * 8 + 4] / n, h[w * 8 + 5] / n, n);
        }
      }
      return;
    }
  }
#endif
  go(k_sweep7<TY, 0>);
}

// 0 = k_sweep7 (every row wave loads its own row), 1 = k_pair8 (DMA loader wave); F3D_PAIR8 overrides
int pair8_enabled()
{
  static const int v = std::getenv("F3D_PAIR8") ? std::atoi(std::getenv("F3D_PAIR8")) : 1;
  return v;
}

PairArgs pair_args(const SolveArgs& a)
{
  PairArgs p = {};
  for (int i = 0; i < 10; ++i) p.in[i] = a.in[i];
  for (int i = 0; i < 3; ++i) p.out[i] = a.out[i];
  p.hx = a.hx; p.hy = a.hy; p.hz = a.hz;
  p.alpha = a.p0;
  p.plain_division = a.plain_division;
  return p;
}

// 8 or 12 core rows per tile of k_pair8 (12 or 16 waves per workgroup), whichever the round model prices lower for this level;
// F3D_PAIR8_TY=8 / 12 pins one, F3D_PAIR8_STEP12 = cost of a 16-wave step in % of a 12-wave one (timing experiments)
int pair8_rows(const F3dGeo& g)
{
  // read per call: the tests force every tile height on small shapes (a getenv per launch is ~50 ns)
  const char* forced_env = std::getenv("F3D_PAIR8_TY");
  const int forced = forced_env ? std::atoi(forced_env) : 0;
  static const long step12 = std::getenv("F3D_PAIR8_STEP12") ? std::atol(std::getenv("F3D_PAIR8_STEP12")) : 128;
  if (forced == 4 || forced == 8 || forced == 12) return forced;
  const long c8 = pair8_plan(g, 8).cost * 100, c12 = pair8_plan(g, 12).cost * step12;
  // 4 rows (8 waves): where a level is so small that every workgroup has a CU to itself, a step of two waves per SIMD takes
  // ~0.8 of a 12-wave step (two sweeps at 24^3 12.9 -> 10.7 us, 40^3 13.2 -> 11.0 us, 64^3 14.1 -> 13.5 us; from 72^3 up 8 rows win); two such
  // workgroups on one CU would fit (80 KB of LDS each) but gain nothing, so the same one-per-CU round model prices them
  static const long step4 = std::getenv("F3D_PAIR8_STEP4") ? std::atol(std::getenv("F3D_PAIR8_STEP4")) : 80;
  if (pair8_plan(g, 4).cost * step4 < std::min(c8, c12)) return 4;
  return c12 < c8 ? 12 : 8;
}

void launch_sweep2(const SolveArgs& a, const F3dGeo& g)
{
  const Tuning& t = tuning();
  if (pair8_enabled() && g.pitch % kLanes == 0) {  // the loader fetches whole 64-float row segments in 16-byte pieces
    if (launch_pair8_ymarch<PAIR_SS>(pair_args(a), g, t.zchunk, t.xcd_remap)) return;
    const int ty = pair8_rows(g);
    if (ty == 4) launch_pair8<PAIR_SS, 4>(pair_args(a), g, t.zchunk, t.xcd_remap);
    else if (ty == 12) launch_pair8<PAIR_SS, 12>(pair_args(a), g, t.zchunk, t.xcd_remap);
    else launch_pair8<PAIR_SS, 8>(pair_args(a), g, t.zchunk, t.xcd_remap);
    return;
  }
  static const int ty = std::getenv("F3D_SWEEP2_TY") ? std::atoi(std::getenv("F3D_SWEEP2_TY")) : 9;
  static const long want = std::getenv("F3D_SWEEP2_WG") ? std::atol(std::getenv("F3D_SWEEP2_WG")) : 0;
  if (ty == 8) launch_sweep2_ty<8>(a, g, want, t.zchunk, t.xcd_remap);
  else if (ty == 5) launch_sweep2_ty<5>(a, g, want, t.zchunk, t.xcd_remap);
  else launch_sweep2_ty<9>(a, g, want, t.zchunk, t.xcd_remap);
}


// 4 or 7 owned rows per tile of k_tri (9 or 12 waves per workgroup), whichever the round model prices lower for this level; a 9-wave
// step is taken at 80 % of a 12-wave one, as for k_pair8.  F3D_TRI_TY=4 / 7 pins one (read per call: the tests force both)
int tri_rows(const F3dGeo& g)
{
  const char* forced_env = std::getenv("F3D_TRI_TY");
  const int forced = forced_env ? std::atoi(forced_env) : 0;
  if (forced == 4 || forced == 7) return forced;
  static const long step4 = std::getenv("F3D_TRI_STEP4") ? std::atol(std::getenv("F3D_TRI_STEP4")) : 80;
  const int planes = g.z_hi - g.z_lo;
  const long c4 = tri_plan_dims(g.W, g.H, planes, 4, max_planes_per_chunk(g)).cost * step4;
  const long c7 = tri_plan_dims(g.W, g.H, planes, 7, max_planes_per_chunk(g)).cost * 100;
  return c4 < c7 ? 4 : 7;
}

// the two three-stage launches (f3d_solve_sweep3, f3d_solve_sweep2_phi_ksi)
int tri_launch(const char* who, bool with_weights, const f3d_devptr (&in)[10], size_t width, size_t height, size_t depth, float hx, float hy,
               float hz, float alpha, float eps_s, float eps_d, const f3d_devptr (&out)[5], const f3d_slab* slab)
{
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, who)) return 1;
  if (g.W < 2 || g.H < 2 || g.D < 2) return f3d::fail("%s: every dimension must be at least 2", who);
  if (!pair_weights_finite(hx, hy, hz, alpha))
    return f3d::fail("%s: alpha / h^2 must be finite and not negative (alpha %g, h %g %g %g)", who, alpha, hx, hy, hz);
  if (g.pitch % kLanes != 0) return f3d::fail("%s: the container pitch must be a multiple of 256 bytes (f3d_alloc_pitched gives that)", who);
  for (int i = 0; i < (with_weights ? 5 : 3); ++i)
    for (int k = 0; k < 10; ++k)
      if (out[i] == in[k]) return f3d::fail("%s: an output buffer is also an input (other tiles still read it)", who);
  if (g.z_lo == g.z_hi) return 0;
  if (!slab_reach_ok(g, 3, who)) return 1;
  PairArgs a = {};
  static const int plain_division = std::getenv("F3D_UDIV") && std::atoi(std::getenv("F3D_UDIV")) == 0;
  a.plain_division = plain_division;
  for (int i = 0; i < 10; ++i) a.in[i] = f3d_ptr<const float>(in[i]);
  for (int i = 0; i < 5; ++i) a.out[i] = f3d_ptr<float>(out[i]);
  a.hx = hx; a.hy = hy; a.hz = hz;
  a.alpha = alpha;
  a.eps_s = eps_s;
  a.eps_d = eps_d;
  const int kid = with_weights ? F3D_K_SWEEP2_PHI_KSI : F3D_K_SWEEP3;
  f3d::prof_begin(kid, static_cast<size_t>(g.W) * g.H * (g.z_hi - g.z_lo));
  const int rows = tri_rows(g);
  if (with_weights) {
    if (rows == 4) launch_tri<TRI_SSP, 4>(a, g, tuning().zchunk, tuning().xcd_remap);
    else launch_tri<TRI_SSP, 7>(a, g, tuning().zchunk, tuning().xcd_remap);
  } else {
    if (rows == 4) launch_tri<TRI_SSS, 4>(a, g, tuning().zchunk, tuning().xcd_remap);
    else launch_tri<TRI_SSS, 7>(a, g, tuning().zchunk, tuning().xcd_remap);
  }
  f3d::prof_end(kid);
  F3D_HIP(hipGetLastError());
  return 0;
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
  static const int plain_division = std::getenv("F3D_UDIV") && std::atoi(std::getenv("F3D_UDIV")) == 0;
  a.plain_division = plain_division;
  const f3d_devptr in[8] = {frame_0, frame_1, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw};
  for (int i = 0; i < 8; ++i) a.in[i] = f3d_ptr<const float>(in[i]);
  a.in[8] = a.in[9] = nullptr;
  a.out[0] = f3d_ptr<float>(phi);
  a.out[1] = f3d_ptr<float>(ksi);
  a.out[2] = nullptr;
  a.hx = hx; a.hy = hy; a.hz = hz;
  a.p0 = equation_smoothness;
  a.p1 = equation_data;
  f3d::prof_begin(F3D_K_PHI_KSI, static_cast<size_t>(g.W) * g.H * (g.z_hi - g.z_lo));
  launch_solver<false>(a, g);
  f3d::prof_end(F3D_K_PHI_KSI);
  F3D_HIP(hipGetLastError());
  return 0;
}

int f3d_phi_ksi_zones(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                      f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, size_t width, size_t height, size_t depth, float hx,
                      float hy, float hz, float equation_smoothness, float equation_data, f3d_devptr phi, f3d_devptr ksi,
                      const f3d_slab* zone_a, const f3d_slab* zone_b)
{
  F3D_REQUIRE_READY("f3d_phi_ksi_zones");
  if (!zone_a || !zone_b) return f3d::fail("f3d_phi_ksi_zones: two windows are required");
  if (zone_a->z_base != zone_b->z_base) return f3d::fail("f3d_phi_ksi_zones: both windows must address the same container planes (z_base)");
  F3dGeo g, g2;
  if (!f3d::make_geo(&g, width, height, depth, zone_a, "f3d_phi_ksi_zones")) return 1;
  if (!f3d::make_geo(&g2, width, height, depth, zone_b, "f3d_phi_ksi_zones")) return 1;
  if (g.W < 2 || g.H < 2 || g.D < 2) return f3d::fail("f3d_phi_ksi_zones: every dimension must be at least 2");
  if (g.z_lo == g.z_hi || g2.z_lo == g2.z_hi)  // one of them empty: the ordinary launch on the other
    return f3d_phi_ksi(frame_0, frame_1, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw, width, height, depth, hx, hy, hz,
                       equation_smoothness, equation_data, phi, ksi, g.z_lo == g.z_hi ? zone_b : zone_a);
  if (!(g.z_hi <= g2.z_lo || g2.z_hi <= g.z_lo)) return f3d::fail("f3d_phi_ksi_zones: the windows overlap");
  if (!slab_reach_ok(g, 1, "f3d_phi_ksi_zones") || !slab_reach_ok(g2, 1, "f3d_phi_ksi_zones")) return 1;
  SolveArgs a;
  static const int plain_division = std::getenv("F3D_UDIV") && std::atoi(std::getenv("F3D_UDIV")) == 0;
  a.plain_division = plain_division;
  const f3d_devptr in[8] = {frame_0, frame_1, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw};
  for (int i = 0; i < 8; ++i) a.in[i] = f3d_ptr<const float>(in[i]);
  a.in[8] = a.in[9] = nullptr;
  a.out[0] = f3d_ptr<float>(phi);
  a.out[1] = f3d_ptr<float>(ksi);
  a.out[2] = nullptr;
  a.hx = hx; a.hy = hy; a.hz = hz;
  a.p0 = equation_smoothness;
  a.p1 = equation_data;
  a.z2_lo = g2.z_lo;
  a.z2_hi = g2.z_hi;
  f3d::prof_begin(F3D_K_PHI_KSI, static_cast<size_t>(g.W) * g.H * (g.z_hi - g.z_lo + g2.z_hi - g2.z_lo));
  launch_solver<false>(a, g);
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
  static const int plain_division = std::getenv("F3D_UDIV") && std::atoi(std::getenv("F3D_UDIV")) == 0;
  a.plain_division = plain_division;
  const f3d_devptr in[10] = {frame_0, frame_1, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw, phi, ksi};
  for (int i = 0; i < 10; ++i) a.in[i] = f3d_ptr<const float>(in[i]);
  a.out[0] = f3d_ptr<float>(temp_du);
  a.out[1] = f3d_ptr<float>(temp_dv);
  a.out[2] = f3d_ptr<float>(temp_dw);
  a.hx = hx; a.hy = hy; a.hz = hz;
  a.p0 = equation_alpha;
  a.p1 = 0.f;
  f3d::prof_begin(F3D_K_SWEEP, static_cast<size_t>(g.W) * g.H * (g.z_hi - g.z_lo));
  launch_solver<true>(a, g);
  f3d::prof_end(F3D_K_SWEEP);
  F3D_HIP(hipGetLastError());
  return 0;
}

int f3d_solve_sweep2(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                     f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi,
                     size_t width, size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha,
                     f3d_devptr temp_du, f3d_devptr temp_dv, f3d_devptr temp_dw, const f3d_slab* slab)
{
  F3D_REQUIRE_READY("f3d_solve_sweep2");
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, "f3d_solve_sweep2")) return 1;
  if (g.W < 2 || g.H < 2 || g.D < 2) return f3d::fail("f3d_solve_sweep2: every dimension must be at least 2");
  if (!pair_weights_finite(hx, hy, hz, equation_alpha))
    return f3d::fail("f3d_solve_sweep2: alpha / h^2 must be finite and not negative (alpha %g, h %g %g %g)", equation_alpha, hx, hy, hz);
  if (g.z_lo == g.z_hi) return 0;
  if (!slab_reach_ok(g, 2, "f3d_solve_sweep2")) return 1;
  SolveArgs a;
  static const int plain_division = std::getenv("F3D_UDIV") && std::atoi(std::getenv("F3D_UDIV")) == 0;
  a.plain_division = plain_division;
  const f3d_devptr in[10] = {frame_0, frame_1, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw, phi, ksi};
  for (int i = 0; i < 10; ++i) a.in[i] = f3d_ptr<const float>(in[i]);
  a.out[0] = f3d_ptr<float>(temp_du);
  a.out[1] = f3d_ptr<float>(temp_dv);
  a.out[2] = f3d_ptr<float>(temp_dw);
  a.hx = hx; a.hy = hy; a.hz = hz;
  a.p0 = equation_alpha;
  a.p1 = 0.f;
  f3d::prof_begin(F3D_K_SWEEP2, static_cast<size_t>(g.W) * g.H * (g.z_hi - g.z_lo));
  launch_sweep2(a, g);
  f3d::prof_end(F3D_K_SWEEP2);
  F3D_HIP(hipGetLastError());
  return 0;
}

int f3d_solve_sweep_phi_ksi_edges(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                                  f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi,
                                  size_t width, size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha,
                                  float equation_smoothness, float equation_data, f3d_devptr temp_du, f3d_devptr temp_dv,
                                  f3d_devptr temp_dw, f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab,
                                  int keep_below, int keep_above)
{
  F3D_REQUIRE_READY("f3d_solve_sweep_phi_ksi");
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, "f3d_solve_sweep_phi_ksi")) return 1;
  if (g.W < 2 || g.H < 2 || g.D < 2) return f3d::fail("f3d_solve_sweep_phi_ksi: every dimension must be at least 2");
  if (!pair_weights_finite(hx, hy, hz, equation_alpha))
    return f3d::fail("f3d_solve_sweep_phi_ksi: alpha / h^2 must be finite and not negative (alpha %g, h %g %g %g)", equation_alpha, hx, hy, hz);
  if (phi_next == phi || ksi_next == ksi || phi_next == ksi || ksi_next == phi)
    return f3d::fail("f3d_solve_sweep_phi_ksi: phi_next / ksi_next must not alias phi / ksi (other tiles still read them)");
  if (g.pitch % kLanes != 0)
    return f3d::fail("f3d_solve_sweep_phi_ksi: the container pitch must be a multiple of 256 bytes (f3d_alloc_pitched gives that)");
  if (g.z_lo == g.z_hi) return 0;
  if (!slab_reach_ok(g, 2, "f3d_solve_sweep_phi_ksi")) return 1;
  PairArgs a = {};
  a.keep_below = keep_below != 0;
  a.keep_above = keep_above != 0;
  static const int plain_division = std::getenv("F3D_UDIV") && std::atoi(std::getenv("F3D_UDIV")) == 0;
  a.plain_division = plain_division;
  const f3d_devptr in[10] = {frame_0, frame_1, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw, phi, ksi};
  for (int i = 0; i < 10; ++i) a.in[i] = f3d_ptr<const float>(in[i]);
  a.out[0] = f3d_ptr<float>(temp_du);
  a.out[1] = f3d_ptr<float>(temp_dv);
  a.out[2] = f3d_ptr<float>(temp_dw);
  a.out[3] = f3d_ptr<float>(phi_next);
  a.out[4] = f3d_ptr<float>(ksi_next);
  a.hx = hx; a.hy = hy; a.hz = hz;
  a.alpha = equation_alpha;
  a.eps_s = equation_smoothness;
  a.eps_d = equation_data;
  f3d::prof_begin(F3D_K_SWEEP_PHI_KSI, static_cast<size_t>(g.W) * g.H * (g.z_hi - g.z_lo));
  if (!a.keep_below && !a.keep_above && launch_pair8_ymarch<PAIR_SP>(a, g, tuning().zchunk, tuning().xcd_remap)) {}
  else if (pair8_rows(g) == 4) launch_pair8<PAIR_SP, 4>(a, g, tuning().zchunk, tuning().xcd_remap);
  else if (pair8_rows(g) == 12) launch_pair8<PAIR_SP, 12>(a, g, tuning().zchunk, tuning().xcd_remap);
  else launch_pair8<PAIR_SP, 8>(a, g, tuning().zchunk, tuning().xcd_remap);
  f3d::prof_end(F3D_K_SWEEP_PHI_KSI);
  F3D_HIP(hipGetLastError());
  return 0;
}

int f3d_solve_sweep_phi_ksi(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                            f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi,
                            size_t width, size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha,
                            float equation_smoothness, float equation_data, f3d_devptr temp_du, f3d_devptr temp_dv,
                            f3d_devptr temp_dw, f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab)
{
  return f3d_solve_sweep_phi_ksi_edges(frame_0, frame_1, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw, phi, ksi, width, height,
                                       depth, hx, hy, hz, equation_alpha, equation_smoothness, equation_data, temp_du, temp_dv,
                                       temp_dw, phi_next, ksi_next, slab, 0, 0);
}

namespace {
struct WeightTally {
  unsigned long long checked, excluded, mismatches;
  unsigned first[4];
};
// every bit pattern in [lo, hi]: weight_fast (where weight_fast_ok lets it through) against the IEEE chain the reference's
// expression compiles to -- the same device functions the fused kernel calls, in the same translation unit
__global__ __launch_bounds__(256) void k_selftest_weights(unsigned lo, unsigned hi, WeightTally* t)
{
  const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
  unsigned long long checked = 0, excluded = 0;
  for (unsigned long long i = lo + static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i <= hi; i += stride) {
    const float a = __uint_as_float(static_cast<unsigned>(i));
    float s;
    const float fast = weight_fast(a, s);
    if (!weight_fast_ok(a, s)) {
      ++excluded;
      continue;
    }
    ++checked;
    if (__float_as_uint(fast) != __float_as_uint(weight_ieee(a))) {
      const unsigned long long n = atomicAdd(&t->mismatches, 1ull);
      if (n < 4) t->first[n] = static_cast<unsigned>(i);
    }
  }
  atomicAdd(&t->checked, checked);
  atomicAdd(&t->excluded, excluded);
}
}  // namespace

int f3d_selftest_weights(unsigned lo_bits, unsigned hi_bits, unsigned long long* checked, unsigned long long* excluded,
                         unsigned long long* mismatches, unsigned* first_mismatch)
{
  F3D_REQUIRE_READY("f3d_selftest_weights");
  if (lo_bits > hi_bits) return f3d::fail("f3d_selftest_weights: empty range");
  WeightTally* d = nullptr;
  F3D_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof(WeightTally)));
  F3D_HIP(hipMemsetAsync(d, 0, sizeof(WeightTally), f3d::stream()));
  hipLaunchKernelGGL(k_selftest_weights, dim3(256 * 16), dim3(256), 0, f3d::stream(), lo_bits, hi_bits, d);
  WeightTally h;
  F3D_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, f3d::stream()));
  F3D_HIP(hipStreamSynchronize(f3d::stream()));
  F3D_HIP(hipFree(d));
  if (checked) *checked = h.checked;
  if (excluded) *excluded = h.excluded;
  if (mismatches) *mismatches = h.mismatches;
  if (first_mismatch) *first_mismatch = h.mismatches ? h.first[0] : 0u;
  return 0;
}

int f3d_frame_derivatives(f3d_devptr frame_0, f3d_devptr frame_1, size_t width, size_t height, size_t depth, float hx, float hy,
                          float hz, f3d_devptr fx, f3d_devptr fy, f3d_devptr fz, f3d_devptr ft, const f3d_slab* slab)
{
  F3D_REQUIRE_READY("f3d_frame_derivatives");
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, "f3d_frame_derivatives")) return 1;
  if (g.W < 2 || g.H < 2 || g.D < 2) return f3d::fail("f3d_frame_derivatives: every dimension must be at least 2");
  for (f3d_devptr o : {fx, fy, fz, ft})
    if (o == frame_0 || o == frame_1) return f3d::fail("f3d_frame_derivatives: an output aliases a frame");
  if (g.z_lo == g.z_hi) return 0;
  if (!slab_reach_ok(g, 1, "f3d_frame_derivatives")) return 1;
  const dim3 grid((g.W + 63) / 64, (g.H + 3) / 4, g.z_hi - g.z_lo), block(64, 4, 1);
  hipLaunchKernelGGL(k_frame_derivatives, grid, block, 0, f3d::stream(), f3d_ptr<const float>(frame_0), f3d_ptr<const float>(frame_1), g, hx,
                     hy, hz, f3d_ptr<float>(fx), f3d_ptr<float>(fy), f3d_ptr<float>(fz), f3d_ptr<float>(ft));
  F3D_HIP(hipGetLastError());
  return 0;
}

namespace {
// the two fused launches on precomputed frame derivatives
int pair8_fd(const char* who, bool with_weights, const f3d_devptr (&in)[12], size_t width, size_t height, size_t depth, float hx, float hy,
             float hz, float alpha, float eps_s, float eps_d, const f3d_devptr (&out)[5], const f3d_slab* slab, int keep_below = 0,
             int keep_above = 0)
{
  F3dGeo g;
  if (!f3d::make_geo(&g, width, height, depth, slab, who)) return 1;
  if (g.W < 2 || g.H < 2 || g.D < 2) return f3d::fail("%s: every dimension must be at least 2", who);
  if (!pair_weights_finite(hx, hy, hz, alpha)) return f3d::fail("%s: alpha / h^2 must be finite and not negative (alpha %g, h %g %g %g)", who, alpha, hx, hy, hz);
  if (g.pitch % kLanes != 0) return f3d::fail("%s: the container pitch must be a multiple of 256 bytes (f3d_alloc_pitched gives that)", who);
  if (with_weights && (out[3] == in[8] || out[4] == in[9] || out[3] == in[9] || out[4] == in[8]))
    return f3d::fail("%s: phi_next / ksi_next must not alias phi / ksi (other tiles still read them)", who);
  if (g.z_lo == g.z_hi) return 0;
  if (!slab_reach_ok(g, 2, who)) return 1;
  PairArgs a = {};
  a.keep_below = with_weights && keep_below != 0;
  a.keep_above = with_weights && keep_above != 0;
  static const int plain_division = std::getenv("F3D_UDIV") && std::atoi(std::getenv("F3D_UDIV")) == 0;
  a.plain_division = plain_division;
  for (int i = 0; i < 12; ++i) a.in[i] = f3d_ptr<const float>(in[i]);
  for (int i = 0; i < 5; ++i) a.out[i] = f3d_ptr<float>(out[i]);
  a.hx = hx; a.hy = hy; a.hz = hz;
  a.alpha = alpha;
  a.eps_s = eps_s;
  a.eps_d = eps_d;
  const int kid = with_weights ? F3D_K_SWEEP_PHI_KSI : F3D_K_SWEEP2;
  f3d::prof_begin(kid, static_cast<size_t>(g.W) * g.H * (g.z_hi - g.z_lo));
  // the tile height the round model picks for the level, as for the frame builds (the centre-only inputs have a two-slot ring of their
  // own, which is what makes twelve inputs fit the LDS of a CU with 16 waves)
  const int rows = pair8_rows(g);
  if (with_weights) {
    if (rows == 12) launch_pair8<PAIR_SP, 12, true>(a, g, tuning().zchunk, tuning().xcd_remap);
    else if (rows == 4) launch_pair8<PAIR_SP, 4, true>(a, g, tuning().zchunk, tuning().xcd_remap);
    else launch_pair8<PAIR_SP, 8, true>(a, g, tuning().zchunk, tuning().xcd_remap);
  } else {
    if (rows == 12) launch_pair8<PAIR_SS, 12, true>(a, g, tuning().zchunk, tuning().xcd_remap);
    else if (rows == 4) launch_pair8<PAIR_SS, 4, true>(a, g, tuning().zchunk, tuning().xcd_remap);
    else launch_pair8<PAIR_SS, 8, true>(a, g, tuning().zchunk, tuning().xcd_remap);
  }
  f3d::prof_end(kid);
  F3D_HIP(hipGetLastError());
  return 0;
}
}  // namespace

int f3d_solve_sweep2_fd(f3d_devptr fx, f3d_devptr fy, f3d_devptr fz, f3d_devptr ft, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                        f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi, size_t width,
                        size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha, f3d_devptr temp_du,
                        f3d_devptr temp_dv, f3d_devptr temp_dw, const f3d_slab* slab)
{
  F3D_REQUIRE_READY("f3d_solve_sweep2_fd");
  const f3d_devptr in[12] = {fx, fy, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw, phi, ksi, fz, ft};
  const f3d_devptr out[5] = {temp_du, temp_dv, temp_dw, 0, 0};
  return pair8_fd("f3d_solve_sweep2_fd", false, in, width, height, depth, hx, hy, hz, equation_alpha, 0.f, 0.f, out, slab);
}

int f3d_solve_sweep_phi_ksi_fd(f3d_devptr fx, f3d_devptr fy, f3d_devptr fz, f3d_devptr ft, f3d_devptr flow_u, f3d_devptr flow_v,
                               f3d_devptr flow_w, f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi,
                               f3d_devptr ksi, size_t width, size_t height, size_t depth, float hx, float hy, float hz,
                               float equation_alpha, float equation_smoothness, float equation_data, f3d_devptr temp_du,
                               f3d_devptr temp_dv, f3d_devptr temp_dw, f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab)
{
  F3D_REQUIRE_READY("f3d_solve_sweep_phi_ksi_fd");
  const f3d_devptr in[12] = {fx, fy, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw, phi, ksi, fz, ft};
  const f3d_devptr out[5] = {temp_du, temp_dv, temp_dw, phi_next, ksi_next};
  return pair8_fd("f3d_solve_sweep_phi_ksi_fd", true, in, width, height, depth, hx, hy, hz, equation_alpha, equation_smoothness,
                  equation_data, out, slab);
}

int f3d_solve_sweep_phi_ksi_edges_fd(f3d_devptr fx, f3d_devptr fy, f3d_devptr fz, f3d_devptr ft, f3d_devptr flow_u, f3d_devptr flow_v,
                                     f3d_devptr flow_w, f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi,
                                     f3d_devptr ksi, size_t width, size_t height, size_t depth, float hx, float hy, float hz,
                                     float equation_alpha, float equation_smoothness, float equation_data, f3d_devptr temp_du,
                                     f3d_devptr temp_dv, f3d_devptr temp_dw, f3d_devptr phi_next, f3d_devptr ksi_next, const f3d_slab* slab,
                                     int keep_below, int keep_above)
{
  F3D_REQUIRE_READY("f3d_solve_sweep_phi_ksi_edges_fd");
  const f3d_devptr in[12] = {fx, fy, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw, phi, ksi, fz, ft};
  const f3d_devptr out[5] = {temp_du, temp_dv, temp_dw, phi_next, ksi_next};
  return pair8_fd("f3d_solve_sweep_phi_ksi_edges_fd", true, in, width, height, depth, hx, hy, hz, equation_alpha, equation_smoothness,
                  equation_data, out, slab, keep_below, keep_above);
}

int f3d_fused_launches_march_along_y(size_t width, size_t height, size_t depth)
{
  // (no device needed: the launchers' own rule on the level's geometry, for a caller that has to choose between the entry points on
  // frames -- which march thin volumes along y -- and the ones on frame derivatives, which march along z only)
  F3dGeo g = {};
  const f3d_size4& c = f3d::container();
  g.W = static_cast<int>(width); g.H = static_cast<int>(height); g.D = static_cast<int>(depth);
  g.Hc = static_cast<int>(c.height); g.pitch = static_cast<int>(c.pitch / sizeof(float));
  g.z_base = 0; g.z_lo = 0; g.z_hi = g.D;
  if (g.pitch <= 0 || g.pitch % kLanes != 0 || !pair8_enabled()) return 0;
  return pair8_ymarch_rows(g) != 0 ? 1 : 0;
}

int f3d_solve_sweep3(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                     f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi, size_t width,
                     size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha, f3d_devptr temp_du,
                     f3d_devptr temp_dv, f3d_devptr temp_dw, const f3d_slab* slab)
{
  F3D_REQUIRE_READY("f3d_solve_sweep3");
  const f3d_devptr in[10] = {frame_0, frame_1, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw, phi, ksi};
  const f3d_devptr out[5] = {temp_du, temp_dv, temp_dw, 0, 0};
  return tri_launch("f3d_solve_sweep3", false, in, width, height, depth, hx, hy, hz, equation_alpha, 0.f, 0.f, out, slab);
}

int f3d_solve_sweep2_phi_ksi(f3d_devptr frame_0, f3d_devptr frame_1, f3d_devptr flow_u, f3d_devptr flow_v, f3d_devptr flow_w,
                             f3d_devptr flow_du, f3d_devptr flow_dv, f3d_devptr flow_dw, f3d_devptr phi, f3d_devptr ksi, size_t width,
                             size_t height, size_t depth, float hx, float hy, float hz, float equation_alpha, float equation_smoothness,
                             float equation_data, f3d_devptr temp_du, f3d_devptr temp_dv, f3d_devptr temp_dw, f3d_devptr phi_next,
                             f3d_devptr ksi_next, const f3d_slab* slab)
{
  F3D_REQUIRE_READY("f3d_solve_sweep2_phi_ksi");
  const f3d_devptr in[10] = {frame_0, frame_1, flow_u, flow_v, flow_w, flow_du, flow_dv, flow_dw, phi, ksi};
  const f3d_devptr out[5] = {temp_du, temp_dv, temp_dw, phi_next, ksi_next};
  return tri_launch("f3d_solve_sweep2_phi_ksi", true, in, width, height, depth, hx, hy, hz, equation_alpha, equation_smoothness,
                    equation_data, out, slab);
}

}  // extern "C"
