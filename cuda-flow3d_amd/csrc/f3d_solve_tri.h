// Three solver stages per launch for small and mid-size levels: k_tri.
//
// Part of f3d_solve.hip (included inside its anonymous namespace after f3d_solve_pair8.h): it uses the per-voxel arithmetic defined
// there (sweep_stage1 / sweep_stage2 / phi_ksi_stage2, PlaneRegs, Face6, Carry, CarryP, the uniform-divisor helpers) and the loader
// idiom of k_pair8 (dma16, uniform_ptr, counted s_waitcnt).
//
// Why it was built, and what it measured (LABBOOK.md, round 4).  Below ~128^3 a fused launch is not bound by bytes or by issue slots
// but by its own skeleton: ~4 us of dispatch + prologue and a handful of dependent plane steps.  A lab build of k_pair8 with a THIRD
// stage in every step (profiles/r04_three_stage_probe.txt) cost 16-23 % more than the two-stage launch, which argued for cutting an
// outer iteration as (S, S, S) + (S, S, P) -- two launches -- instead of (S, S) + (S, S) + (S, P).  This kernel is that cut, bit for
// bit; measured (profiles/r04_three_stage_kbench.txt, r04_three_stage_solves.txt) it is NOT faster: a three-stage z-chunk marches its
// planes + 4 steps where a two-stage chunk marches planes + 2 (the probe had added one), and the small levels are cut into one-plane
// chunks -- five steps against three -- so two launches of this kernel cost what three of k_pair8 cost (24^3: 29.6 against 29.7 us per
// outer iteration), and from ~96^3 up its tile shape loses outright.  It is therefore OPT-IN (F3D_TRI=1, host/hip_utils.cpp), kept
// tested and ISA-checked for a machine whose launches cost more.  The reference's loop: cuda_operation_solve.cpp:194-266 (phi/ksi,
// then `inner` sweeps with a buffer swap after each).
//
//   TRI_SSS  three consecutive sweeps                      -- f3d_solve_sweep3
//   TRI_SSP  two sweeps, then phi/ksi of the NEXT outer iteration from the increments they leave   -- f3d_solve_sweep2_phi_ksi
//
// Both keep the reference's expression trees operation for operation; what the stages of a voxel share
// are values the reference computes again from the same operands.
//
// Shape.  The skeleton is k_pair8's -- a loader wave feeds a three-slot LDS ring of raw planes by DMA, row waves march along z, a
// stage-k result reaches the row neighbours through an LDS image, the lane neighbours through DPP and the plane neighbours through
// registers -- with two differences that make a third stage affordable to write and to run:
//   * NO column wave and no x-halo pieces.  A tile is 64 lanes wide but owns only 56 columns: tile column t > 0 starts four columns
//     early (x = 56 t - 4, 16-byte aligned for the DMA pieces), so the three lanes either side of the owned range that a third stage
//     reaches into are computed by the tile itself, redundantly.  A lane's x neighbours are always the lanes beside it.  On levels
//     this size the vector unit has slots to spare; a column wave two columns deep with its own two-stage pipeline does not come free.
//   * TY + 4 row waves (rows y0-2 .. y0+TY+1): stage 1 on all of them, stage 2 on rows y0-1 .. y0+TY, stage 3 on the TY owned rows.
// Mirror rule: rows and planes outside the volume are fetched mirrored by address (stage 1 sees the reference's operands); stage-1 and
// stage-2 RESULTS of voxels outside the volume are never looked at -- at a face the missing neighbour is the opposite one (index
// -1 -> 1, n -> n-2), substituted where it is used, exactly as k_pair8 does.
//
// Pipeline of a row wave at step q (planes: M = q-1, C = q, P = q+1 raw, finished to faces):
//     stage 1 on plane q     from C, its lane / row / plane neighbours (raw)            -> s1 (S = U + dU'), carry k, carry p
//     stage 2 on plane q-1   from the s1 of plane q-1: rows via img1, lanes via DPP, planes q-2 (h1M) and q (s1 just made)  -> s2
//     stage 3 on plane q-2   from the s2 of plane q-2: rows via img2, lanes via DPP, planes q-3 (h2M) and q-1 (s2 just made) -> out
// A chunk [z0, z1) therefore runs q from z0-2 to z1+1 behind a three-plane prologue.

enum { TRI_SSS = 0, TRI_SSP = 1 };
constexpr int kTriStride = 56;  // owned columns per tile column (64 lanes: 4 + 56 + 4)

template <int TY>
struct TriLds {
  static constexpr int NR = TY + 4;   // row waves: rows y0-2 .. y0+TY+1
  static constexpr int NJ = TY + 6;   // ring rows: y0-3 .. y0+TY+2
  static constexpr int NK = (NJ + 3) / 4;
  static constexpr int NS = 10;       // f0, f1(warped), u, v, w, du, dv, dw, phi, ksi
  static constexpr int kSlotFloats = NS * NJ * kLanes;
  static constexpr int kSlots = 3;
  static constexpr int kPerPlane = NS * NK;   // DMA instructions per plane: the count the steady-state wait carries
  static_assert(kPerPlane <= 63, "s_waitcnt vmcnt(kPerPlane): the counter has six bits");
};

// tile columns of a level: one tile when both x faces fit its 64 lanes, else owned ranges of 56
inline int tri_tile_columns(int width) { return width <= kLanes ? 1 : (width + kTriStride - 1) / kTriStride; }

// phi_ksi_stage2 (f3d_solve_pair8.h) in its two halves -- the same operations in the same order -- because k_tri forms the two weights
// of a voxel at different steps: ksi depends on the voxel's own new increments only, so it is made (and stored) right behind the second
// sweep, and only the nine differences U[+1] - U[-1] travel on to the step that has the neighbours' increments for phi.  One guarded
// road to the weight each (weight_fast / weight_ieee give the same float wherever the guard lets the short road through).
__device__ __forceinline__ float tri_weight(float arg, bool counts, bool fast_weights)
{
  float root;
  const float w = weight_fast(arg, root);
  const bool lane_ok = !counts || weight_fast_ok(arg, root);
  if (__builtin_expect(fast_weights && __builtin_amdgcn_ballot_w64(!lane_ok) == 0, 1)) return w;
  return weight_ieee(arg);
}
__device__ __forceinline__ float tri_ksi(float fx, float fy, float fz, float ft, float du, float dv_c, float dw, float eps_d2, bool counts,
                                         bool fast_weights)
{
  const float J11 = fx * fx, J22 = fy * fy, J33 = fz * fz;
  const float J12 = fx * fy, J13 = fx * fz, J23 = fy * fz;
  const float J14 = fx * ft, J24 = fy * ft, J34 = fz * ft, J44 = ft * ft;
  float s = (J11 * du + J12 * dv_c + J13 * dw + J14) * du + (J12 * du + J22 * dv_c + J23 * dw + J24) * dv_c +
            (J13 * du + J23 * dv_c + J33 * dw + J34) * dw + (J14 * du + J24 * dv_c + J34 * dw + J44);
  s = static_cast<float>(s > 0) * s;
  return tri_weight(s + eps_d2, counts, fast_weights);
}
__device__ __forceinline__ float tri_phi(const float (&D)[9], const S3& xm, const S3& xp, const S3& ym, const S3& yp, const S3& zm,
                                         const S3& zp, const SolveDivs& dv, float eps_s2, bool counts, bool fast_weights)
{
  float q[9] = {D[0] + xp.u - xm.u, D[1] + yp.u - ym.u, D[2] + zp.u - zm.u, D[3] + xp.v - xm.v, D[4] + yp.v - ym.v,
                D[5] + zp.v - zm.v, D[6] + xp.w - xm.w, D[7] + yp.w - ym.w, D[8] + zp.w - zm.w};
  if (__builtin_expect(dv.ok && udiv_all_safe(q), 1)) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      q[3 * c + 0] = udiv(q[3 * c + 0], dv.x2);
      q[3 * c + 1] = udiv(q[3 * c + 1], dv.y2);
      q[3 * c + 2] = udiv(q[3 * c + 2], dv.z2);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      q[3 * c + 0] = q[3 * c + 0] / dv.x2.d;
      q[3 * c + 1] = q[3 * c + 1] / dv.y2.d;
      q[3 * c + 2] = q[3 * c + 2] / dv.z2.d;
    }
  }
  const float dux = q[0], duy = q[1], duz = q[2], dvx = q[3], dvy = q[4], dvz = q[5], dwx = q[6], dwy = q[7], dwz = q[8];
  const float a_phi = dux * dux + duy * duy + duz * duz + dvx * dvx + dvy * dvy + dvz * dvz + dwx * dwx + dwy * dwy + dwz * dwz + eps_s2;
  return tri_weight(a_phi, counts, fast_weights);
}

template <int MODE, int TY>
__global__ __launch_bounds__(kLanes*(TY + 5)) void k_tri(PairArgs a, F3dGeo g, int zchunk, int ntx, int nty, int n_tiles, int xcd_remap)
{
  using L = TriLds<TY>;
  constexpr int NR = L::NR, NJ = L::NJ, NK = L::NK, NS = L::NS;
  __shared__ __attribute__((aligned(16))) float ring[L::kSlots][L::kSlotFloats];
  __shared__ float img1[2][3][NR][kLanes];  // stage-1 results S = U + dU' of the row waves, by plane parity
  __shared__ float img2[2][3][NR][kLanes];  // stage-2 results: S = U + dU'' (SSS) or dU'' (SSP)

  int tile = static_cast<int>(blockIdx.x);
  if (xcd_remap) {
    const int per_xcd = (n_tiles + 7) / 8;
    tile = (tile % 8) * per_xcd + tile / 8;
  }
  if (tile >= n_tiles) return;
  const int tx = tile % ntx;
  const int ty = (tile / ntx) % nty;
  const int tz = tile / (ntx * nty);

  int lane = threadIdx.x;   // (SSP re-makes it at the top of every step: see fresh_lane in f3d_solve_pair8.h)
  const int r = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  const bool loader = r == NR;
  const int D = g.D;
  const int z0 = g.z_lo + tz * zchunk;
  const int z1 = min(z0 + zchunk, g.z_hi);
  const int qs = z0 - 2 > 0 ? z0 - 2 : 0;          // first plane of stage 1
  const int qe = z1 + 1 < D ? z1 + 1 : D - 1;      // last plane of stage 1
  const int q_end = z1 + 1;                        // last step: stage 3 of plane z1-1
  const int p_last = qe + 1;                       // last plane the ring ever holds (mirrored when it is D)
  const int xs = tx == 0 ? 0 : tx * kTriStride - 4;  // column of lane 0
  const int u_lo = tx == 0 ? 0 : tx * kTriStride;    // owned columns [u_lo, u_hi)
  const int u_hi = ntx == 1 ? g.W : min(g.W, tx * kTriStride + kTriStride);
  const int y0 = ty * TY;
  const bool tile_at_x_face = __builtin_amdgcn_readfirstlane(static_cast<int>(xs == 0 || xs + kLanes >= g.W)) != 0;

  const int zb = qs > 0 ? qs - 1 : 0;  // lowest plane touched: byte offsets inside the chunk stay small and positive
  const unsigned plane_b = static_cast<unsigned>(g.Hc) * static_cast<unsigned>(g.pitch) * 4u;
  const unsigned row_b = static_cast<unsigned>(g.pitch) * 4u;
  const size_t base_off = f3d_row(g, 0, zb);

  // ================================================== loader wave ==================================================
  if (loader) {
    __builtin_amdgcn_s_setprio(3);
    const float* base[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) base[i] = uniform_ptr(a.in[i] + base_off);
    // row pieces: lane -> (ring row 4k + lane/16, floats 4 (lane%16) ..).  The last tile column reaches beyond the row: its pieces
    // are fetched from the end of the row instead (lanes beyond the volume are never looked at), so no address leaves the container
    int xpiece = xs + 4 * (lane & 15);
    if (xpiece + 4 > g.pitch) xpiece = g.pitch - 4;
    unsigned rowb[NK];
    bool rowv[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int j = 4 * k + (lane >> 4);
      rowv[k] = j < NJ;
      const int yrow = f3d_clampi(f3d_mir(y0 - 3 + (rowv[k] ? j : 0), g.H), 0, g.H - 1);
      rowb[k] = static_cast<unsigned>(yrow) * row_b + static_cast<unsigned>(xpiece) * 4u;
    }
    auto issue = [&](int p) {  // plane p (mirrored for the address) into slot (p - (qs-1)) mod 3
      float* slot = &ring[(p - qs + 1) % L::kSlots][0];
      const int zz = f3d_mir(p, D);
      const unsigned poff = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(zz - zb) * plane_b)));
      unsigned off[NK];
#pragma unroll
      for (int k = 0; k < NK; ++k) off[k] = rowb[k] + poff;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          if (NJ % 4 == 0 || rowv[k]) dma16(base[i], off[k], slot + (i * NJ + 4 * k) * kLanes);
        }
      }
    };
    // prologue: planes qs-1, qs, qs+1 fill the three slots and must have landed before anybody reads
    issue(qs - 1);
    issue(qs);
    issue(qs + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int q = qs; q <= q_end; ++q) {
      __syncthreads();  // B_q: the slots of planes q-1 and q have been read for the last time
      if (q == qs && q + 2 <= p_last) issue(q + 2);  // steady state: issued one step ago
      if (q + 3 <= p_last) {
        issue(q + 3);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::kPerPlane) : "memory");  // plane q+2 has landed, q+3 stays in flight
        continue;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    return;
  }

  // ================================================= row waves =================================================
  FDivs fdivs;
  fdivs.x4 = UDiv{a.r4[0], a.d4[0]}; fdivs.y4 = UDiv{a.r4[1], a.d4[1]}; fdivs.z4 = UDiv{a.r4[2], a.d4[2]};
  fdivs.ok = a.fdivs_ok != 0;
  SolveDivs sdivs = {};
  if (MODE == TRI_SSP) {
    sdivs.x2 = UDiv{a.r2[0], a.d2[0]}; sdivs.y2 = UDiv{a.r2[1], a.d2[1]}; sdivs.z2 = UDiv{a.r2[2], a.d2[2]};
    sdivs.x4 = fdivs.x4; sdivs.y4 = fdivs.y4; sdivs.z4 = fdivs.z4;
    sdivs.ok = a.sdivs_ok != 0;
  }

  const int y = y0 - 2 + r;
  const int yy = f3d_clampi(f3d_mir(y, g.H), 0, g.H - 1);
  const bool wave2 = r >= 1 && r <= NR - 2;   // rows y0-1 .. y0+TY: stage 2
  const bool core = r >= 2 && r <= NR - 3;    // rows y0 .. y0+TY-1: stage 3 and the stores
  const bool owner_row = core && y < g.H;
  const int jr = r + 1;  // ring row of this wave's row
  // image rows of the row neighbours; at a y face of the volume the missing neighbour is the opposite row (mirror rule).  Waves
  // that run no later stage never look at an image: their indices are only kept inside the arrays.
  const int r_ym = f3d_clampi(y == 0 ? r + 1 : r - 1, 0, NR - 1);
  const int r_yp = f3d_clampi(y == g.H - 1 ? r - 1 : r + 1, 0, NR - 1);

  float* obase[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) obase[i] = (i < 3 || MODE == TRI_SSP) ? uniform_ptr(a.out[i] + base_off) : nullptr;
  auto rowoff = [&](int yrow, int zz) __attribute__((always_inline)) {
    return static_cast<unsigned>(__builtin_amdgcn_readfirstlane(
        static_cast<int>(static_cast<unsigned>(zz - zb) * plane_b + static_cast<unsigned>(yrow) * row_b)));
  };

  // raw values of one ring row (own lane)
  auto row_raw = [&](PlaneRegs& p, const float* slot, int j, bool with_ksi) __attribute__((always_inline)) {
    const float* d = slot + j * kLanes + lane;
    constexpr int st = NJ * kLanes;
    p.f0 = d[F0 * st]; p.f1 = d[F1 * st];
    p.u = d[U * st]; p.v = d[V * st]; p.w = d[Wf * st];
    p.su = d[DU * st]; p.dv = d[DV * st]; p.dw = d[DW * st]; p.phi = d[PHI * st];
    if (with_ksi) p.ksi = d[9 * st];
  };

  // Raw row neighbours of a plane for stage 1, fetched at the END of the step before the one that uses them (the plane has been in
  // the ring since the barrier before last) and finished to faces right there, as in k_pair8
  Face6 nYm = {}, nYp = {};
  S3 nDy = {0.f, 0.f, 0.f};  // SSP: U[y+1] - U[y-1], the first operation of phi/ksi's y derivatives
  auto fetch_neighbours = [&](const float* S) __attribute__((always_inline)) {
    PlaneRegs T0, T1;
    row_raw(T0, S, jr - 1, false);
    row_raw(T1, S, jr + 1, false);
    if (MODE == TRI_SSP) nDy = {T1.u - T0.u, T1.v - T0.v, T1.w - T0.w};
    plane_finish(T0);
    plane_finish(T1);
    nYm = plane_face(T0);
    nYp = plane_face(T1);
  };

  // what travels from step to step
  S3 h1M = {0.f, 0.f, 0.f}, h1C = {0.f, 0.f, 0.f};   // stage-1 results S of planes q-2, q-1
  float h1C_dv = 0.f, h1C_dw = 0.f;                   // dV', dW' of plane q-1
  S3 h2M = {0.f, 0.f, 0.f}, h2C = {0.f, 0.f, 0.f};   // stage-2 results of planes q-3, q-2 (SSS: S; SSP: dU'')
  float h2C_dv = 0.f, h2C_dw = 0.f;                   // SSS: dV'', dW'' of plane q-2
  Carry k1 = {}, k2 = {};                             // stage-1 carries of planes q-1, q-2
  CarryP p1 = {};                                     // SSP: fx .. ft and the nine U[+1] - U[-1] of plane q-1
  float d2[9] = {};                                   // SSP: the nine differences of plane q-2 (all that phi still needs)

  auto step = [&](auto slot_c, PlaneRegs& M, PlaneRegs& C, PlaneRegs& P, int q) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_c)::value;   // ring slot of plane q+1: a compile-time constant (march unrolled three deep)
    if constexpr (MODE == TRI_SSP) lane = fresh_lane();
    // per-lane values that do not change from step to step are formed here rather than kept: registers are short (168 with 12 waves)
    const int x = xs + lane;
    const unsigned xb = static_cast<unsigned>(x) * 4u;
    const bool owner = owner_row && x >= u_lo && x < u_hi;
    __syncthreads();  // B_q: plane q+1 is in the ring; img1 of plane q-1 and img2 of plane q-2 are complete
    const float* Sp = &ring[SLOT][0];
    const int b = q & 1;

    // ---- stage 1 on plane q ----
    float r_du = 0.f, r_dv = 0.f, r_dw = 0.f;
    Carry kN;
    CarryP pN;
    {
      row_raw(P, Sp, jr, true);
      plane_finish(P);
      const Face6 cf = plane_face(C);
      Face6 xm, xp;
#pragma unroll
      for (int i = 0; i < kNL; ++i) {
        xm.v[i] = lane_left_or(cf.v[i], cf.v[i]);
        xp.v[i] = lane_right_or(cf.v[i], cf.v[i]);
      }
      S3 rxm = {}, rxp = {};
      if (MODE == TRI_SSP) {
        rxm = {lane_left_or(C.u, C.u), lane_left_or(C.v, C.v), lane_left_or(C.w, C.w)};
        rxp = {lane_right_or(C.u, C.u), lane_right_or(C.v, C.v), lane_right_or(C.w, C.w)};
      }
      // mirror rule at the x faces of the volume (index -1 -> 1, W -> W-2): the missing neighbour is the opposite one
      if (tile_at_x_face) {
        if (x == 0) {
          xm = xp;
          rxm = rxp;
        }
        if (x == g.W - 1) {
          xp = xm;
          rxp = rxm;
        }
      }
      const Face6 fM = plane_face(M), fP = plane_face(P);
      sweep_stage1<false, true>(xm, xp, nYm, nYp, fM, fP, cf.v, C.u, C.v, C.w, C.dv, C.dw, C.ksi, a.hx, a.hy, a.hz, fdivs, a.alpha,
                                x < g.W - 1, x > 0, y < g.H - 1, y > 0, q < D - 1, q > 0, r_du, r_dv, r_dw, kN, C.f0, C.f1, C.phi, C.ksi,
                                tile_at_x_face, a.w[0], a.w[1], a.w[2]);
      pN.fx = kN.fx; pN.fy = kN.fy; pN.fz = kN.fz; pN.ft = kN.ft;
      pN.D[0] = rxp.u - rxm.u; pN.D[1] = nDy.u; pN.D[2] = P.u - M.u;
      pN.D[3] = rxp.v - rxm.v; pN.D[4] = nDy.v; pN.D[5] = P.v - M.v;
      pN.D[6] = rxp.w - rxm.w; pN.D[7] = nDy.w; pN.D[8] = P.w - M.w;
    }
    const S3 s1N = {C.u + r_du, C.v + r_dv, C.w + r_dw};  // what a neighbour reads of this voxel in stage 2
    img1[b][0][r][lane] = s1N.u;
    img1[b][1][r][lane] = s1N.v;
    img1[b][2][r][lane] = s1N.w;

    // ---- stage 2 (a sweep) on plane t2 = q - 1 ----
    const int t2 = q - 1;
    const bool do2 = wave2 && t2 >= 0 && t2 <= D - 1 && t2 >= z0 - 1 && t2 <= z1;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    S3 s2N = {0.f, 0.f, 0.f};
    if (do2) {
      const int pb = t2 & 1;
      S3 ym, yp, xm, xp, zm, zp;
      ym.u = img1[pb][0][r_ym][lane]; ym.v = img1[pb][1][r_ym][lane]; ym.w = img1[pb][2][r_ym][lane];
      yp.u = img1[pb][0][r_yp][lane]; yp.v = img1[pb][1][r_yp][lane]; yp.w = img1[pb][2][r_yp][lane];
      xm.u = lane_left_or(h1C.u, h1C.u); xm.v = lane_left_or(h1C.v, h1C.v); xm.w = lane_left_or(h1C.w, h1C.w);
      xp.u = lane_right_or(h1C.u, h1C.u); xp.v = lane_right_or(h1C.v, h1C.v); xp.w = lane_right_or(h1C.w, h1C.w);
      zm = h1M;
      zp = s1N;
      if (tile_at_x_face) {
        if (x == 0) xm = xp;
        if (x == g.W - 1) xp = xm;
      }
      if (t2 == 0) zm = zp;
      if (t2 == D - 1) zp = zm;
      Carry kk = k1;
      if (MODE == TRI_SSP) {
        // register diet: the six J products of the voxel are not carried beside fx .. ft (which phi/ksi needs anyway) but formed
        // again from them -- the same six multiplications, the same bits
        kk.J12 = p1.fx * p1.fy; kk.J13 = p1.fx * p1.fz; kk.J23 = p1.fy * p1.fz;
        kk.J14 = p1.fx * p1.ft; kk.J24 = p1.fy * p1.ft; kk.J34 = p1.fz * p1.ft;
      }
      sweep_stage2(kk, xm, xp, ym, yp, zm, zp, h1C_dv, h1C_dw, o0, o1, o2);
      s2N = MODE == TRI_SSS ? S3{k1.U + o0, k1.V + o1, k1.W + o2} : S3{o0, o1, o2};
      img2[pb][0][r][lane] = s2N.u;
      img2[pb][1][r][lane] = s2N.v;
      img2[pb][2][r][lane] = s2N.w;
      // SSP: the second sweep's result is final, and ksi of the next outer iteration depends on this voxel's own increments only:
      // both are stored here for the planes this chunk owns
      if (MODE == TRI_SSP && core && t2 >= z0 && t2 < z1) {
        const float ksi_next = tri_ksi(p1.fx, p1.fy, p1.fz, p1.ft, o0, o1, o2, a.eps_d2, owner, a.plain_division == 0);
        if (owner) {
          const unsigned off = xb + rowoff(yy, t2);
          gst(obase[0], off, o0);
          gst(obase[1], off, o1);
          gst(obase[2], off, o2);
          gst(obase[4], off, ksi_next);
        }
      }
    }

    // ---- stage 3 on plane t3 = q - 2: the third sweep (SSS) or phi/ksi of the next outer iteration (SSP) ----
    const int t3 = q - 2;
    const bool do3 = core && t3 >= z0 && t3 < z1;
    float e0 = 0.f, e1 = 0.f, e2 = 0.f;
    if (do3) {
      const int pb = t3 & 1;
      S3 ym, yp, xm, xp, zm, zp;
      ym.u = img2[pb][0][r_ym][lane]; ym.v = img2[pb][1][r_ym][lane]; ym.w = img2[pb][2][r_ym][lane];
      yp.u = img2[pb][0][r_yp][lane]; yp.v = img2[pb][1][r_yp][lane]; yp.w = img2[pb][2][r_yp][lane];
      xm.u = lane_left_or(h2C.u, h2C.u); xm.v = lane_left_or(h2C.v, h2C.v); xm.w = lane_left_or(h2C.w, h2C.w);
      xp.u = lane_right_or(h2C.u, h2C.u); xp.v = lane_right_or(h2C.v, h2C.v); xp.w = lane_right_or(h2C.w, h2C.w);
      zm = h2M;
      zp = s2N;
      if (tile_at_x_face) {
        if (x == 0) xm = xp;
        if (x == g.W - 1) xp = xm;
      }
      if (t3 == 0) zm = zp;
      if (t3 == D - 1) zp = zm;
      if (MODE == TRI_SSS)
        sweep_stage2(k2, xm, xp, ym, yp, zm, zp, h2C_dv, h2C_dw, e0, e1, e2);
      else
        e0 = tri_phi(d2, xm, xp, ym, yp, zm, zp, sdivs, a.eps_s2, owner, a.plain_division == 0);
    }
    asm volatile("" ::"v"(e0), "v"(e1), "v"(e2), "v"(s2N.u), "v"(s2N.v), "v"(s2N.w), "v"(s1N.u), "v"(s1N.v), "v"(s1N.w));
    h1M = h1C;
    h1C = s1N;
    h1C_dv = r_dv;
    h1C_dw = r_dw;
    h2M = h2C;
    h2C = s2N;
    h2C_dv = o1;
    h2C_dw = o2;
    k2 = k1;
    k1 = kN;
    if (MODE == TRI_SSP) k1.J12 = k1.J13 = k1.J23 = k1.J14 = k1.J24 = k1.J34 = 0.f;   // (re-made from p1 where stage 2 wants them)
#pragma unroll
    for (int i = 0; i < 9; ++i) d2[i] = p1.D[i];
    p1 = pN;
    if (do3 && owner) {
      const unsigned off = xb + rowoff(yy, t3);
      if (MODE == TRI_SSS) {
        gst(obase[0], off, e0);
        gst(obase[1], off, e1);
        gst(obase[2], off, e2);
      } else {
        gst(obase[3], off, e0);   // (ksi of this plane went out one step ago, behind its second sweep)
      }
    }
    fetch_neighbours(Sp);  // for step q+1
  };

  __syncthreads();  // prologue barrier: planes qs-1, qs, qs+1 are in the ring
  PlaneRegs A = {}, B = {}, Cc = {};
  row_raw(A, &ring[0][0], jr, false);   // plane qs-1
  row_raw(B, &ring[1][0], jr, true);    // plane qs
  plane_finish(A);
  plane_finish(B);
  fetch_neighbours(&ring[1][0]);
  using Slot0 = std::integral_constant<int, 0>;
  using Slot1 = std::integral_constant<int, 1>;
  using Slot2 = std::integral_constant<int, 2>;
  int q = qs;
  for (; q + 2 <= q_end; q += 3) {  // step q reads plane q+1 from slot (q - qs + 2) mod 3
    step(Slot2{}, A, B, Cc, q);
    step(Slot0{}, B, Cc, A, q + 1);
    step(Slot1{}, Cc, A, B, q + 2);
  }
  if (q <= q_end) step(Slot2{}, A, B, Cc, q);
  if (q + 1 <= q_end) step(Slot0{}, B, Cc, A, q + 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores issued by hand
}

// z-chunks by the round model of k_pair8 (a chunk costs its planes plus the prologue and the two trailing stages: ~9 steps)
inline Pair8Plan tri_plan_dims(int width, int rows, int planes, int ty, int zc_limit, long per_round = 256)
{
  const long tiles = static_cast<long>(tri_tile_columns(width)) * ((rows + ty - 1) / ty);
  const int max_chunks = planes > 0 ? planes : 1;
  Pair8Plan p = {std::min(planes, zc_limit), -1};
  for (int nzc = 1; nzc <= max_chunks; ++nzc) {
    const int zc = (planes + nzc - 1) / nzc;
    if (zc > zc_limit) continue;
    const long wgs = tiles * ((planes + zc - 1) / zc);
    const long cost = ((wgs + per_round - 1) / per_round) * (zc + 9);
    if (p.cost < 0 || cost < p.cost) {
      p.cost = cost;
      p.zchunk = zc;
      p.wgs = wgs;
    }
  }
  if (p.cost < 0) {
    p.cost = static_cast<long>((tiles + per_round - 1) / per_round) * (p.zchunk + 9);
    p.wgs = tiles;
  }
  return p;
}

template <int MODE, int TY>
void launch_tri(const PairArgs& args, const F3dGeo& g, int force_zchunk, int xcd_remap)
{
  PairArgs a = args;
  pair_consts(a);
  const int planes = g.z_hi - g.z_lo;
  const int ntx = tri_tile_columns(g.W);
  const int nty = (g.H + TY - 1) / TY;
  const int zc_limit = max_planes_per_chunk(g);
  int zchunk = tri_plan_dims(g.W, g.H, planes, TY, zc_limit).zchunk;
  if (force_zchunk > 0) zchunk = force_zchunk;
  zchunk = std::min(zchunk, zc_limit);
  const int nz = (planes + zchunk - 1) / zchunk;
  const int n_tiles = ntx * nty * nz;
  const int per_xcd = (n_tiles + 7) / 8;
  const int blocks = xcd_remap ? per_xcd * 8 : n_tiles;
  const dim3 grid(blocks, 1, 1), block(kLanes, TY + 5, 1);
  hipLaunchKernelGGL((k_tri<MODE, TY>), grid, block, 0, f3d::stream(), a, g, zchunk, ntx, nty, n_tiles, xcd_remap);
}
