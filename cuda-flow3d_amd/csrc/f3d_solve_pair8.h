// Two solver stages per launch, fed by a DMA loader wave: k_pair8.
//
// This header is part of f3d_solve.hip (included inside its anonymous namespace, after the per-voxel arithmetic and
// k_sweep7): it uses sweep_stage1 / sweep_stage2 / PlaneRegs / Face6 / the uniform-divisor helpers defined there.
//
// Why.  k_sweep7 (two fused sweeps, every row wave loads its own row with `global_load_dword`) moves the bytes of one
// sweep for the work of two but spends a sixth of every step in the vector-memory issue stage: a CU accepts ~128
// vector-memory instructions, a dword-per-lane load carries 256 B, so a CU never has more than ~32 KB on its way
// (profiles/r01_summary.md, tools/lab/issue_lab).  Here NO compute wave issues a load:
//
//   * one LOADER wave per workgroup fetches every input plane with `global_load_lds_dwordx4` -- 16 B per lane, 1 KiB per
//     instruction, four 64-float row segments at a time, straight into an LDS ring of three raw planes (34 instructions per
//     plane instead of ~150) -- and keeps TWO planes in flight behind a counted `s_waitcnt vmcnt`: everything a step reads of
//     plane p has been read when the barrier of step p falls, so three slots are enough for that;
//   * the row waves read their operands from the ring with ds_read: their own row of plane q+1 (kept in registers for the
//     three steps it serves as z+1, centre and z-1) after the barrier, the rows above and below and the x-halo column of the
//     next step's plane at the END of the step; nothing is re-published, so the 40 KB face image and the halo rings of
//     k_sweep7 are gone and the plane-in-flight registers with them;
//   * stage 2 works exactly like k_sweep7's: stage-1 results of the neighbours come from an LDS image (rows), DPP (lanes), the
//     column wave (tile edges) and registers (planes).
// What it buys and what bounds it now (with 12-row tiles at four waves per SIMD: the vector-issue rate, 8.6e8 instructions
// per 512^3 two-sweep launch x 4 cycles / 1024 SIMDs = the whole launch): DESIGN.md section 3.
//
// Two flavours of the second stage:
//   PAIR_SS  stage 1 = sweep, stage 2 = sweep              -- f3d_solve_sweep2 (two iterations of cuda_operation_solve.cpp:222-255)
//   PAIR_SP  stage 1 = sweep, stage 2 = phi/ksi of the NEXT outer iteration from the increments stage 1 just produced
//                                                          -- f3d_solve_sweep_phi_ksi (last solve_3d launch of outer iteration i
//                                                             and compute_phi_ksi_3d of iteration i+1, cuda_operation_solve.cpp:215-252)
// Both keep the reference's expression trees operation for operation; what is shared between
// the stages are values the reference computes twice from the same operands (fx, fy, fz, ft, the J products, U[x+1] - U[x-1]).
//
// Tile: TY core rows x 64 columns, marching along z.  Waves: TY + 2 row waves (rows y0-1 .. y0+TY), one column wave
// (stage 1 of the 2 x TY voxels left and right of the tile), one loader wave.  TY = 12 -> 16 waves, four per SIMD, 117 / 125
// VGPRs (SS / SP); TY = 8 -> 12 waves; TY = 4 -> 8 waves for levels so small that a workgroup has its CU to itself.  The
// launcher prices the three shapes per level (pair8_rows in f3d_solve.hip).  Nothing that is the same for every lane of a wave
// is computed by the vector unit: the reciprocals of 2h and 4h, alpha / h^2 and eps^2 arrive as kernel arguments (pair_consts),
// the x-face weights are applied by selection in tiles that touch a face, the lane number is re-made per step where registers
// are short -- formed per wave they cost 11-22 VGPRs of loop-invariant values and with them the fourth wave per SIMD.
//
// Thin volumes march along y (YM).  A volume of 4 or 5 planes -- BASELINE config 3, 584 x 388 x 5 -- leaves a z march nothing to
// march over: three prologue planes for four or five steps, and the tile rows cut the one long axis that is left.  With YM the
// roles of y and z are exchanged in the DATA MOVEMENT only: the "rows" of a tile are the volume's z planes (all of them: TY = 4, 5
// or 8), the march runs along y (388 steps), rows are Hc * pitch apart and march steps one pitch -- the loader, the ring, the
// images and the registers never know.  The ARITHMETIC keeps its geometry: where the stages are called the neighbours held as
// planes (M, P, hM, sN) are handed in as the y neighbours and the tile's row neighbours as the z neighbours, the y / z face flags
// and mirror rules swap with them, so every expression sees the operands the z-marching kernel (and the reference) gives it.
//
// x faces.  A 16-byte DMA piece cannot mirror inside itself, so lanes beyond the volume (x >= W) hold whatever the padded
// row holds and the reference's mirror rule is applied where it matters: at x = 0 the left neighbour IS the right one
// (index -1 -> 1) and at x = W-1 the right one is the left one, so stage 1 substitutes the whole neighbour there (rows and
// planes are mirrored by address, as before).  Halo pieces of tiles at an x face are fetched from columns inside the row
// (their values are never used) so that no address leaves the container.

enum { PAIR_SS = 0, PAIR_SP = 1 };

struct PairArgs {
  // f0, f1(warped), u, v, w, du, dv, dw, phi, ksi; on precomputed frame derivatives (FD): fx, fy in place of f0, f1 and
  // fz, ft as arrays 10, 11
  const float* in[12];
  float* out[5];        // temp_du, temp_dv, temp_dw; PAIR_SP: new phi, new ksi
  float hx, hy, hz, alpha;
  float eps_s, eps_d;   // PAIR_SP
  int plain_division;   // timing experiments (F3D_UDIV=0)
  // PAIR_SP: also store the sweep on plane z_lo-1 / z_hi of the window -- the launch computes them anyway for the weights of
  // planes z_lo and z_hi-1 (f3d_solve_sweep_phi_ksi_edges: a z-slab that owns one plane more than it can give weights to)
  int keep_below, keep_above;
  // Wave-uniform constants, made on the HOST by pair_consts() with the operations the kernels used to run per wave (float
  // products, IEEE float / double divisions: the same bits): as kernel arguments they sit in scalar registers or are
  // re-read from the argument segment by a scalar load when registers are short.  Formed on the device they were vector
  // values -- 10 + VGPRs of loop-invariant doubles that a 128-register build kept in scratch.
  double r4[3], r2[3];   // RN64(1 / (4 h)), RN64(1 / (2 h))
  float d4[3], d2[3];    // 4 h, 2 h
  float w[3];            // alpha / (h * h)
  float eps_s2, eps_d2;  // PAIR_SP: eps * eps
  int fdivs_ok, sdivs_ok;
};

// The fused kernels apply the face weights alpha / h^2 by selection: (float)(flag) * w is w or +0 only for a finite w that is not
// negative (0 * inf is NaN, 0 * -w is -0), so their entry points refuse spacings and alphas outside that -- documented in
// include/f3d.h; the host drivers take the one-sweep launches (k_sweep6: the reference's multiply) for such parameters
// (host/hip_utils.cpp: NoteSolveWeights), so that the operator API still returns what the reference would.
inline bool pair_weights_finite(float hx, float hy, float hz, float alpha)
{
  for (float h : {hx, hy, hz}) {
    const float w = alpha / (h * h);
    if (!(w - w == 0.f) || std::signbit(w)) return false;
  }
  return true;
}
inline bool host_divisor_ok(float d) { return d >= 0x1p-20f && d <= 0x1p20f; }  // udiv_divisor_ok
inline void pair_consts(PairArgs& a)
{
  const float h[3] = {a.hx, a.hy, a.hz};
  a.fdivs_ok = a.sdivs_ok = 1;
  for (int i = 0; i < 3; ++i) {
    a.d4[i] = 4.f * h[i];
    a.d2[i] = 2.f * h[i];
    a.r4[i] = 1.0 / static_cast<double>(a.d4[i]);
    a.r2[i] = 1.0 / static_cast<double>(a.d2[i]);
    a.w[i] = a.alpha / (h[i] * h[i]);
    a.fdivs_ok = a.fdivs_ok && host_divisor_ok(a.d4[i]);
    a.sdivs_ok = a.sdivs_ok && host_divisor_ok(a.d4[i]) && host_divisor_ok(a.d2[i]);
  }
  a.eps_s2 = a.eps_s * a.eps_s;
  a.eps_d2 = a.eps_d * a.eps_d;
  if (a.plain_division) a.fdivs_ok = a.sdivs_ok = 0;
}

// FD (frame derivatives read instead of formed): twelve inputs, of which only seven are STENCILLED (u, v, w, du, dv, dw, phi: a voxel
// reads its neighbours' values) -- fx, fy, fz, ft and ksi are looked at by their own voxel only.  The stencilled ones keep the ring
// described here; the five centre-only ones get a ring of their own (`cring`, C* below) that holds just the rows a stage 1 runs on
// (y0-1 .. y0+TY) and the halo columns of the core rows, and only TWO slots: a centre-only plane is read once, at the very start of
// the step in which it is the z+1 plane, so its slot is free one step earlier than a stencilled plane's -- which is what lets a
// 12-row tile with twelve inputs fit the 160 KB of a CU (3 x 32 KB + 2 x 19.5 KB + 22.5 KB of stage-1 images).
// TIGHT (y-marching builds whose tile holds ALL rows of the volume, i.e. every z plane of a thin volume): the two halo row waves and the
// four halo ring rows would only hold mirror images of rows the tile has anyway, so they are left out -- TY row waves, TY ring rows,
// the mirrored neighbour of a face row is read from the opposite row -- and two such workgroups share a CU.
template <int TY, int NA = 10, bool FD = false, bool TIGHT = false>
struct Pair8Lds {
  static constexpr int NR = TY + (TIGHT ? 0 : 2);     // row waves
  static constexpr int NJ = TY + (TIGHT ? 0 : 4);     // ring rows: y0-2 .. y0+TY+1 (TIGHT: y0 .. y0+TY-1)
  static constexpr int NK = (NJ + 3) / 4;             // row pieces (4 rows x 64 floats = 1 KiB) per array and plane
  static constexpr int NJP = NJ;                      // rows per array in the ring (a partial last piece masks its surplus lanes)
  static constexpr int NS = FD ? 7 : NA;              // arrays in the three-slot ring
  static constexpr int kHaloLanes = NS * 2 * NJ;      // 16-byte x-halo pieces per plane: [array][side][row]
  static constexpr int NH = (kHaloLanes + 63) / 64;   // halo instructions per plane
  static constexpr int kRowFloats = NS * NJP * 64;
  static constexpr int kHaloOff = kRowFloats;         // float offset of the halo area inside a slot
  static constexpr int kSlotFloats = kRowFloats + NH * 256;
  // Three slots hold TWO planes in flight: everything a step reads of plane p (its rows as z+1 plane during step p-1, its
  // neighbour rows and halo columns at the end of step p-1) has been read when barrier B_p falls, so at step q the loader
  // refills the slot of plane q with plane q+3 while q+1 is being read and q+2 is landing.
  static constexpr int kSlots = 3;
  // DMA instructions per plane (the counted wait leaves one plane in flight).  vmcnt is a 6-bit counter: with more than 31 pieces per
  // plane (TY = 8, 12) the second plane's issue stalls in the counter until enough of the first has landed -- "two planes in flight"
  // means two planes REQUESTED; harmless for a wave that does nothing else, and the wait count itself must fit
  // (tools/isa_hazards.py checks kPerPlane <= 63 and that exactly one or two planes are issued between a barrier and the wait).
  static constexpr int kPerPlane = NS * NK + NH;
  static_assert(kPerPlane <= 63, "s_waitcnt vmcnt(kPerPlane): the counter has six bits");
  // the centre-only ring of the FD builds
  static constexpr int NC = FD ? 5 : 0;               // fx, fy, ksi, fz, ft
  static constexpr int NRC = TY + 2;                  // rows y0-1 .. y0+TY
  static constexpr int NKC = (NRC + 3) / 4;
  static constexpr int kCHaloLanes = NC * 2 * TY;     // [array][side][core row]
  static constexpr int NHC = (kCHaloLanes + 63) / 64;
  static constexpr int kCRowFloats = NC * NRC * 64;
  static constexpr int kCHaloOff = kCRowFloats;
  static constexpr int kCSlotFloats = FD ? kCRowFloats + NHC * 256 : 4;
  static constexpr int kCSlots = 2;
  static constexpr int kPerPlaneC = NC * NKC + NHC;   // DMA instructions per centre-only plane
};

// A wave-uniform pointer moved into scalar registers for good: the "s" operands of the hand-issued memory instructions need
// one, and the compiler does not always insert the v_readfirstlane itself when it has formed the address with vector
// 64-bit arithmetic (it then hands the assembler a VGPR pair: "invalid operand for instruction").
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p)
{
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return reinterpret_cast<T*>((static_cast<unsigned long long>(hi) << 32) | lo);
}

// timing experiments: a value the compiler knows nothing about, made from a register (stands in for an LDS read)
__device__ __forceinline__ float opaque(float seed)
{
  asm volatile("" : "+v"(seed));
  return seed;
}

// The lane number, made anew: 16 waves per workgroup (TY = 12) leave 128 registers per lane, and the one value the
// allocator then keeps in scratch across the march is the lane number itself -- reloading it costs a scratch load per step
// whose s_waitcnt vmcnt(0) also waits for the stores issued by hand; two vector instructions make it again instead.
__device__ __forceinline__ int fresh_lane()
{
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// one 1-KiB piece: lane L's 16 bytes land at lds_dst + 16 L
__device__ __forceinline__ void dma16(const float* base, unsigned byte_off, float* lds_dst)
{
  const unsigned m0v = static_cast<unsigned>(reinterpret_cast<unsigned long>((LdsFloat*)lds_dst));
  asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(byte_off), "s"(base), "{m0}"(m0v) : "memory");
}
__device__ __forceinline__ void dma16_lane(const float* lane_addr, float* lds_dst)
{
  const unsigned m0v = static_cast<unsigned>(reinterpret_cast<unsigned long>((LdsFloat*)lds_dst));
  asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(lane_addr), "{m0}"(m0v) : "memory");
}

struct CarryP {  // PAIR_SP: what phi/ksi of a voxel reuses from its sweep
  float fx, fy, fz, ft;
  float D[9];  // U[x+1]-U[x-1], U[y+1]-U[y-1], U[z+1]-U[z-1], then V, W: the first operation of the nine flow derivatives
};

// ---- the weights 1 / (2 sqrt(a)) without the IEEE square root and division sequences ----------------------------------------
// The reference's phi = 1.f / (2.f * sqrtf(a)) (solve_3d.cu:203-204, 259-260) costs two of the longest sequences the compiler
// emits: a correctly rounded square root (v_sqrt_f32 and two fused corrections with their selects and range scaling, ~17 issue
// slots) and a correctly rounded division (v_div_scale x 2, v_rcp_f32, five fused multiply-adds, v_div_fmas, v_div_fixup, ~14).
// It is a function of ONE binary32 argument, so another way to the same float can be checked against it for every argument
// there is -- not by sampling.  weight_fast():
//     y = v_rsq_f32(a); h = y / 2; s0 = a y; s = fma(fma(-s0, s0, a), h, s0)      s = RN(sqrt(a))
//     t = s + s;                   r = fma(fma(-t, h, 1), h, h)                    r = RN(1 / t)
// (explicit v_fma_f32: single correctly rounded operations, so the result is a deterministic function of the bits of a on this
// chip).  tools/lab/phi_exact_lab and, in the test suite, f3d_selftest_weights sweep all 1 677 721 601 arguments in
// [2^-100, 2^100]: s equals sqrtf(a) everywhere, and r equals the IEEE chain except where the significand of s is all ones (two
// arguments per binade: the one case in which the fused correction of a reciprocal lands on a tie).  weight_fast_ok() excludes those and
// everything outside the range (zero, subnormal, negative, infinite, NaN) with integer compares; a wave with such a lane that counts
// takes the IEEE sequences for all its lanes.  11 issue slots + 4 for the guard instead of ~32 per weight.
constexpr unsigned kWeightLo = 0x0d800000u, kWeightHi = 0x71800000u;   // 2^-100 .. 2^100
__device__ __forceinline__ float weight_fast(float a, float& s_out)
{
  const float y = __builtin_amdgcn_rsqf(a);
  const float h = 0.5f * y;
  const float s0 = a * y;
  const float s = __builtin_fmaf(__builtin_fmaf(-s0, s0, a), h, s0);
  s_out = s;
  const float t = s + s;
  return __builtin_fmaf(__builtin_fmaf(-t, h, 1.f), h, h);
}
__device__ __forceinline__ bool weight_fast_ok(float a, float s)
{
  return (__float_as_uint(a) - kWeightLo) <= (kWeightHi - kWeightLo) && (__float_as_uint(s) & 0x7fffffu) != 0x7fffffu;
}
__device__ __forceinline__ float weight_ieee(float a) { return 1.f / (2.f * sqrtf(a)); }

// A.3 for one voxel from the increments after the sweep: n?.{u,v,w} = dU, dV, dW of the six neighbours
// counts: the lane's result is stored (lanes beyond the volume hold padding: they must not send the wave down the slow roads)
__device__ __forceinline__ void phi_ksi_stage2(const CarryP& k, const S3& xm, const S3& xp, const S3& ym, const S3& yp,
                                               const S3& zm, const S3& zp, float du, float dv_c, float dw,
                                               const SolveDivs& dv, float eps_s2, float eps_d2, float& phi, float& ksi,
                                               bool counts = true, bool fast_weights = true)
{
  float q[9] = {k.D[0] + xp.u - xm.u, k.D[1] + yp.u - ym.u, k.D[2] + zp.u - zm.u,
                k.D[3] + xp.v - xm.v, k.D[4] + yp.v - ym.v, k.D[5] + zp.v - zm.v,
                k.D[6] + xp.w - xm.w, k.D[7] + yp.w - ym.w, k.D[8] + zp.w - zm.w};
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
  const float fx = k.fx, fy = k.fy, fz = k.fz, ft = k.ft;
  const float J11 = fx * fx, J22 = fy * fy, J33 = fz * fz;
  const float J12 = fx * fy, J13 = fx * fz, J23 = fy * fz;
  const float J14 = fx * ft, J24 = fy * ft, J34 = fz * ft, J44 = ft * ft;
  float s = (J11 * du + J12 * dv_c + J13 * dw + J14) * du + (J12 * du + J22 * dv_c + J23 * dw + J24) * dv_c +
            (J13 * du + J23 * dv_c + J33 * dw + J34) * dw + (J14 * du + J24 * dv_c + J34 * dw + J44);
  s = static_cast<float>(s > 0) * s;
  const float a_ksi = s + eps_d2;
  float s_phi, s_ksi;
  const float w_phi = weight_fast(a_phi, s_phi), w_ksi = weight_fast(a_ksi, s_ksi);
  const bool lane_ok = !counts || (weight_fast_ok(a_phi, s_phi) && weight_fast_ok(a_ksi, s_ksi));
  if (__builtin_expect(fast_weights && __builtin_amdgcn_ballot_w64(!lane_ok) == 0, 1)) {
    phi = w_phi;
    ksi = w_ksi;
  } else {
    phi = weight_ieee(a_phi);
    ksi = weight_ieee(a_ksi);
  }
}

// ABL (timing experiments only, wrong results): bit 0 = the loader issues nothing after the prologue, bit 1 = no stage
// arithmetic (LDS traffic, barriers and stores stay), bit 2 = the compute waves only keep the barriers, bit 3 = arithmetic only behind the
// prologue (no LDS reads, no step barriers), bit 5 (with bit 3: F3D_ABLATE8=40) = the prologue fetches one plane instead of three
// bit 6 (F3D_ABLATE8=64, two sweeps, TY <= 8): a THIRD stage per step -- what a (sweep, sweep, sweep) launch would cost: stage 2's
// results go to a third LDS image, a second carry set is kept for the plane before, stage 2's arithmetic runs once more on the
// neighbours read back from that image, and every chunk marches one plane further (the third stage trails by one more plane)
// FD: the kernel reads the frame derivatives fx, fy, fz, ft (k_frame_derivatives, once per level) instead of the frames: they
// are centre values, so the frame entries of every neighbour -- their LDS reads, lane shifts, differences and the three
// divisions by 4h -- drop out of stage 1 (a seventh of its arithmetic), for two more arrays to stream.
template <int MODE, int TY, int ABL, bool FD, bool YM, bool TIGHT>
__device__ __forceinline__ void pair8_body(const PairArgs& a, const F3dGeo& g, int zchunk, int ntx, int nty, int n_tiles, int xcd_remap)
{
  static_assert(!(YM && FD), "the frame-derivative launchers march along z only");
  static_assert(!TIGHT || (YM && ABL == 0), "a tile without halo rows holds every row of the volume: thin volumes marched along y");
  constexpr int RH = TIGHT ? 0 : 1;   // halo rows on either side of the tile's rows that get a row wave (and twice that many ring rows)
  // rows of a tile / march direction: (y, z) or, for thin volumes, (z, y).  YM launches cover the whole volume (no slab window).
  const int RDIM = YM ? g.D : g.H;   // extent along the tile's rows
  const int MDIM = YM ? g.H : g.D;   // extent along the march
  const int m_lo = YM ? 0 : g.z_lo, m_hi = YM ? g.H : g.z_hi;
  constexpr int NA = FD ? 12 : 10;
  using L = Pair8Lds<TY, NA, FD, TIGHT>;
  constexpr int NR = L::NR, NJ = L::NJ, NK = L::NK, NJP = L::NJP, NH = L::NH;
  static_assert(TY <= 32, "the column wave holds one halo voxel per lane: 2 x TY <= 64");
  __shared__ __attribute__((aligned(16))) float ring[L::kSlots][L::kSlotFloats];
  __shared__ __attribute__((aligned(16))) float cring[L::kCSlots][L::kCSlotFloats];  // FD: the centre-only inputs (4 floats otherwise)
  // FD: ring index of a stencilled input (u .. phi are inputs 2 .. 8) and centre-ring index of fx, fy, ksi, fz, ft (inputs 0, 1, 9, 10, 11)
  constexpr int SB = FD ? 2 : 0;   // ring array = input - SB
  __shared__ float img1[2][3][NR][kLanes];  // stage-1 results of the row waves: S = U + dU' (SS) or dU' (SP)
  __shared__ float hc1[2][3][2][32];        // the same for the two halo columns: [component][side][core row]
  __shared__ float img3[(ABL & 64) ? 3 : 1][(ABL & 64) ? NR : 1][(ABL & 64) ? kLanes : 1];  // lab: stage-2 results for a third stage

  int tile = static_cast<int>(blockIdx.x);
  if (xcd_remap) {
    const int per_xcd = (n_tiles + 7) / 8;
    tile = (tile % 8) * per_xcd + tile / 8;
  }
  if (tile >= n_tiles) return;
  const int tx = tile % ntx;
  const int ty = (tile / ntx) % nty;
  const int tz = tile / (ntx * nty);

  int lane = threadIdx.x;  // re-made at the top of every step where registers are short (see fresh_lane)
  const int r = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  const bool colw = r == NR;
  const bool loader = r == NR + 1;
  const int z0 = m_lo + tz * zchunk;
  const int z1 = min(z0 + zchunk + ((ABL & 64) ? 1 : 0), m_hi);   // (lab, bit 6: the third stage trails by one more plane)
  const int qs = z0 > 0 ? z0 - 1 : 0;        // first and last plane of stage 1
  const int qe = z1 < MDIM ? z1 : MDIM - 1;
  const int q_end = z1 < MDIM ? qe : qe + 1;  // the top chunk takes one more step: stage 2 of plane D-1 alone
  const int p_last = qe + 1;                 // last plane the ring ever holds (mirrored when it is D)
  const int x0 = tx * kLanes;
  const int y0 = ty * TY;
  const bool left_face = tx == 0;
  const bool right_face = x0 + kLanes >= g.W;
  const bool tile_at_x_face = __builtin_amdgcn_readfirstlane(static_cast<int>(left_face || right_face)) != 0;
  // the column wave's right halo column is the last column of the volume when W = 64 k + 1
  const bool col_at_x_face = __builtin_amdgcn_readfirstlane(static_cast<int>(left_face || right_face || x0 + kLanes == g.W - 1)) != 0;

  // PAIR_SP: planes whose sweep result this chunk stores -- its own, and at the ends of a window that keeps its edge planes one more
  const int st_lo = (a.keep_below && z0 == m_lo) ? z0 - 1 : z0;
  const int st_hi = (a.keep_above && z1 == m_hi) ? z1 + 1 : z1;
  const int zb = qs > 0 ? qs - 1 : 0;  // lowest plane touched: byte offsets inside the chunk stay small and positive
  const unsigned z_stride_b = static_cast<unsigned>(g.Hc) * static_cast<unsigned>(g.pitch) * 4u;
  const unsigned y_stride_b = static_cast<unsigned>(g.pitch) * 4u;
  const unsigned plane_b = YM ? y_stride_b : z_stride_b;   // one march step
  const unsigned row_b = YM ? z_stride_b : y_stride_b;     // one tile row
  const size_t base_off = YM ? static_cast<size_t>(zb) * static_cast<size_t>(g.pitch) : f3d_row(g, 0, zb);

  // ================================================== loader wave ==================================================
  if (loader) {
    // the youngest wave of its SIMD would otherwise get the issue slots the arithmetic of the older two leaves over
    __builtin_amdgcn_s_setprio(3);
    const float* base[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) base[i] = uniform_ptr(a.in[i] + base_off);
    // row pieces: lane -> (row 4k + lane/16, floats 4 (lane%16) ..)
    unsigned rowb[NK];
    bool rowv[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int j = 4 * k + (lane >> 4);
      rowv[k] = j < NJ;
      const int yrow = f3d_clampi(f3d_mir(y0 - 2 * RH + (rowv[k] ? j : 0), RDIM), 0, RDIM - 1);
      rowb[k] = static_cast<unsigned>(yrow) * row_b + static_cast<unsigned>(x0 + 4 * (lane & 15)) * 4u;
    }
    // halo pieces: lane' = 64 h + lane -> [array][side][row]; four floats left of the tile (x0-4 ..) or right of it (x0+64 ..)
    const float* hptr[NH];
    bool hv[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int lp = 64 * h + lane;
      hv[h] = lp < L::kHaloLanes;
      const int lq = hv[h] ? lp : 0;
      const int arr = lq / (2 * NJ) + SB;
      const int s = (lq / NJ) & 1;
      const int j = lq % NJ;
      const float* b = base[SB];
#pragma unroll
      for (int i = SB + 1; i < SB + L::NS; ++i)
        if (arr == i) b = base[i];
      const int yrow = f3d_clampi(f3d_mir(y0 - 2 * RH + j, RDIM), 0, RDIM - 1);
      // tiles at an x face fetch a piece from inside the row instead (never used: the mirror rule substitutes there)
      const int xc = s == 0 ? (left_face ? 0 : x0 - 4) : (right_face ? x0 + kLanes - 4 : x0 + kLanes);
      hptr[h] = b + static_cast<size_t>(yrow) * static_cast<size_t>(row_b >> 2) + xc;
    }
    auto issue = [&](int p) {  // plane p (mirrored for the address) into slot (p - (qs-1)) mod 3
      float* slot = &ring[(p - qs + 1) % L::kSlots][0];
      const int zz = f3d_mir(p, MDIM);
      const unsigned poff = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(zz - zb) * plane_b)));
      unsigned off[NK];
#pragma unroll
      for (int k = 0; k < NK; ++k) off[k] = rowb[k] + poff;
#pragma unroll
      for (int i = 0; i < L::NS; ++i) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          if (NJ % 4 == 0 || rowv[k]) dma16(base[SB + i], off[k], slot + (i * NJP + 4 * k) * kLanes);
        }
      }
#pragma unroll
      for (int h = 0; h < NH; ++h)
        if (hv[h]) dma16_lane(hptr[h] + (poff >> 2), slot + L::kHaloOff + h * 256);
    };
    // FD: the centre-only inputs of plane p into centre slot (p - qs) mod 2: rows y0-1 .. y0+TY and the halo columns of the core rows
    constexpr int NKC = FD ? L::NKC : 1, NHC = FD ? L::NHC : 1;
    constexpr int kCIn[5] = {0, 1, 9, 10, 11};
    unsigned crowb[NKC];
    bool crowv[NKC];
    const float* chptr[NHC];
    bool chv[NHC];
    if constexpr (FD) {
#pragma unroll
      for (int k = 0; k < NKC; ++k) {
        const int j = 4 * k + (lane >> 4);
        crowv[k] = j < L::NRC;
        const int yrow = f3d_clampi(f3d_mir(y0 - 1 + (crowv[k] ? j : 0), RDIM), 0, RDIM - 1);
        crowb[k] = static_cast<unsigned>(yrow) * row_b + static_cast<unsigned>(x0 + 4 * (lane & 15)) * 4u;
      }
#pragma unroll
      for (int h = 0; h < NHC; ++h) {
        const int lp = 64 * h + lane;
        chv[h] = lp < L::kCHaloLanes;
        const int lq = chv[h] ? lp : 0;
        const int arr = lq / (2 * TY);
        const int s = (lq / TY) & 1;
        const int j = lq % TY;
        const float* b = base[kCIn[0]];
#pragma unroll
        for (int i = 1; i < 5; ++i)
          if (arr == i) b = base[kCIn[i]];
        const int yrow = f3d_clampi(f3d_mir(y0 + j, RDIM), 0, RDIM - 1);
        const int xc = s == 0 ? (left_face ? 0 : x0 - 4) : (right_face ? x0 + kLanes - 4 : x0 + kLanes);
        chptr[h] = b + static_cast<size_t>(yrow) * static_cast<size_t>(row_b >> 2) + xc;
      }
    }
    auto issue_c = [&](int p) {
      if constexpr (FD) {
        float* slot = &cring[(p - qs) & 1][0];
        const int zz = f3d_mir(p, MDIM);
        const unsigned poff = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(zz - zb) * plane_b)));
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
          for (int k = 0; k < NKC; ++k)
            if (L::NRC % 4 == 0 || crowv[k]) dma16(base[kCIn[i]], crowb[k] + poff, slot + (i * L::NRC + 4 * k) * kLanes);
        }
#pragma unroll
        for (int h = 0; h < NHC; ++h)
          if (chv[h]) dma16_lane(chptr[h] + (poff >> 2), slot + L::kCHaloOff + h * 256);
      }
    };
    // prologue: planes qs-1, qs, qs+1 fill the three slots and must have landed before anybody reads
    issue(qs - 1);
    if (!(ABL & 32)) {  // ABL 32 (with 8: prologue-only probe): ONE plane instead of three -- is the prologue issue- or latency-bound?
      issue(qs);
      issue(qs + 1);
      issue_c(qs);      // (a centre-only plane is needed as z+1 plane and as centre: never plane qs-1)
      issue_c(qs + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ABL & 8) return;
    for (int q = qs; q <= q_end; ++q) {
      __syncthreads();  // B_q: the slots of planes q-1 and q have been read for the last time
      if (!(ABL & 1)) {
        if (q == qs && q + 2 <= p_last) issue(q + 2);  // steady state: issued one step ago
        // FD: the centre-only plane q+2 goes into the slot of plane q, read for the last time before B_q (it is read only at the very
        // start of step q-1), and must have landed at B_{q+1}: it is issued BEFORE plane q+3 of the ring, so the counted wait below
        // -- everything but the youngest kPerPlane instructions -- covers it
        if (q + 3 <= p_last) {
          issue_c(q + 2);
          issue(q + 3);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::kPerPlane) : "memory");  // plane q+2 has landed, q+3 stays in flight
          continue;
        }
        if (q + 2 <= p_last) issue_c(q + 2);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    return;
  }

  // ================================================= compute waves =================================================
  FDivs fdivs;
  fdivs.x4 = UDiv{a.r4[0], a.d4[0]}; fdivs.y4 = UDiv{a.r4[1], a.d4[1]}; fdivs.z4 = UDiv{a.r4[2], a.d4[2]};
  fdivs.ok = a.fdivs_ok != 0;
  SolveDivs sdivs = {};
  if (MODE == PAIR_SP) {
    sdivs.x2 = UDiv{a.r2[0], a.d2[0]}; sdivs.y2 = UDiv{a.r2[1], a.d2[1]}; sdivs.z2 = UDiv{a.r2[2], a.d2[2]};
    sdivs.x4 = fdivs.x4; sdivs.y4 = fdivs.y4; sdivs.z4 = fdivs.z4;
    sdivs.ok = a.sdivs_ok != 0;
  }

  // row waves
  const int y = y0 - RH + r;
  const int yy = f3d_clampi(f3d_mir(y, RDIM), 0, RDIM - 1);
  const int x = x0 + lane;
  const bool core = r >= RH && r < RH + TY;
  const bool owner = core && x < g.W && y < RDIM;
  const int side = lane < 32 ? 0 : 1;
  const int jr = r + RH;  // ring row of this wave's row
  // ring rows of its two row neighbours: the rows beside it -- or, in a tile without halo rows, the mirror image at a face of the volume
  const int jr_m = (TIGHT && y == 0) ? jr + 1 : jr - 1;
  const int jr_p = (TIGHT && y == RDIM - 1) ? jr - 1 : jr + 1;
  // Stage 2 reads the stage-1 results of rows y-1 and y+1 from the LDS image; at a y face of the volume the missing neighbour is
  // the opposite one (mirror rule), which for a row wave is simply the other image row: chosen here, once, by a scalar select
  // instead of six vector selects per step.
  const int r_ym = y == 0 ? r + 1 : r - 1;
  const int r_yp = y == RDIM - 1 ? r - 1 : r + 1;
  const unsigned xb = static_cast<unsigned>(x) * 4u;
  // column wave: lane = side * 32 + core row (lanes beyond TY rows repeat the last row and publish nothing)
  const bool cactive = (lane & 31) < TY;
  const int crow = cactive ? (lane & 31) : TY - 1;
  const int cy = y0 + crow;
  const int cx = side == 0 ? x0 - 1 : x0 + kLanes;
  const int jc = crow + 2 * RH;  // ring row of the column wave's voxel
  const int jc_m = (TIGHT && cy == 0) ? jc + 1 : jc - 1;
  const int jc_p = (TIGHT && cy == RDIM - 1) ? jc - 1 : jc + 1;
  const int e_near = side == 0 ? 3 : 0;  // element of the 4-float halo piece next to the tile, and the one beyond it
  const int e_far = side == 0 ? 2 : 1;

  float* obase[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) obase[i] = (i < 3 || MODE == PAIR_SP) ? uniform_ptr(a.out[i] + base_off) : nullptr;
  auto rowoff = [&](int yrow, int zz) __attribute__((always_inline)) {
    return static_cast<unsigned>(__builtin_amdgcn_readfirstlane(
        static_cast<int>(static_cast<unsigned>(zz - zb) * plane_b + static_cast<unsigned>(yrow) * row_b)));
  };

  // raw values of one ring row (own lane) / of one halo piece element
  float seedv = static_cast<float>(lane) * 0.01f + 1.f;
  auto fake_raw = [&](PlaneRegs& p) __attribute__((always_inline)) {  // ABL bit 3: no LDS traffic, values from registers
    p.f0 = opaque(seedv); p.f1 = opaque(seedv); p.u = opaque(seedv); p.v = opaque(seedv); p.w = opaque(seedv);
    p.su = opaque(seedv); p.dv = opaque(seedv); p.dw = opaque(seedv); p.phi = opaque(seedv); p.ksi = opaque(seedv);
    p.fz = opaque(seedv); p.ft = opaque(seedv);
  };
  // `cs` (FD, with_ksi only): the centre-only slot of the same plane; its rows start at y0-1, i.e. ring row j is centre row j-1
  auto row_raw = [&](PlaneRegs& p, const float* slot, int j, bool with_ksi, const float* cs = nullptr) __attribute__((always_inline)) {
    if (ABL & 8) return fake_raw(p);
    const float* d = slot + j * kLanes + lane;
    constexpr int st = NJP * kLanes;
    if (!FD) {
      p.f0 = d[F0 * st];
      p.f1 = d[F1 * st];
    }
    p.u = d[(U - SB) * st]; p.v = d[(V - SB) * st]; p.w = d[(Wf - SB) * st];
    p.su = d[(DU - SB) * st]; p.dv = d[(DV - SB) * st]; p.dw = d[(DW - SB) * st]; p.phi = d[(PHI - SB) * st];
    if (!FD && with_ksi) p.ksi = d[9 * st];
    if (FD && !with_ksi) p.f0 = p.f1 = 0.f;   // neighbours' frame values are not looked at in FD builds
    if (FD && with_ksi) {
      const float* c = cs + (j - 1) * kLanes + lane;
      constexpr int ct = L::NRC * kLanes;
      p.f0 = c[0 * ct]; p.f1 = c[1 * ct]; p.ksi = c[2 * ct]; p.fz = c[3 * ct]; p.ft = c[4 * ct];   // fx, fy, ksi, fz, ft
    }
  };
  // column wave, own voxel (FD, with_ksi): centre-only halo pieces are indexed by CORE row, ring row j = core row + 2
  auto halo_raw = [&](PlaneRegs& p, const float* slot, int s, int j, int e, bool with_ksi, const float* cs = nullptr) __attribute__((always_inline)) {
    if (ABL & 8) return fake_raw(p);
    const float* d = slot + L::kHaloOff + (s * NJ + j) * 4 + e;
    constexpr int st = 2 * NJ * 4;
    if (!FD) {
      p.f0 = d[F0 * st];
      p.f1 = d[F1 * st];
    }
    p.u = d[(U - SB) * st]; p.v = d[(V - SB) * st]; p.w = d[(Wf - SB) * st];
    p.su = d[(DU - SB) * st]; p.dv = d[(DV - SB) * st]; p.dw = d[(DW - SB) * st]; p.phi = d[(PHI - SB) * st];
    if (!FD && with_ksi) p.ksi = d[9 * st];
    if (FD && !with_ksi) p.f0 = p.f1 = 0.f;
    if (FD && with_ksi) {
      const float* c = cs + L::kCHaloOff + (s * TY + (j - 2)) * 4 + e;
      constexpr int ct = 2 * TY * 4;
      p.f0 = c[0 * ct]; p.f1 = c[1 * ct]; p.ksi = c[2 * ct]; p.fz = c[3 * ct]; p.ft = c[4 * ct];
    }
  };

  S3 hM = {0.f, 0.f, 0.f}, hC = {0.f, 0.f, 0.f};  // stage-1 result of planes q-2 and q-1 (SS: S = U + dU'; SP: dU')
  float hC_dv = 0.f, hC_dw = 0.f;                 // SS: dV', dW' of plane q-1
  Carry kC = {};
  Carry kC2 = {};   // lab, bit 6: the carry of the plane before (a third stage needs it one step longer)
  CarryP pC = {};

  // Raw neighbours of a plane for stage 1 -- the rows above and below and the x-halo column (column wave: its two y
  // neighbours and the column beyond).  They are fetched at the END of the step before the one that uses them: the plane
  // has been in the ring since the barrier before last, so the reads need no barrier of their own, and issued there they
  // run in the shadow of the other waves' arithmetic instead of in a burst of all twelve waves right after the barrier
  // (with every wave in the same phase the LDS pipe -- 128 B per clock for 4-byte reads -- was a phase of its own, a third
  // of the step, during which the vector unit idled).
  // They are finished (S = U + dU) right there, so six values per neighbour cross the barrier, not nine; PAIR_SP also keeps
  // U[y+1] - U[y-1] (the first operation of its y derivatives) and the raw U, V, W of the halo column.
  Face6 nYm = {}, nYp = {}, nX = {};
  S3 nDy = {0.f, 0.f, 0.f}, nXr = {0.f, 0.f, 0.f};
  Face6 nIn = {};                    // column wave: the tile's own edge column
  S3 nInr = {0.f, 0.f, 0.f};
  auto fetch_neighbours = [&](auto colw_c, const float* S) __attribute__((always_inline)) {
    constexpr bool CW = decltype(colw_c)::value;
    PlaneRegs T0, T1, T2;
    if constexpr (CW) {
      PlaneRegs T;
      const float* d = S + jc * kLanes + (side ? kLanes - 1 : 0);
      constexpr int st = NJP * kLanes;
      if (ABL & 8) {
        fake_raw(T);
      } else {
        if (!FD) {
          T.f0 = d[F0 * st];
          T.f1 = d[F1 * st];
        } else {
          T.f0 = T.f1 = 0.f;
        }
        T.u = d[(U - SB) * st]; T.v = d[(V - SB) * st]; T.w = d[(Wf - SB) * st];
        T.su = d[(DU - SB) * st]; T.dv = d[(DV - SB) * st]; T.dw = d[(DW - SB) * st]; T.phi = d[(PHI - SB) * st];
      }
      nInr = {T.u, T.v, T.w};
      plane_finish(T);
      nIn = plane_face(T);
    }
    if constexpr (CW) {
      halo_raw(T0, S, side, jc_m, e_near, false);
      halo_raw(T1, S, side, jc_p, e_near, false);
      halo_raw(T2, S, side, jc, e_far, false);
    } else {
      row_raw(T0, S, jr_m, false);
      row_raw(T1, S, jr_p, false);
      halo_raw(T2, S, side, jr, e_near, false);
    }
    if (MODE == PAIR_SP) {
      nDy = {T1.u - T0.u, T1.v - T0.v, T1.w - T0.w};
      nXr = {T2.u, T2.v, T2.w};
    }
    plane_finish(T0);
    plane_finish(T1);
    plane_finish(T2);
    nYm = plane_face(T0);
    nYp = plane_face(T1);
    nX = plane_face(T2);
  };

  // M, C, P: finished planes q-1, q, q+1; P is read from the ring at the start of the step.  Stage 1 is computed in every
  // step, also in the one extra step of the top chunk (q = D, from whatever the ring holds): its results are neither stored
  // nor looked at there, and an unconditional body spares the carried values a copy per step.
  // SLOT = ring slot of plane q+1, a compile-time constant: slots are numbered from plane qs-1 and the march is unrolled
  // three steps deep, so every LDS address below is a lane offset plus an immediate
  auto step = [&](auto colw_c, auto slot_c, PlaneRegs& M, PlaneRegs& C, PlaneRegs& P, int q) __attribute__((always_inline)) {
    constexpr bool CW = decltype(colw_c)::value;  // the column wave runs a loop of its own: no value merges with the row waves
    constexpr int SLOT = decltype(slot_c)::value;
    if constexpr (TY > 8 || TIGHT) lane = fresh_lane();
    if (!(ABL & 8)) __syncthreads();  // B_q: plane q+1 is in the ring, img1 / hc1 of plane q-1 are complete
    if (ABL & 4) return;
    const float* Sp = &ring[SLOT][0];
    const float* Cp = &cring[FD ? ((q + 1 - qs) & 1) : 0][0];   // FD: centre-only inputs of plane q+1
    const bool do1 = q <= qe;
    const int b = q & 1;

    float r_du = 0.f, r_dv = 0.f, r_dw = 0.f;
    Carry kN;
    CarryP pN;
    {
      Face6 xm, xp;
      const Face6 ym = nYm, yp = nYp;
      S3 rxm = {}, rxp = {};  // PAIR_SP: raw u, v, w of the x neighbours
      int vx, vy;
      if constexpr (CW) {
        halo_raw(P, Sp, side, jc, e_near, true, Cp);
        plane_finish(P);
        const S3 router = nXr;
        const Face6 outer = nX;
        const S3 rinner = nInr;
        const Face6 inner = nIn;
#pragma unroll
        for (int i = 0; i < kNL; ++i) {
          xm.v[i] = side ? inner.v[i] : outer.v[i];
          xp.v[i] = side ? outer.v[i] : inner.v[i];
        }
        rxm = side ? rinner : router;
        rxp = side ? router : rinner;
        vx = cx;
        vy = cy;
        // A halo column can be the LAST column of the volume (W = 64 k + 1): its right neighbour lies beyond the row, where
        // the 16-byte piece holds padding instead of the mirrored column, so the mirror rule is applied here as well.
        if (cx == g.W - 1) {
          xp = xm;
          rxp = rxm;
        }
      } else {
        row_raw(P, Sp, jr, true, Cp);
        plane_finish(P);
        const S3 rx = nXr;
        const Face6 xf = nX;
        const Face6 cf = plane_face(C);
#pragma unroll
        for (int i = 0; i < kNL; ++i) {
          xm.v[i] = lane_left_or(cf.v[i], xf.v[i]);
          xp.v[i] = lane_right_or(cf.v[i], xf.v[i]);
        }
        if (MODE == PAIR_SP) {
          rxm = {lane_left_or(C.u, rx.u), lane_left_or(C.v, rx.v), lane_left_or(C.w, rx.w)};
          rxp = {lane_right_or(C.u, rx.u), lane_right_or(C.v, rx.v), lane_right_or(C.w, rx.w)};
        }
        vx = x;
        vy = y;
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
      }
      const Face6 cfc = plane_face(C);
      if (ABL & 2) {
        kN = Carry{};
        r_du = xm.v[0] + xp.v[1] + ym.v[2] + yp.v[3] + M.su + P.sv + C.ksi + rxm.u + nDy.v;
        r_dv = xm.v[4] + xp.v[5] + ym.v[0] + yp.v[1] + M.f0 + P.f1 + C.u + rxp.w + nDy.u;
        r_dw = xm.v[2] + xp.v[3] + ym.v[4] + yp.v[5] + M.phi + P.phi + C.dv + C.dw + C.v + C.w;
        kN.J12 = r_du; kN.d1 = r_dv; kN.pw[0] = r_dw;
      } else
      {
        // the tile's row neighbours (ym, yp here) and the march neighbours (M, P) in their geometric roles
        const Face6 fM = plane_face(M), fP = plane_face(P);
        const bool row_p = vy < RDIM - 1, row_m = vy > 0, mar_p = q < MDIM - 1, mar_m = q > 0;
        // ABL bit 4 (timing only, wrong results): the frame derivatives are taken from registers as if they had been read (what a
        // frame-derivative build of THIS tile shape would save in arithmetic, without its two extra arrays)
        constexpr bool FDA = FD || (ABL & 16) != 0;
        sweep_stage1<FDA, true>(xm, xp, YM ? fM : ym, YM ? fP : yp, YM ? ym : fM, YM ? yp : fP, cfc.v, C.u, C.v, C.w, C.dv, C.dw,
                                C.ksi, a.hx, a.hy, a.hz, fdivs, a.alpha, vx < g.W - 1, vx > 0, YM ? mar_p : row_p, YM ? mar_m : row_m,
                                YM ? row_p : mar_p, YM ? row_m : mar_m, r_du, r_dv, r_dw, kN, C.f0, C.f1, FD ? C.fz : C.phi,
                                FD ? C.ft : C.ksi, CW ? col_at_x_face : tile_at_x_face, a.w[0], a.w[1], a.w[2]);
      }
      pN.fx = kN.fx; pN.fy = kN.fy; pN.fz = kN.fz; pN.ft = kN.ft;
      {
        const S3 dRow = nDy, dMar = {P.u - M.u, P.v - M.v, P.w - M.w};   // U[+1] - U[-1] along the rows / the march
        const S3 dY = YM ? dMar : dRow, dZ = YM ? dRow : dMar;
        pN.D[0] = rxp.u - rxm.u; pN.D[1] = dY.u; pN.D[2] = dZ.u;
        pN.D[3] = rxp.v - rxm.v; pN.D[4] = dY.v; pN.D[5] = dZ.v;
        pN.D[6] = rxp.w - rxm.w; pN.D[7] = dY.w; pN.D[8] = dZ.w;
      }
    }
    // what a neighbour reads of this voxel in stage 2: SS U + dU', SP dU'
    const S3 sN = MODE == PAIR_SS ? S3{C.u + r_du, C.v + r_dv, C.w + r_dw} : S3{r_du, r_dv, r_dw};
    if (do1 && !(ABL & 8)) {
      if constexpr (CW) {
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
    // PAIR_SP: the sweep's result is final -- store it for the planes this chunk owns
    if (MODE == PAIR_SP && do1 && owner && q >= st_lo && q < st_hi) {
      const unsigned off = xb + rowoff(yy, q);
      gst(obase[0], off, r_du);
      gst(obase[1], off, r_dv);
      gst(obase[2], off, r_dw);
    }

    // stage 2 of plane t = q - 1
    const int t = q - 1;
    const bool do2 = !CW && core && t >= z0;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (do2) {
      const int pb = t & 1;
      S3 ym, yp, xm, xp, zm, zp;
      float eu, ev, ew;
      if (ABL & 8) {
        ym.u = opaque(seedv); ym.v = opaque(seedv); ym.w = opaque(seedv);
        yp.u = opaque(seedv); yp.v = opaque(seedv); yp.w = opaque(seedv);
        eu = opaque(seedv); ev = opaque(seedv); ew = opaque(seedv);
      } else {
        ym.u = img1[pb][0][r_ym][lane]; ym.v = img1[pb][1][r_ym][lane]; ym.w = img1[pb][2][r_ym][lane];
        yp.u = img1[pb][0][r_yp][lane]; yp.v = img1[pb][1][r_yp][lane]; yp.w = img1[pb][2][r_yp][lane];
        eu = hc1[pb][0][side][r - RH]; ev = hc1[pb][1][side][r - RH]; ew = hc1[pb][2][side][r - RH];
      }
      xm.u = lane_left_or(hC.u, eu); xm.v = lane_left_or(hC.v, ev); xm.w = lane_left_or(hC.w, ew);
      xp.u = lane_right_or(hC.u, eu); xp.v = lane_right_or(hC.v, ev); xp.w = lane_right_or(hC.w, ew);
      zm = hM;
      zp = sN;
      // mirror rule at the faces of the volume: the missing neighbour is the opposite one (stage-1 values of voxels outside
      // the volume are never looked at)
      if (tile_at_x_face) {
        if (x == 0) xm = xp;
        if (x == g.W - 1) xp = xm;
      }
      // (y faces: r_ym / r_yp already name the opposite row there)
      if (t == 0) zm = zp;
      if (t == MDIM - 1) zp = zm;
      if (ABL & 2) {
        o0 = xm.u + xp.v + ym.w + kC.J12;
        o1 = yp.u + zm.v + zp.w + kC.d1;
        o2 = xm.w + yp.v + zp.u + kC.pw[0] + hC_dv + hC_dw;
      } else if (MODE == PAIR_SS)   // (ym, yp) = row neighbours, (zm, zp) = march neighbours: handed over in their geometric roles
        sweep_stage2(kC, xm, xp, YM ? zm : ym, YM ? zp : yp, YM ? ym : zm, YM ? yp : zp, hC_dv, hC_dw, o0, o1, o2);
      else
        phi_ksi_stage2(pC, xm, xp, YM ? zm : ym, YM ? zp : yp, YM ? ym : zm, YM ? yp : zp, hC.u, hC.v, hC.w, sdivs, a.eps_s2,
                       a.eps_d2, o0, o1, owner, a.plain_division == 0);
    }
    if constexpr ((ABL & 64) != 0) {
      // lab: a third stage with the instruction stream of a real one (WRONG values): publish stage 2's results, read the row
      // neighbours and the halo column back, shift the lanes, run the sweep's second-stage arithmetic on the carry of plane q-2
      if (do2) {
        img3[0][r][lane] = o0; img3[1][r][lane] = o1; img3[2][r][lane] = o2;
        S3 ym, yp, xm, xp;
        ym.u = img3[0][r_ym][lane]; ym.v = img3[1][r_ym][lane]; ym.w = img3[2][r_ym][lane];
        yp.u = img3[0][r_yp][lane]; yp.v = img3[1][r_yp][lane]; yp.w = img3[2][r_yp][lane];
        const float eu = hc1[b][0][side][r - 1], ev = hc1[b][1][side][r - 1], ew = hc1[b][2][side][r - 1];
        xm.u = lane_left_or(o0, eu); xm.v = lane_left_or(o1, ev); xm.w = lane_left_or(o2, ew);
        xp.u = lane_right_or(o0, eu); xp.v = lane_right_or(o1, ev); xp.w = lane_right_or(o2, ew);
        const S3 zm = hM, zp = {o0, o1, o2};
        float p0, p1, p2;
        sweep_stage2(kC2, xm, xp, ym, yp, zm, zp, o1, o2, p0, p1, p2);
        o0 = p0; o1 = p1; o2 = p2;
      }
      kC2 = kC;
    }
    asm volatile("" ::"v"(o0), "v"(o1), "v"(o2), "v"(sN.u), "v"(sN.v), "v"(sN.w));
    hM = hC;
    hC = sN;
    hC_dv = r_dv;
    hC_dw = r_dw;
    kC = kN;
    pC = pN;
    if (do2 && owner) {
      const unsigned off = xb + rowoff(yy, t);
      if (MODE == PAIR_SS) {
        gst(obase[0], off, o0);
        gst(obase[1], off, o1);
        gst(obase[2], off, o2);
      } else {
        gst(obase[3], off, o0);
        gst(obase[4], off, o1);
      }
    }
    fetch_neighbours(colw_c, Sp);  // for step q+1
  };

  __syncthreads();  // prologue barrier: planes qs-1, qs, qs+1 are in the ring
  PlaneRegs A = {}, B = {}, Cc = {};
  const float* Sm = &ring[0][0];  // plane qs-1
  const float* S0 = &ring[1][0];  // plane qs
  using Slot0 = std::integral_constant<int, 0>;
  using Slot1 = std::integral_constant<int, 1>;
  using Slot2 = std::integral_constant<int, 2>;
  auto march = [&](auto colw_c) __attribute__((always_inline)) {
    constexpr bool CW = decltype(colw_c)::value;
    if constexpr (CW) {
      halo_raw(A, Sm, side, jc, e_near, false);
      halo_raw(B, S0, side, jc, e_near, true, &cring[0][0]);   // plane qs sits in centre slot 0
    } else {
      row_raw(A, Sm, jr, false);
      row_raw(B, S0, jr, true, &cring[0][0]);
    }
    plane_finish(A);
    plane_finish(B);
    fetch_neighbours(colw_c, S0);
    int q = qs;
    for (; q + 2 <= q_end; q += 3) {  // step q reads plane q+1 from slot (q - qs + 2) mod 3
      step(colw_c, Slot2{}, A, B, Cc, q);
      step(colw_c, Slot0{}, B, Cc, A, q + 1);
      step(colw_c, Slot1{}, Cc, A, B, q + 2);
    }
    if (q <= q_end) step(colw_c, Slot2{}, A, B, Cc, q);
    if (q + 1 <= q_end) step(colw_c, Slot0{}, B, Cc, A, q + 1);
  };
  if (colw) march(std::true_type{});
  else march(std::false_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores issued by hand
}

// the kernels: one body, two sets of launch attributes
template <int MODE, int TY, int ABL = 0, bool FD = false, bool YM = false>
__global__ __launch_bounds__(kLanes*(TY + 4)) void k_pair8(PairArgs a, F3dGeo g, int zchunk, int ntx, int nty, int n_tiles,
                                                           int xcd_remap)
{
  pair8_body<MODE, TY, ABL, FD, YM, false>(a, g, zchunk, ntx, nty, n_tiles, xcd_remap);
}
// ... a tile without halo rows (thin volumes marched along y, all planes in the tile): TY + 2 waves and ~50 KB of LDS per workgroup, held to
// 128 registers so that TWO workgroups share a CU (four waves per SIMD) -- a level of BASELINE config 3 is then spread over twice as
// many concurrent row waves
template <int MODE, int TY>
__global__ __launch_bounds__(kLanes*(TY + 2)) __attribute__((amdgpu_waves_per_eu(4))) void k_pair8t(PairArgs a, F3dGeo g, int zchunk, int ntx,
                                                                                                   int nty, int n_tiles, int xcd_remap)
{
  pair8_body<MODE, TY, 0, false, true, true>(a, g, zchunk, ntx, nty, n_tiles, xcd_remap);
}

// one workgroup per CU at a time: z-chunks by the round model of k_sweep7 (a chunk costs its planes plus ~7 steps of prologue
// and repeated stage-1 planes, 256 workgroups run per round).  `cost` is in plane steps of ONE
// workgroup; a step of a 16-wave workgroup (TY = 12) takes ~1.28 x a step of a 12-wave one (TY = 8) -- measured at 128^3 ... 512^3
// (tools/kbench.py with F3D_PAIR8_TY): 12 rows win where the rows divide well (384^3: -9.5 %, 512^3: -4 %), 8 rows where one
// round of workgroups covers the level (256^3: +6 %, 128^3: +6 %) -- so the caller compares cost x step.
struct Pair8Plan {
  int zchunk;
  long cost;
  long wgs = 0;
};
// `rows` / `planes`: extent along the tile rows and along the march (H and the z window; D and H for a y march)
inline Pair8Plan pair8_plan_dims(int width, int rows, int planes, int ty, int zc_limit, long per_round = 256)
{
  const long tiles = static_cast<long>((width + kLanes - 1) / kLanes) * ((rows + ty - 1) / ty);
  const int max_chunks = planes > 0 ? planes : 1;  // down to one plane per chunk: three steps instead of four where one round covers it
  // what a chunk costs beside its planes, in plane steps (F3D_PAIR8_CHUNK_STEPS: launch-geometry experiments)
  static const int extra = std::getenv("F3D_PAIR8_CHUNK_STEPS") ? std::atoi(std::getenv("F3D_PAIR8_CHUNK_STEPS")) : 7;
  Pair8Plan p = {std::min(planes, zc_limit), -1};
  for (int nzc = 1; nzc <= max_chunks; ++nzc) {
    const int zc = (planes + nzc - 1) / nzc;
    if (zc > zc_limit) continue;
    const long wgs = tiles * ((planes + zc - 1) / zc);
    const long cost = ((wgs + per_round - 1) / per_round) * (zc + extra);
    if (p.cost < 0 || cost < p.cost) {
      p.cost = cost;
      p.zchunk = zc;
      p.wgs = wgs;
    }
  }
  if (p.cost < 0) {
    p.cost = static_cast<long>((tiles + per_round - 1) / per_round) * (p.zchunk + extra);
    p.wgs = tiles;
  }
  return p;
}
inline Pair8Plan pair8_plan(const F3dGeo& g, int ty, long per_round = 256)
{
  return pair8_plan_dims(g.W, g.H, g.z_hi - g.z_lo, ty, max_planes_per_chunk(g), per_round);
}

// Thin volumes march along y (YM, see the top of this file): the whole level in one launch, no slab window, every byte offset
// inside the container below 4 GiB.  Returns the tile height (4, 5 or 8 = the number of z planes a tile holds) or 0.
// F3D_PAIR8_YMARCH=0 keeps the z march, =1 takes the y march wherever it is possible (read per call: the tests run both).
inline int pair8_ymarch_rows(const F3dGeo& g)
{
  const char* e = std::getenv("F3D_PAIR8_YMARCH");
  const int mode = e ? std::atoi(e) : -1;
  if (mode == 0) return 0;
  if (g.z_base != 0 || g.z_lo != 0 || g.z_hi != g.D || g.D > 8 || g.D < 2 || g.H < 2) return 0;
  const unsigned long long bytes = static_cast<unsigned long long>(g.Hc) * static_cast<unsigned long long>(g.pitch) * 4ull *
                                   static_cast<unsigned long long>(g.D + 1);
  if (bytes >= 0xf0000000ull) return 0;
  // Measured (tools/r3_job2.sh, two sweeps / sweep + phi/ksi): 584 x 388 x 5 34.2 -> 33.8 / 40.4 -> 39.2 us, 555 x 369 x 5 32.6 -> 31.0 /
  // 38.5 -> 36.2 us, but 501 x 333 x 4 21.4 -> 23.0 / 24.7 -> 26.8 us: a level of ~1 M voxels is a handful of steps per CU either way
  // and bound by the latency of a step, not by the march direction.  By default only five planes and more, where it wins.
  // Round 4: a tile that holds exactly the volume's 4 or 5 planes needs no halo rows (k_pair8t, two workgroups per CU) and wins at both
  // depths (profiles/r04_thin_tile_kbench.txt: 584 x 388 x 5 33.1 -> 28.2 / 38.6 -> 31.1 us, 501 x 333 x 4 21.2 -> 20.3 / 23.6 -> 22.3 us
  // against the z march), so four planes march along y by default as well -- unless F3D_PAIR8_TIGHT=0 brings the halo rows back.
  const char* te = std::getenv("F3D_PAIR8_TIGHT");
  const bool tight = !(te && te[0] == '0');
  const int min_planes = tight ? 4 : 5;
  if (mode != 1 && (g.D < min_planes || g.H < 8 * g.D)) return 0;
  return g.D <= 4 ? 4 : (g.D == 5 ? 5 : 8);
}

template <int MODE, int TY, bool FD = false, bool YM = false, bool TIGHT = false>
void launch_pair8(const PairArgs& args, const F3dGeo& g, int force_zchunk, int xcd_remap)
{
  PairArgs a = args;
  pair_consts(a);
  const int rows = YM ? g.D : g.H;
  const int planes = YM ? g.H : g.z_hi - g.z_lo;
  const int ntx = (g.W + kLanes - 1) / kLanes;
  const int nty = (rows + TY - 1) / TY;
  const int zc_limit = YM ? planes : max_planes_per_chunk(g);
  // (two workgroups of a tile without halo rows share a CU: 512 per round)
  int zchunk = pair8_plan_dims(g.W, rows, planes, TY, zc_limit, TIGHT ? 512 : 256).zchunk;
  if (force_zchunk > 0) zchunk = force_zchunk;
  zchunk = std::min(zchunk, zc_limit);
  const int nz = (planes + zchunk - 1) / zchunk;
  const int n_tiles = ntx * nty * nz;
  const int per_xcd = (n_tiles + 7) / 8;
  const int blocks = xcd_remap ? per_xcd * 8 : n_tiles;
  const dim3 grid(blocks, 1, 1), block(kLanes, TY + (TIGHT ? 2 : 4), 1);
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, block, 0, f3d::stream(), a, g, zchunk, ntx, nty, n_tiles, xcd_remap); };
  if constexpr (YM && TIGHT) return go(k_pair8t<MODE, TY>);
  else if constexpr (YM) return go(k_pair8<MODE, TY, 0, false, true>);
  else if constexpr (FD) return go(k_pair8<MODE, TY, 0, true>);
  else {
#ifdef F3D_LAB  // timing builds that skip parts of the work (WRONG results): only in lib/lab/libf3d_hip.so (make lab, tools/kbench.py
                // --ablate); the shipped library has no switch that changes a result and no ABL != 0 instantiation
    static const int abl = std::getenv("F3D_ABLATE8") ? std::atoi(std::getenv("F3D_ABLATE8")) : 0;
    if constexpr (TY == 12) {
      if (abl == 16) return go(k_pair8<MODE, TY, 16>);
    }
    if constexpr (MODE == PAIR_SS) {
      if (abl == 1) return go(k_pair8<MODE, TY, 1>);
      if (abl == 2) return go(k_pair8<MODE, TY, 2>);
      if (abl == 3) return go(k_pair8<MODE, TY, 3>);
      if (abl == 4) return go(k_pair8<MODE, TY, 4>);
      if (abl == 8) return go(k_pair8<MODE, TY, 8>);
      if (abl == 40) return go(k_pair8<MODE, TY, 40>);
      if constexpr (TY <= 8) {
        if (abl == 64) return go(k_pair8<MODE, TY, 64>);
      }
    }
#endif
    go(k_pair8<MODE, TY, 0>);
  }
}

// thin volume: the y-marching build whose tile holds all z planes
template <int MODE>
bool launch_pair8_ymarch(const PairArgs& a, const F3dGeo& g, int force_zchunk, int xcd_remap)
{
  // a tile that holds exactly the planes of the volume needs no halo rows (TIGHT): 4 and 5 planes, the depths of BASELINE config 3.
  // F3D_PAIR8_TIGHT=0 keeps the halo rows (A/B timing; read per call like the march switch)
  const char* e = std::getenv("F3D_PAIR8_TIGHT");
  const bool tight = !(e && e[0] == '0');
  switch (pair8_ymarch_rows(g)) {
    case 4:
      if (tight && g.D == 4) launch_pair8<MODE, 4, false, true, true>(a, g, force_zchunk, xcd_remap);
      else launch_pair8<MODE, 4, false, true>(a, g, force_zchunk, xcd_remap);
      return true;
    case 5:
      if (tight && g.D == 5) launch_pair8<MODE, 5, false, true, true>(a, g, force_zchunk, xcd_remap);
      else launch_pair8<MODE, 5, false, true>(a, g, force_zchunk, xcd_remap);
      return true;
    case 8: launch_pair8<MODE, 8, false, true>(a, g, force_zchunk, xcd_remap); return true;
    default: return false;
  }
}
