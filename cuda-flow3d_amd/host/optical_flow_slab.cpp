#include "optical_flow_slab.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>
#include <utility>

#include "common_utils.h"
#include "hip_utils.h"

OpticalFlowSlab::OpticalFlowSlab(int n_ranks, std::vector<int> local_ranks, int halo_capacity)
    : OpticalFlowBase("Optical Flow z-slab Multi GPU"), n_ranks_(n_ranks), local_ranks_(std::move(local_ranks)), halo_(halo_capacity)
{
  // slabs at least this thick run the overlapped order (0 = never); thinner ones have too little interior to hide anything
  const char* e = std::getenv("F3D_OVERLAP_MIN_PLANES");
  overlap_min_planes_ = e ? std::atoi(e) : 32;
  // thin slabs of small levels: several outer iterations per exchange (0 = the rule in Pyramid(), n = force n)
  if (const char* f = std::getenv("F3D_SLAB_OUTER_PER_EXCHANGE")) forced_outer_per_exchange_ = std::atoi(f);
  if (const char* f = std::getenv("F3D_SLAB_SMALL_LEVEL_VOXELS")) small_level_voxels_ = std::atof(f);
  if (const char* f = std::getenv("F3D_SLAB_FUSED_PHI_KSI")) fused_weights_ = std::atoi(f) != 0;
  if (const char* f = std::getenv("F3D_SLAB_EXCHANGE")) exchange_per_stage_ = std::string(f) == "stage";
}

OpticalFlowSlab::~OpticalFlowSlab() { Destroy(); }

bool OpticalFlowSlab::Check(int status)
{
  if (CheckDeviceError(status)) failed_ = true;
  return status == 0;
}

bool OpticalFlowSlab::Initialize(const DataSize4& data_size)
{
  if (n_ranks_ < 1 || local_ranks_.empty()) return false;
  for (int r : local_ranks_)
    if (r < 0 || r >= n_ranks_) return false;
  if (local_ranks_.size() != 1 && static_cast<int>(local_ranks_.size()) != n_ranks_) {
    std::printf("'%s': a process computes either one rank (RCCL) or all of them (one-GPU rehearsal).\n", GetName());
    return false;
  }
  // the in-process exchange looks a peer up by its rank number: the list must be exactly 0, 1, ..., n_ranks - 1
  if (local_ranks_.size() > 1)
    for (size_t i = 0; i < local_ranks_.size(); ++i)
      if (local_ranks_[i] != static_cast<int>(i)) {
        std::printf("'%s': with every rank in one process the rank list must be 0 .. %d in order.\n", GetName(), n_ranks_ - 1);
        return false;
      }
  full_size_ = data_size;
  full_size_.pitch = 0;
  const size_t max_planes = (data_size.depth + n_ranks_ - 1) / n_ranks_ + 1;
  local_container_ = {data_size.width, data_size.height, max_planes + 2 * static_cast<size_t>(halo_), 0};

  const size_t rows = local_container_.height * local_container_.depth;
  for (int r : local_ranks_) {
    Local l;
    l.rank = r;
    for (int i = 0; i < kRoles; ++i) l.buf[i] = 0;
    for (int i = 0; i < kRequiredRoles; ++i) {
      size_t pitch = 0;
      if (!Check(f3d_alloc_pitched(&l.buf[i], &pitch, data_size.width * sizeof(float), rows))) return false;
      if (local_container_.pitch && pitch != local_container_.pitch) return false;
      local_container_.pitch = pitch;
    }
    // four more for the frame derivatives the fused launches read (round 4; F3D_FRAME_DERIVATIVES=0 does without): optional -- without
    // them (no room, another pitch) the fused launches form the derivatives from the frames themselves, same bits
    if (FrameDerivativesEnabled() && FusedSweepsEnabled()) {
      bool all = true;
      for (int i = FDX; i <= FDT && all; ++i) {
        size_t pitch = 0;
        all = f3d_alloc_pitched(&l.buf[i], &pitch, data_size.width * sizeof(float), rows) == 0 && pitch == local_container_.pitch;
      }
      if (!all)
        for (int i = FDX; i <= FDT; ++i) {
          if (l.buf[i]) f3d_free(l.buf[i]);
          l.buf[i] = 0;
        }
    }
    locals_.push_back(l);
  }
  // staging for the RCCL path: both halos of the deepest exchange (5 fields, halo_ planes each side)
  stage_floats_ = static_cast<size_t>(2 * halo_) * 5 * data_size.width * data_size.height;
  if (local_ranks_.size() == 1 && n_ranks_ > 1) {
    size_t p = 0;
    if (!Check(f3d_alloc_pitched(&stage_send_, &p, stage_floats_ * sizeof(float), 1))) return false;
    if (!Check(f3d_alloc_pitched(&stage_recv_, &p, stage_floats_ * sizeof(float), 1))) return false;
  }
  f3d_size4 c = {local_container_.width, local_container_.height, local_container_.depth, local_container_.pitch};
  if (!Check(f3d_set_container(&c))) return false;
  initialized_ = true;
  return true;
}

void OpticalFlowSlab::Destroy()
{
  for (Local& l : locals_)
    for (DevicePtr& p : l.buf) {
      if (p) f3d_free(p);
      p = 0;
    }
  locals_.clear();
  if (stage_send_) f3d_free(stage_send_);
  if (stage_recv_) f3d_free(stage_recv_);
  stage_send_ = stage_recv_ = 0;
  if (f1_wide_) f3d_free(f1_wide_);
  f1_wide_ = 0;
  wide_planes_ = 0;
  initialized_ = false;
}

// ---- warp reach beyond the halo room ------------------------------------------------------------------------------------------
// The wide buffers hold, per local rank, planes [own.lo - need, own.lo - need + wide_planes_) of ONE field of the level, laid out
// like a local container (same pitch and height, only deeper).
bool OpticalFlowSlab::GatherPlanes(int depth, size_t width, size_t height, Role src_role, int need)
{
  const size_t plane_bytes = local_container_.pitch * local_container_.height;
  auto wide_of = [&](size_t local_index) { return f1_wide_ + static_cast<DevicePtr>(local_index * wide_planes_ * plane_bytes); };
  // pack / unpack / plane copies check their plane numbers against the current container: make it as deep as the deeper of the two
  f3d_size4 deep = {local_container_.width, local_container_.height, std::max(local_container_.depth, wide_planes_), local_container_.pitch};
  f3d_size4 normal = {local_container_.width, local_container_.height, local_container_.depth, local_container_.pitch};
  if (!Check(f3d_set_container(&deep))) return false;
  bool ok = true;
  for (size_t i = 0; ok && i < locals_.size(); ++i) {  // own planes: local container -> wide buffer
    const PlaneRange own = OwnedPlanes(depth, locals_[i].rank, n_ranks_);
    if (own.empty()) continue;
    ok = Check(f3d_copy_planes(wide_of(i), need, locals_[i].buf[src_role], halo_, own.size(), width, height));
  }
  if (ok && locals_.size() > 1) {  // every rank lives here
    for (size_t i = 0; ok && i < locals_.size(); ++i) {
      const Local& me = locals_[i];
      const int wide_base = OwnedPlanes(depth, me.rank, n_ranks_).lo - need;
      for (const HaloTransfer& t : PlanHaloExchange(depth, me.rank, n_ranks_, need, need)) {
        if (t.recv.empty()) continue;
        const Local& peer = locals_[t.peer];
        ok = Check(f3d_copy_planes(wide_of(i), t.recv.lo - wide_base, peer.buf[src_role], t.recv.lo - ZBase(depth, peer.rank), t.recv.size(),
                                   width, height));
        if (!ok) break;
      }
    }
  } else if (ok && n_ranks_ > 1) {  // one rank per process: pack own planes, grouped send / recv, unpack into the wide buffer
    const size_t plane = width * height;
    Local& me = locals_[0];
    const int my_base = ZBase(depth, me.rank);
    const int wide_base = OwnedPlanes(depth, me.rank, n_ranks_).lo - need;
    std::vector<size_t> s_off, s_cnt, r_off, r_cnt;
    std::vector<int> peers;
    size_t s_pos = 0, r_pos = 0;
    struct Seg { int plane, count; size_t off; };
    std::vector<Seg> packs, unpacks;
    for (const HaloTransfer& t : PlanHaloExchange(depth, me.rank, n_ranks_, need, need)) {
      peers.push_back(t.peer);
      s_off.push_back(s_pos);
      r_off.push_back(r_pos);
      if (!t.send.empty()) {
        packs.push_back({t.send.lo - my_base, t.send.size(), s_pos});
        s_pos += static_cast<size_t>(t.send.size()) * plane;
      }
      if (!t.recv.empty()) {
        unpacks.push_back({t.recv.lo - wide_base, t.recv.size(), r_pos});
        r_pos += static_cast<size_t>(t.recv.size()) * plane;
      }
      s_cnt.push_back(s_pos - s_off.back());
      r_cnt.push_back(r_pos - r_off.back());
    }
    // The decision to give up must be COLLECTIVE: what a rank packs and unpacks differs by rank (edge ranks move half of what middle
    // ranks do, multi-hop plans vary further), and a rank that returned here while the others entered the grouped send / recv would
    // leave them blocked.  Every rank therefore prices the plan of EVERY rank and all fail on the largest (advisor, round 3).
    size_t worst = 0;
    for (int r = 0; r < n_ranks_; ++r) {
      size_t s_all = 0, r_all = 0;
      for (const HaloTransfer& t : PlanHaloExchange(depth, r, n_ranks_, need, need)) {
        s_all += static_cast<size_t>(t.send.size()) * plane;
        r_all += static_cast<size_t>(t.recv.size()) * plane;
      }
      worst = std::max(worst, std::max(s_all, r_all));
    }
    if (worst > stage_floats_) {
      std::printf("'%s': staging buffer too small to gather %d planes of frame 1 on either side.\n", GetName(), need);
      failed_ = true;
      ok = false;
    }
    for (size_t k = 0; ok && k < packs.size(); ++k)
      ok = Check(f3d_pack_planes(me.buf[src_role], packs[k].plane, packs[k].count, width, height, stage_send_, packs[k].off));
    ok = ok && Check(f3d_comm_sendrecv(stage_send_, s_off.data(), s_cnt.data(), stage_recv_, r_off.data(), r_cnt.data(), peers.data(),
                                       static_cast<int>(peers.size())));
    for (size_t k = 0; ok && k < unpacks.size(); ++k)
      ok = Check(f3d_unpack_planes(wide_of(0), unpacks[k].plane, unpacks[k].count, width, height, stage_recv_, unpacks[k].off));
  }
  return Check(f3d_set_container(&normal)) && ok;
}

bool OpticalFlowSlab::WarpWithGatheredFrame(int D, size_t W, size_t H, float hx, float hy, float hz, int wide, int need)
{
  int max_slab = 0;
  for (int r = 0; r < n_ranks_; ++r) max_slab = std::max(max_slab, OwnedPlanes(D, r, n_ranks_).size());
  const size_t planes = static_cast<size_t>(max_slab) + 2 * static_cast<size_t>(need);
  const size_t plane_bytes = local_container_.pitch * local_container_.height;
  if (planes > wide_planes_) {  // grows with the largest reach seen; released in Destroy()  (`planes` is the same on every rank)
    if (f1_wide_) f3d_free(f1_wide_);
    f1_wide_ = 0;
    wide_planes_ = 0;
    size_t pitch = 0;
    const bool got = f3d_alloc_pitched(&f1_wide_, &pitch, local_container_.width * sizeof(float),
                                       local_container_.height * planes * locals_.size()) == 0 && pitch == local_container_.pitch;
    // whether the memory is there can differ by rank: agree on it before anybody enters the gather's grouped send / recv
    float nobody_failed = got ? 0.f : 1.f;
    if (locals_.size() == 1 && n_ranks_ > 1 && !Check(f3d_comm_allreduce_max_f32(&nobody_failed))) return false;
    if (nobody_failed != 0.f) {
      if (got) f3d_free(f1_wide_);
      f1_wide_ = 0;
      std::printf("'%s': no room for %zu planes of frame 1 (warp reach %d beyond the halo capacity %d)%s.\n", GetName(), planes, need - wide,
                  halo_, got ? " on another rank" : "");
      failed_ = true;
      return false;
    }
    wide_planes_ = planes;
  }
  if (!GatherPlanes(D, W, H, F1R, need)) return false;
  // the warp on the wide container: frame 1 from it, the other operands through pointers that make THEIR first plane
  // (own.lo - halo_) appear at the plane number it has in the wide geometry (whose first plane is own.lo - need)
  f3d_size4 deep = {local_container_.width, local_container_.height, wide_planes_, local_container_.pitch};
  f3d_size4 normal = {local_container_.width, local_container_.height, local_container_.depth, local_container_.pitch};
  if (!Check(f3d_set_container(&deep))) return false;
  bool ok = true;
  // (need may be smaller than halo_ on a level shallower than the reach: the shift is then negative; two's complement does it)
  const DevicePtr shift = static_cast<DevicePtr>(static_cast<long long>(need - halo_) * static_cast<long long>(plane_bytes));
  for (size_t i = 0; ok && i < locals_.size(); ++i) {
    Local& l = locals_[i];
    const PlaneRange own = OwnedPlanes(D, l.rank, n_ranks_);
    if (own.empty()) continue;
    f3d_slab win;
    win.z_base = own.lo - need;
    win.z_lo = std::max(0, own.lo - wide);
    win.z_hi = std::min(D, own.hi + wide);
    const DevicePtr f1 = f1_wide_ + static_cast<DevicePtr>(i * wide_planes_ * plane_bytes);
    ok = Check(f3d_warp(l.buf[F0R] - shift, f1, l.buf[FU] - shift, l.buf[FV] - shift, l.buf[FW] - shift, W, H, D, hx, hy, hz,
                        l.buf[TMP] - shift, &win));
    if (ok) std::swap(l.buf[F1R], l.buf[TMP]);
  }
  ++wide_warps_;
  return Check(f3d_set_container(&normal)) && ok;
}

f3d_slab OpticalFlowSlab::Window(int depth, int rank, int grow_lo, int grow_hi) const
{
  const PlaneRange own = OwnedPlanes(depth, rank, n_ranks_);
  f3d_slab s;
  s.z_base = own.lo - halo_;
  if (own.empty()) {
    s.z_lo = s.z_hi = std::max(0, own.lo);
    if (s.z_lo < s.z_base) s.z_base = s.z_lo;
    return s;
  }
  s.z_lo = std::max(0, own.lo - grow_lo);
  s.z_hi = std::min(depth, own.hi + grow_hi);
  return s;
}

bool OpticalFlowSlab::Exchange(int depth, size_t width, size_t height, const std::vector<Role>& roles, int need_lo, int need_hi)
{
  if (n_ranks_ == 1 || (need_lo <= 0 && need_hi <= 0)) return true;
  if (need_lo > halo_ || need_hi > halo_) {
    std::printf("'%s': a halo of %d/%d planes exceeds the capacity of %d; raise halo_capacity.\n", GetName(), need_lo, need_hi, halo_);
    failed_ = true;
    return false;
  }
  if (locals_.size() > 1) {  // every rank lives here: copy planes between their containers -- the whole exchange in a few launches
    std::vector<f3d_devptr> dst, src;
    std::vector<int> dst_plane, src_plane, count;
    for (Local& me : locals_) {
      const int my_base = ZBase(depth, me.rank);
      for (const HaloTransfer& t : PlanHaloExchange(depth, me.rank, n_ranks_, need_lo, need_hi)) {
        if (t.recv.empty()) continue;
        const Local& peer = locals_[t.peer];
        const int peer_base = ZBase(depth, peer.rank);
        for (Role role : roles) {
          dst.push_back(me.buf[role]);
          dst_plane.push_back(t.recv.lo - my_base);
          src.push_back(peer.buf[role]);
          src_plane.push_back(t.recv.lo - peer_base);
          count.push_back(t.recv.size());
        }
      }
    }
    // sources are owned planes, destinations halo planes: no segment reads what another one writes
    return Check(f3d_copy_plane_segments(dst.data(), dst_plane.data(), src.data(), src_plane.data(), count.data(),
                                         static_cast<int>(dst.size()), width, height));
  }
  // one rank per process: pack -> grouped send/recv -> unpack (f3d_comm_mark: the interval bench.py reports as microseconds per
  // exchange when it has switched the timing on; nothing otherwise)
  f3d_comm_mark(0, 0);
  const bool ok = ExchangeBegin(depth, width, height, roles, roles, need_lo, need_hi) && ExchangeEnd(width, height);
  f3d_comm_mark(1, 0);
  return ok;
}

bool OpticalFlowSlab::ExchangeBegin(int depth, size_t width, size_t height, const std::vector<Role>& send_roles,
                                    const std::vector<Role>& recv_roles, int need_lo, int need_hi)
{
  const size_t plane = width * height;
  Local& me = locals_[0];
  const int my_base = ZBase(depth, me.rank);
  const std::vector<HaloTransfer> plan = PlanHaloExchange(depth, me.rank, n_ranks_, need_lo, need_hi);
  std::vector<size_t> s_off, s_cnt, r_off, r_cnt;
  std::vector<int> peers;
  // one pack launch, one grouped send/recv, one unpack launch per exchange
  std::vector<f3d_devptr> pk_field;
  std::vector<int> pk_plane, pk_count;
  std::vector<size_t> pk_off;
  unpack_ = Unpack();
  size_t s_pos = 0, r_pos = 0;
  for (const HaloTransfer& t : plan) {
    peers.push_back(t.peer);
    s_off.push_back(s_pos);
    r_off.push_back(r_pos);
    for (size_t k = 0; k < send_roles.size(); ++k) {
      if (!t.send.empty()) {
        pk_field.push_back(me.buf[send_roles[k]]);
        pk_plane.push_back(t.send.lo - my_base);
        pk_count.push_back(t.send.size());
        pk_off.push_back(s_pos);
        s_pos += static_cast<size_t>(t.send.size()) * plane;
      }
      if (!t.recv.empty()) {
        unpack_.field.push_back(me.buf[recv_roles[k]]);
        unpack_.plane.push_back(t.recv.lo - my_base);
        unpack_.count.push_back(t.recv.size());
        unpack_.offset.push_back(r_pos);
        r_pos += static_cast<size_t>(t.recv.size()) * plane;
      }
    }
    s_cnt.push_back(s_pos - s_off.back());
    r_cnt.push_back(r_pos - r_off.back());
  }
  if (s_pos > stage_floats_ || r_pos > stage_floats_) {
    std::printf("'%s': staging buffer too small for this exchange.\n", GetName());
    failed_ = true;
    return false;
  }
  constexpr size_t kBatch = 32;  // segments per launch (f3d_pack_segments limit)
  for (size_t i = 0; i < pk_field.size(); i += kBatch) {
    const int n = static_cast<int>(std::min(kBatch, pk_field.size() - i));
    if (!Check(f3d_pack_segments(&pk_field[i], &pk_plane[i], &pk_count[i], &pk_off[i], n, width, height, stage_send_))) return false;
  }
  return Check(f3d_comm_sendrecv_begin(stage_send_, s_off.data(), s_cnt.data(), stage_recv_, r_off.data(), r_cnt.data(), peers.data(),
                                       static_cast<int>(peers.size())));
}

bool OpticalFlowSlab::ExchangeEnd(size_t width, size_t height)
{
  if (!Check(f3d_comm_sendrecv_end())) return false;
  constexpr size_t kBatch = 32;
  for (size_t i = 0; i < unpack_.field.size(); i += kBatch) {
    const int n = static_cast<int>(std::min(kBatch, unpack_.field.size() - i));
    if (!Check(f3d_unpack_segments(&unpack_.field[i], &unpack_.plane[i], &unpack_.count[i], &unpack_.offset[i], n, width, height,
                                   stage_recv_)))
      return false;
  }
  return true;
}

// The weights of an outer iteration on planes [lo, hi): whatever the fused last sweep of the previous iteration has not
// written already.  With an exchange in between that is the zone next to each neighbour (the plane whose weights need the
// neighbour's new increments, and the halo planes); inside a group of iterations without exchange it is nothing.
bool OpticalFlowSlab::CompleteWeights(Local& l, int lo, int hi, int D, size_t W, size_t H, float hx, float hy, float hz,
                                      float equation_smoothness, float equation_data)
{
  lo = std::max(0, lo);
  hi = std::min(D, hi);
  auto run = [&](int from, int to) {
    if (to <= from) return true;
    f3d_slab s;
    s.z_base = ZBase(D, l.rank);
    s.z_lo = from;
    s.z_hi = to;
    return Check(f3d_phi_ksi(l.buf[F0R], l.buf[F1R], l.buf[FU], l.buf[FV], l.buf[FW], l.buf[DU], l.buf[DV], l.buf[DW], W, H, D, hx, hy, hz,
                             equation_smoothness, equation_data, l.buf[PHI], l.buf[KSI], &s));
  };
  const bool some = l.weights_hi > l.weights_lo;
  const int have_lo = l.weights_lo, have_hi = l.weights_hi;
  l.weights_lo = l.weights_hi = 0;
  if (!some) return run(lo, hi);
  const int a_hi = std::min(hi, have_lo), b_lo = std::max(lo, have_hi);
  if (a_hi > lo && hi > b_lo) {  // both zones: one launch
    f3d_slab za, zb;
    za.z_base = zb.z_base = ZBase(D, l.rank);
    za.z_lo = lo;
    za.z_hi = a_hi;
    zb.z_lo = b_lo;
    zb.z_hi = hi;
    return Check(f3d_phi_ksi_zones(l.buf[F0R], l.buf[F1R], l.buf[FU], l.buf[FV], l.buf[FW], l.buf[DU], l.buf[DV], l.buf[DW], W, H, D, hx,
                                   hy, hz, equation_smoothness, equation_data, l.buf[PHI], l.buf[KSI], &za, &zb));
  }
  return run(lo, a_hi) && run(b_lo, hi);
}

// fx, fy, fz, ft of the level on every plane a fused launch of this level computes a stage 1 on: the frames are valid on the slab
// widened by `valid_halo` planes (the warp's window), a z derivative needs one plane more on either side, so the derivatives are
// made on the slab widened by valid_halo - 1 (clipped at the faces of the volume, where the mirror plane is the rank's own)
bool OpticalFlowSlab::FrameDerivatives(Local& l, int D, size_t W, size_t H, float hx, float hy, float hz, int valid_halo)
{
  l.derivatives = false;
  if (!l.buf[FDX] || !FusedSweepsEnabled() || local_container_.pitch % 256 != 0) return true;
  const PlaneRange own = OwnedPlanes(D, l.rank, n_ranks_);
  if (own.empty() || valid_halo < 1) return true;
  f3d_slab win;
  win.z_base = ZBase(D, l.rank);
  win.z_lo = own.lo - valid_halo <= 0 ? 0 : own.lo - valid_halo + 1;
  win.z_hi = own.hi + valid_halo >= D ? D : own.hi + valid_halo - 1;
  if (!Check(f3d_frame_derivatives(l.buf[F0R], l.buf[F1R], W, H, D, hx, hy, hz, l.buf[FDX], l.buf[FDY], l.buf[FDZ], l.buf[FDT], &win)))
    return false;
  l.derivatives = true;
  return true;
}

bool OpticalFlowSlab::Sweeps(Local& l, bool pair, const Role* in, const Role* out, size_t W, size_t H, int D, float hx, float hy, float hz,
                             float equation_alpha, const f3d_slab& win)
{
  if (pair && l.derivatives)
    return Check(f3d_solve_sweep2_fd(l.buf[FDX], l.buf[FDY], l.buf[FDZ], l.buf[FDT], l.buf[FU], l.buf[FV], l.buf[FW], l.buf[in[0]],
                                     l.buf[in[1]], l.buf[in[2]], l.buf[PHI], l.buf[KSI], W, H, D, hx, hy, hz, equation_alpha, l.buf[out[0]],
                                     l.buf[out[1]], l.buf[out[2]], &win));
  auto* fn = pair ? f3d_solve_sweep2 : f3d_solve_sweep;
  return Check(fn(l.buf[F0R], l.buf[F1R], l.buf[FU], l.buf[FV], l.buf[FW], l.buf[in[0]], l.buf[in[1]], l.buf[in[2]], l.buf[PHI], l.buf[KSI],
                  W, H, D, hx, hy, hz, equation_alpha, l.buf[out[0]], l.buf[out[1]], l.buf[out[2]], &win));
}

bool OpticalFlowSlab::SweepAndNextWeights(Local& l, const Role (&in)[3], const Role (&out)[3], int sweep_lo, int sweep_hi, int D,
                                          size_t W, size_t H, float hx, float hy, float hz, float equation_alpha,
                                          float equation_smoothness, float equation_data, bool& launched)
{
  launched = false;
  // the weights of a plane need the new increments of both z neighbours: inside the volume the first and the last plane of
  // the range have to wait for a neighbour's (at a face the mirrored plane is the rank's own)
  const bool shrink_lo = sweep_lo > 0, shrink_hi = sweep_hi < D;
  f3d_slab s;
  s.z_base = ZBase(D, l.rank);
  s.z_lo = sweep_lo + (shrink_lo ? 1 : 0);
  s.z_hi = sweep_hi - (shrink_hi ? 1 : 0);
  if (s.z_hi - s.z_lo < 1) return true;
  const int status =
      l.derivatives
          ? f3d_solve_sweep_phi_ksi_edges_fd(l.buf[FDX], l.buf[FDY], l.buf[FDZ], l.buf[FDT], l.buf[FU], l.buf[FV], l.buf[FW], l.buf[in[0]],
                                             l.buf[in[1]], l.buf[in[2]], l.buf[PHI], l.buf[KSI], W, H, D, hx, hy, hz, equation_alpha,
                                             equation_smoothness, equation_data, l.buf[out[0]], l.buf[out[1]], l.buf[out[2]], l.buf[PHI2],
                                             l.buf[KSI2], &s, shrink_lo ? 1 : 0, shrink_hi ? 1 : 0)
          : f3d_solve_sweep_phi_ksi_edges(l.buf[F0R], l.buf[F1R], l.buf[FU], l.buf[FV], l.buf[FW], l.buf[in[0]], l.buf[in[1]], l.buf[in[2]],
                                          l.buf[PHI], l.buf[KSI], W, H, D, hx, hy, hz, equation_alpha, equation_smoothness, equation_data,
                                          l.buf[out[0]], l.buf[out[1]], l.buf[out[2]], l.buf[PHI2], l.buf[KSI2], &s, shrink_lo ? 1 : 0,
                                          shrink_hi ? 1 : 0);
  if (!Check(status))
    return false;
  std::swap(l.buf[PHI], l.buf[PHI2]);
  std::swap(l.buf[KSI], l.buf[KSI2]);
  l.weights_lo = s.z_lo;
  l.weights_hi = s.z_hi;
  launched = true;
  return true;
}

// One outer iteration (phi/ksi + K sweeps) of a rank that has neighbours, ordered so that the exchange of the new
// increments runs beside most of the arithmetic.  With own planes [a, b), H = K + 1 planes travelling each way and R_s
// sweeps still to come after stage s (a stage = a fused pair or a single sweep):
//   * the LOW ZONE of stage s is [a - R_s, a + H + R_s), the HIGH ZONE [b - H - R_s, b + R_s): the cone of dependence of
//     the H planes a neighbour needs.  Both zones go through phi/ksi and all stages first; the last stage writes its H
//     planes into the edge containers (EDU, EDV, EDW), from which they are packed and sent;
//   * the INTERIOR [a + H + R_s, b - H - R_s) follows while the transfer is in flight.  Stage s of the interior reads
//     what stage s-1 wrote in the zones and in the interior alike (same containers).  The ping-pong containers never
//     clash: a zone's stage-s output ends exactly where the interior's stage-(s-1) input begins -- except for the last
//     stage, whose zone output would overwrite planes the interior's last-but-one stage still reads; hence the edge
//     containers, copied into place at the end;
//   * finally the received halos are unpacked.  Every voxel is computed once, by the same kernels on the same inputs as in
//     the plain order, so the bits do not change (tests/test_gpu_slab_procs.py);
//   * the interior's last sweep also writes the weights of the NEXT outer iteration for its planes (one launch, see
//     SweepAndNextWeights); the next call then computes phi/ksi only for the two zones and the halo planes.
bool OpticalFlowSlab::SweepsOverlapped(Local& l, int D, size_t W, size_t H, int K, float hx, float hy, float hz,
                                       float equation_alpha, float equation_smoothness, float equation_data)
{
  const PlaneRange own = OwnedPlanes(D, l.rank, n_ranks_);
  const int a = own.lo, b = own.hi, Hs = K + 1;
  const bool has_lo = a > 0, has_hi = b < D;
  struct Stage {
    bool pair;
    int rest;  // sweeps still to come after this stage
  };
  std::vector<Stage> stages;
  for (int j = 0; j < K;) {
    const bool pair = FusedSweepsEnabled() && j + 2 <= K;
    j += pair ? 2 : 1;
    stages.push_back({pair, K - j});
  }
  auto slab = [&](int lo, int hi) {
    f3d_slab s;
    s.z_base = a - halo_;
    s.z_lo = std::max(0, lo);
    s.z_hi = std::min(D, hi);
    if (s.z_hi < s.z_lo) s.z_hi = s.z_lo;
    return s;
  };
  enum Zone { LOW, HIGH, INNER };
  auto zone = [&](Zone z, int grow) {
    const int lo_edge = a + Hs + grow, hi_edge = b - Hs - grow;
    if (z == LOW) return slab(a - grow, lo_edge);
    if (z == HIGH) return slab(hi_edge, b + grow);
    return slab(has_lo ? lo_edge : a - grow, has_hi ? hi_edge : b + grow);
  };
  const Role cur[3] = {DU, DV, DW}, tmp[3] = {TDU, TDV, TDW}, edge[3] = {EDU, EDV, EDW};
  auto phi = [&](const f3d_slab& win) {
    return Check(f3d_phi_ksi(l.buf[F0R], l.buf[F1R], l.buf[FU], l.buf[FV], l.buf[FW], l.buf[DU], l.buf[DV], l.buf[DW], W, H, D, hx, hy, hz,
                             equation_smoothness, equation_data, l.buf[PHI], l.buf[KSI], &win));
  };
  auto stage = [&](size_t s, const f3d_slab& win, bool to_edge) {
    const Role* in = (s % 2 == 0) ? cur : tmp;
    const Role* out = to_edge ? edge : ((s % 2 == 0) ? tmp : cur);
    return Sweeps(l, stages[s].pair, in, out, W, H, D, hx, hy, hz, equation_alpha, win);
  };
  const size_t last = stages.size() - 1;
  const Role* final_out = (last % 2 == 0) ? tmp : cur;
  // the last stage of the interior can take the next weights along when it is a single sweep (an exchange always follows here)
  const bool fuse_last = fused_weights_ && FusedSweepsEnabled() && FusedPhiKsiEnabled() && !stages[last].pair;
  const bool had_weights = l.weights_hi > l.weights_lo;
  if (fuse_last || had_weights) {
    if (!CompleteWeights(l, a - K, b + K, D, W, H, hx, hy, hz, equation_smoothness, equation_data)) return false;
  }
  const bool weights_done = fuse_last || had_weights;
  for (Zone z : {LOW, HIGH}) {
    if ((z == LOW && !has_lo) || (z == HIGH && !has_hi)) continue;
    if (!weights_done && !phi(zone(z, K))) return false;
    for (size_t s = 0; s <= last; ++s)
      if (!stage(s, zone(z, stages[s].rest), s == last)) return false;
  }
  f3d_comm_mark(0, 1);   // (timing runs only) an exchange whose transfer runs beside the interior's launches
  if (!ExchangeBegin(D, W, H, {edge[0], edge[1], edge[2]}, {final_out[0], final_out[1], final_out[2]}, Hs, Hs)) return false;
  if (!weights_done && !phi(zone(INNER, K))) return false;
  for (size_t s = 0; s <= last; ++s) {
    const f3d_slab win = zone(INNER, stages[s].rest);
    if (s == last && fuse_last) {
      const Role* in = (s % 2 == 0) ? cur : tmp;
      bool launched = false;
      if (!SweepAndNextWeights(l, {in[0], in[1], in[2]}, {final_out[0], final_out[1], final_out[2]}, win.z_lo, win.z_hi, D, W, H, hx, hy,
                               hz, equation_alpha, equation_smoothness, equation_data, launched))
        return false;
      if (launched) continue;
    }
    if (!stage(s, win, false)) return false;
  }
  const int base = a - halo_;
  for (int k = 0; k < 3; ++k) {
    if (has_lo && !Check(f3d_copy_planes(l.buf[final_out[k]], a - base, l.buf[edge[k]], a - base, Hs, W, H))) return false;
    if (has_hi && !Check(f3d_copy_planes(l.buf[final_out[k]], b - Hs - base, l.buf[edge[k]], b - Hs - base, Hs, W, H))) return false;
  }
  if (!ExchangeEnd(W, H)) return false;
  f3d_comm_mark(1, 1);
  if (final_out == tmp) {
    std::swap(l.buf[DU], l.buf[TDU]);
    std::swap(l.buf[DV], l.buf[TDV]);
    std::swap(l.buf[DW], l.buf[TDW]);
  }
  return true;
}

void OpticalFlowSlab::UploadFrames(Data3D& frame_0, Data3D& frame_1)
{
  if (!IsInitialized()) return;
  const int D = static_cast<int>(full_size_.depth);
  const size_t plane = full_size_.width * full_size_.height;
  for (Local& l : locals_) {
    // raw planes [own.lo - halo, own.hi + halo) of the volume: enough for the blur's z taps
    const PlaneRange own = OwnedPlanes(D, l.rank, n_ranks_);
    if (own.empty()) continue;
    const int lo = std::max(0, own.lo - halo_), hi = std::min(D, own.hi + halo_);
    const int base = ZBase(D, l.rank);
    Check(f3d_copy3d_h2d(l.buf[RAW0], local_container_.pitch, local_container_.height, lo - base,
                         frame_0.DataPtr() + static_cast<size_t>(lo) * plane, full_size_.width, full_size_.height, hi - lo));
    Check(f3d_copy3d_h2d(l.buf[RAW1], local_container_.pitch, local_container_.height, lo - base,
                         frame_1.DataPtr() + static_cast<size_t>(lo) * plane, full_size_.width, full_size_.height, hi - lo));
  }
}

void OpticalFlowSlab::DownloadFlow(Data3D& flow_u, Data3D& flow_v, Data3D& flow_w)
{
  const int D = static_cast<int>(full_size_.depth);
  const size_t plane = full_size_.width * full_size_.height;
  Data3D* out[3] = {&flow_u, &flow_v, &flow_w};
  const Role roles[3] = {FU, FV, FW};
  for (Local& l : locals_) {
    const PlaneRange own = OwnedPlanes(D, l.rank, n_ranks_);
    if (own.empty()) continue;
    for (int i = 0; i < 3; ++i)
      Check(f3d_copy3d_d2h(out[i]->DataPtr() + static_cast<size_t>(own.lo) * plane, full_size_.width, full_size_.height,
                           own.size(), l.buf[roles[i]], local_container_.pitch, local_container_.height, halo_));
  }
}

void OpticalFlowSlab::ComputeFlow(Data3D& frame_0, Data3D& frame_1, Data3D& flow_u, Data3D& flow_v, Data3D& flow_w,
                                  OperationParameters& params)
{
  if (!IsInitialized()) return;
  UploadFrames(frame_0, frame_1);
  if (ComputeResident(params)) DownloadFlow(flow_u, flow_v, flow_w);
}

bool OpticalFlowSlab::ComputeResident(OperationParameters& params)
{
  if (!IsInitialized()) return false;
  failed_ = false;
  f3d_event ev_start = nullptr, ev_stop = nullptr;
  Check(f3d_event_create(&ev_start));
  Check(f3d_event_create(&ev_stop));
  Check(f3d_event_record(ev_start));
  const bool ok = Pyramid(params) && !failed_;
  float ms = 0.f;
  Check(f3d_event_record(ev_stop));
  Check(f3d_event_sync(ev_stop));
  Check(f3d_event_elapsed_ms(&ms, ev_start, ev_stop));
  last_device_seconds_ = ms / 1000.f;
  f3d_event_destroy(ev_start);
  f3d_event_destroy(ev_stop);
  return ok;
}

bool OpticalFlowSlab::Pyramid(OperationParameters& params)
{
  overlapped_iterations_ = 0;
  batched_exchanges_ = 0;
  wide_warps_ = 0;
  stage_exchanges_ = 0;
  size_t warp_levels_count, outer_iterations_count, inner_iterations_count, median_radius;
  float warp_scale_factor, equation_alpha, equation_smoothness, equation_data, gaussian_sigma;
  GET_PARAM_OR_RETURN_VALUE(params, size_t, warp_levels_count, "warp_levels_count", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, warp_scale_factor, "warp_scale_factor", false);
  GET_PARAM_OR_RETURN_VALUE(params, size_t, outer_iterations_count, "outer_iterations_count", false);
  GET_PARAM_OR_RETURN_VALUE(params, size_t, inner_iterations_count, "inner_iterations_count", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, equation_alpha, "equation_alpha", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, equation_smoothness, "equation_smoothness", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, equation_data, "equation_data", false);
  GET_PARAM_OR_RETURN_VALUE(params, size_t, median_radius, "median_radius", false);
  GET_PARAM_OR_RETURN_VALUE(params, float, gaussian_sigma, "gaussian_sigma", false);

  const size_t W0 = full_size_.width, H0 = full_size_.height;
  const int D0 = static_cast<int>(full_size_.depth);
  const int K = static_cast<int>(inner_iterations_count);
  const size_t crows = local_container_.height * local_container_.depth;
  const size_t cpitch = local_container_.pitch;

  f3d_size4 c = {local_container_.width, local_container_.height, local_container_.depth, local_container_.pitch};
  if (!Check(f3d_set_container(&c))) return false;

  // ---- pre-blur on the original planes: rows and columns on the slab widened by the tap radius, slices on the slab
  if (gaussian_sigma > 0.0) {
    taps_.ComputeGaussianKernel(gaussian_sigma, 3, 1.0);
    const size_t R = taps_.KernelRadius();
    if (static_cast<int>(R) > halo_) {
      std::printf("'%s': Gaussian radius %zu exceeds the halo capacity %d.\n", GetName(), R, halo_);
      return false;
    }
    if (!Check(f3d_set_conv_taps(taps_.Kernel(), 2 * R + 1))) return false;
    for (Local& l : locals_) {
      const f3d_slab wide = Window(D0, l.rank, static_cast<int>(R), static_cast<int>(R));
      const f3d_slab own = Window(D0, l.rank, 0, 0);
      const Role src[2] = {RAW0, RAW1}, dst[2] = {F0, F1};
      for (int i = 0; i < 2; ++i) {
        if (!Check(f3d_conv_rows_cols(l.buf[TMP], l.buf[src[i]], W0, H0, D0, R, &wide))) return false;  // rows + columns
        if (!Check(f3d_conv_slices(l.buf[dst[i]], l.buf[TMP], W0, H0, D0, R, &own))) return false;
      }
    }
  } else {
    const size_t bytes = cpitch * crows;
    for (Local& l : locals_) {
      if (!Check(f3d_copy_d2d(l.buf[F0], l.buf[RAW0], bytes))) return false;
      if (!Check(f3d_copy_d2d(l.buf[F1], l.buf[RAW1], bytes))) return false;
    }
  }

  DataSize4 original = {W0, H0, static_cast<size_t>(D0), 0};
  const size_t max_warp_level = GetMaxWarpLevel(W0, H0, D0, warp_scale_factor);
  int level = static_cast<int>(std::min(warp_levels_count, max_warp_level)) - 1;
  DataSize4 prev = {0, 0, 0, 0};

  if (level < 0)
    for (Local& l : locals_)
      for (Role r : {FU, FV, FW})
        if (!Check(f3d_memset2d(l.buf[r], cpitch, 0, cpitch, crows))) return false;

  // Two-pass (x, y) resample on the source slab, halo exchange of the intermediate, z pass onto the destination slab.
  // `items` pairs (source role, intermediate role, destination role); all share the source/destination geometry.
  struct Item { Role src, mid, dst; };
  auto resample = [&](const std::vector<Item>& items, const DataSize4& from, const DataSize4& to) -> bool {
    const int Din = static_cast<int>(from.depth), Dout = static_cast<int>(to.depth);
    int need_lo = 0, need_hi = 0;
    for (int r = 0; r < n_ranks_; ++r) {  // every rank must ask for the same depths: take the maximum
      const PlaneRange out = OwnedPlanes(Dout, r, n_ranks_), in = OwnedPlanes(Din, r, n_ranks_);
      if (out.empty()) continue;
      const PlaneRange src = ResampleSourcePlanes(Din, Dout, out);
      need_lo = std::max(need_lo, in.lo - src.lo);
      need_hi = std::max(need_hi, src.hi - in.hi);
    }
    // the items (two frames or three flow components) of a pass in one launch: f3d_resample_*_n
    const size_t n_items = items.size();
    DevicePtr srcs[3], mids_p[3], dsts[3];
    if (n_items == 0 || n_items > 3) return false;
    for (Local& l : locals_) {
      const f3d_slab in_own = Window(Din, l.rank, 0, 0);
      for (size_t k = 0; k < n_items; ++k) {
        srcs[k] = l.buf[items[k].src];
        mids_p[k] = l.buf[items[k].mid];
        dsts[k] = l.buf[items[k].dst];
      }
      if (!Check(f3d_resample_x_n(srcs, dsts, n_items, to.width, from.height, Din, from.width, &in_own))) return false;
      if (!Check(f3d_resample_y_n(dsts, mids_p, n_items, to.width, to.height, Din, from.height, &in_own))) return false;
    }
    std::vector<Role> mids;
    for (const Item& it : items) mids.push_back(it.mid);
    if (!Exchange(Din, to.width, to.height, mids, need_lo, need_hi)) return false;
    for (Local& l : locals_) {
      const f3d_slab in_win = Window(Din, l.rank, need_lo, need_hi);
      const f3d_slab out_own = Window(Dout, l.rank, 0, 0);
      for (size_t k = 0; k < n_items; ++k) {
        mids_p[k] = l.buf[items[k].mid];
        dsts[k] = l.buf[items[k].dst];
      }
      if (!Check(f3d_resample_z_n(mids_p, dsts, n_items, to.width, to.height, Dout, Din, &in_win, &out_own))) return false;
    }
    return true;
  };

  while (level >= 0) {
    const PyramidLevel lv = GetLevel(original, warp_scale_factor, level);
    const DataSize4 cur = lv.size;
    const size_t W = cur.width, H = cur.height;
    const int D = static_cast<int>(cur.depth);
    const float hx = lv.hx, hy = lv.hy, hz = lv.hz;
    if (!silent) std::printf("Solve level %2d (%4zu x%4zu x%4d) on %d slabs\n", level, W, H, D, n_ranks_);

    // frames of this level
    if (level == 0) {
      for (Local& l : locals_) {
        std::swap(l.buf[F0], l.buf[F0R]);
        std::swap(l.buf[F1], l.buf[F1R]);
      }
    } else {
      if (!resample({{F0, PHI, F0R}, {F1, KSI, F1R}}, original, cur)) return false;
    }
    // flow of the previous level (values stay in original-voxel units)
    if (prev.width == 0) {
      for (Local& l : locals_)
        for (Role r : {FU, FV, FW})
          if (!Check(f3d_memset2d(l.buf[r], cpitch, 0, cpitch, crows))) return false;
    } else {
      if (!resample({{FU, TDU, DU}, {FV, TDV, DV}, {FW, TDW, DW}}, prev, cur)) return false;
      for (Local& l : locals_) {
        std::swap(l.buf[FU], l.buf[DU]);
        std::swap(l.buf[FV], l.buf[DV]);
        std::swap(l.buf[FW], l.buf[DW]);
      }
    }

    // level-static halos: K+1 planes of u, v, w, f0 for the widened sweeps; f1 additionally by the warp's z reach
    float max_w = 0.f;
    for (Local& l : locals_) {
      const f3d_slab own = Window(D, l.rank, 0, 0);
      float m = 0.f;
      if (!Check(f3d_abs_max(l.buf[FW], W, H, D, &own, &m))) return false;
      max_w = std::max(max_w, m);  // f3d_abs_max returns the largest FINITE |w|: NaN / Inf flows are warped to frame_0
    }
    if (locals_.size() == 1 && n_ranks_ > 1 && !Check(f3d_comm_allreduce_max_f32(&max_w))) return false;
    // the float -> int conversion below is only defined for a finite, moderate quotient; every rank holds the same max_w here
    // (the all-reduce above), so they fail together
    if (!std::isfinite(max_w) || !(hz > 0.f) || max_w / hz > 1.0e6f) {
      std::printf("'%s': the flow is out of range at this level (max |w| = %g); cannot size the warp halo.\n", GetName(),
                  static_cast<double>(max_w));
      failed_ = true;
      return false;
    }
    const int reach = static_cast<int>(std::ceil(max_w / hz)) + 1;
    // Outer iterations per exchange of the increments.  Thick slabs exchange after every outer iteration (K + 1 planes, hidden
    // behind the interior where possible).  Thin slabs of a small level are latency-bound -- a launch costs the same with a
    // few planes more, an exchange costs a fixed ~50 us -- so they take n (K + 1) planes at once and run n outer iterations
    // on nested windows before the next exchange, like the out-of-core solver does per residency.  The rule uses only
    // quantities every rank agrees on.
    int n_ex = 1;
    {
      int min_slab = D, max_slab = 0;
      for (int r = 0; r < n_ranks_; ++r) {
        const int p = OwnedPlanes(D, r, n_ranks_).size();
        min_slab = std::min(min_slab, p);
        max_slab = std::max(max_slab, p);
      }
      const bool thick = overlap_min_planes_ > 0 && min_slab >= std::max(overlap_min_planes_, 4 * K + 4);
      auto affordable = [&](int n) {
        return n * (K + 1) + reach <= halo_ &&
               static_cast<double>(W) * static_cast<double>(H) * (max_slab + 2 * n * (K + 1)) <= small_level_voxels_;
      };
      if (n_ranks_ > 1 && !thick)
        for (int n = std::min<int>(max_outer_per_exchange_, static_cast<int>(outer_iterations_count)); n >= 2; --n)
          if (affordable(n)) {
            n_ex = n;
            break;
          }
      if (forced_outer_per_exchange_ > 0)
        n_ex = std::max(1, std::min(forced_outer_per_exchange_, std::min((halo_ - reach) / (K + 1), static_cast<int>(outer_iterations_count))));
    }
    // exchange after every solver stage: the deepest stage reads p_max planes of the increments and of the weights beyond the
    // slab, the weights p_max + 1 planes of everything else
    const bool per_stage = exchange_per_stage_ && n_ranks_ > 1;
    NoteSolveWeights(equation_alpha, hx, hy, hz);
    const int p_max = (FusedSweepsEnabled() && K >= 2) ? 2 : 1;
    if (per_stage) n_ex = 1;
    const int wide = per_stage ? p_max + 1 : n_ex * (K + 1);
    if (!Exchange(D, W, H, {FU, FV, FW, F0R}, wide, wide)) return false;
    if (n_ranks_ > 1 && wide + reach > halo_) {
      // the flow reaches further along z than the local containers have room for: frame 1 goes through a container of its own
      if (!WarpWithGatheredFrame(D, W, H, hx, hy, hz, wide, std::min(D, wide + reach))) return false;
    } else {
      if (!Exchange(D, W, H, {F1R}, wide + reach, wide + reach)) return false;
      // warp on the widened slab
      for (Local& l : locals_) {
        const f3d_slab win = Window(D, l.rank, wide, wide);
        if (!Check(f3d_warp(l.buf[F0R], l.buf[F1R], l.buf[FU], l.buf[FV], l.buf[FW], W, H, D, hx, hy, hz, l.buf[TMP], &win))) return false;
        std::swap(l.buf[F1R], l.buf[TMP]);
      }
    }

    // the frame derivatives of the level, where the rank holds containers for them: both frames are valid on the slab widened by
    // `wide` planes now (the exchange of frame 0, the window of the warp)
    for (Local& l : locals_)
      if (!FrameDerivatives(l, D, W, H, hx, hy, hz, wide)) return false;

    // solver: outer x (phi/ksi + K sweeps on shrinking windows), increments exchanged once per n_ex outer iterations
    for (Local& l : locals_) {
      l.weights_lo = l.weights_hi = 0;
      // the increments start from zero on every plane of the level the container can hold (own planes and halo room), one launch
      const f3d_slab room = Window(D, l.rank, halo_, halo_);
      const DevicePtr incr[3] = {l.buf[DU], l.buf[DV], l.buf[DW]};
      if (room.z_hi > room.z_lo && !Check(f3d_clear_box_n(incr, 3, W, H, D, &room))) return false;
    }
    // Overlapped order: one rank per process, a slab thick enough that zones and interior are distinct, and an exchange
    // to hide (not after the last outer iteration)
    const PlaneRange own_here = OwnedPlanes(D, locals_[0].rank, n_ranks_);
    const bool can_overlap = !per_stage && n_ex == 1 && locals_.size() == 1 && n_ranks_ > 1 && overlap_min_planes_ > 0 &&
                             own_here.size() >= std::max(overlap_min_planes_, 4 * K + 4) && (own_here.lo > 0 || own_here.hi < D);
    for (size_t i = 0; per_stage && i < outer_iterations_count; ++i) {
      // ---- one exchange per solver stage (F3D_SLAB_EXCHANGE=stage): every launch on the slab itself ----
      struct Stage { bool pair; };
      std::vector<Stage> stages;
      for (int s = 0; s < K;) {
        const bool pair = FusedSweepsEnabled() && s + 2 <= K;
        stages.push_back({pair});
        s += pair ? 2 : 1;
      }
      const bool more = i + 1 < outer_iterations_count;
      for (Local& l : locals_) {
        const PlaneRange own = OwnedPlanes(D, l.rank, n_ranks_);
        if (own.empty()) continue;
        if (!CompleteWeights(l, own.lo - p_max, own.hi + p_max, D, W, H, hx, hy, hz, equation_smoothness, equation_data)) return false;
      }
      // One rank per process and a slab thick enough: the exchange that follows a stage is hidden behind the stage's own interior --
      // the `next` planes at either end of the slab (what the neighbours are waiting for) are computed first, the transfer starts on
      // the side stream, the interior follows on the library stream, then the halos are unpacked.  Every voxel is computed once by
      // the same launches on the same inputs; only their order and their cut along z change (tests/test_gpu_slab_procs.py).
      const PlaneRange own0 = OwnedPlanes(D, locals_[0].rank, n_ranks_);
      const bool hide_stage_exchanges = locals_.size() == 1 && overlap_min_planes_ > 0 &&
                                        own0.size() >= std::max(overlap_min_planes_, 4 * K + 4) && (own0.lo > 0 || own0.hi < D);
      bool hidden_any = false;
      for (size_t st = 0; hide_stage_exchanges && st < stages.size(); ++st) {
        const bool last = st + 1 == stages.size();
        const int next = !last ? (stages[st + 1].pair ? 2 : 1) : (more ? p_max + 1 : 0);
        Local& l = locals_[0];
        const int a = own0.lo, b = own0.hi;
        const bool has_lo = a > 0, has_hi = b < D;
        const Role from[3] = {DU, DV, DW}, to[3] = {TDU, TDV, TDW};
        auto part = [&](int lo, int hi) {
          f3d_slab s;
          s.z_base = a - halo_;
          s.z_lo = lo;
          s.z_hi = hi;
          return s;
        };
        const int cut_lo = (has_lo && next > 0) ? a + next : a, cut_hi = (has_hi && next > 0) ? b - next : b;
        // the two zones the neighbours need (plain launches of this stage), then the transfer, then the interior
        if (cut_lo > a && !Sweeps(l, stages[st].pair, from, to, W, H, D, hx, hy, hz, equation_alpha, part(a, cut_lo))) return false;
        if (cut_hi < b && !Sweeps(l, stages[st].pair, from, to, W, H, D, hx, hy, hz, equation_alpha, part(cut_hi, b))) return false;
        if (next > 0) {
          f3d_comm_mark(0, 1);
          if (!ExchangeBegin(D, W, H, {to[0], to[1], to[2]}, {to[0], to[1], to[2]}, next, next)) return false;
          ++stage_exchanges_;
          hidden_any = true;
        }
        bool launched = false;
        if (last && !stages[st].pair && more && fused_weights_ && FusedSweepsEnabled() && FusedPhiKsiEnabled()) {
          if (!SweepAndNextWeights(l, {DU, DV, DW}, {TDU, TDV, TDW}, cut_lo, cut_hi, D, W, H, hx, hy, hz, equation_alpha, equation_smoothness,
                                   equation_data, launched))
            return false;
        }
        if (!launched && !Sweeps(l, stages[st].pair, from, to, W, H, D, hx, hy, hz, equation_alpha, part(cut_lo, cut_hi))) return false;
        if (next > 0) {
          if (!ExchangeEnd(W, H)) return false;
          f3d_comm_mark(1, 1);
        }
        std::swap(l.buf[DU], l.buf[TDU]);
        std::swap(l.buf[DV], l.buf[TDV]);
        std::swap(l.buf[DW], l.buf[TDW]);
      }
      if (hidden_any) ++overlapped_iterations_;
      for (size_t st = 0; !hide_stage_exchanges && st < stages.size(); ++st) {
        const bool last = st + 1 == stages.size();
        for (Local& l : locals_) {
          const PlaneRange own = OwnedPlanes(D, l.rank, n_ranks_);
          if (own.empty()) continue;
          const f3d_slab sw = Window(D, l.rank, 0, 0);
          bool launched = false;
          if (last && !stages[st].pair && more && fused_weights_ && FusedSweepsEnabled() && FusedPhiKsiEnabled()) {
            if (!SweepAndNextWeights(l, {DU, DV, DW}, {TDU, TDV, TDW}, sw.z_lo, sw.z_hi, D, W, H, hx, hy, hz, equation_alpha,
                                     equation_smoothness, equation_data, launched))
              return false;
          }
          if (!launched) {
            const Role from[3] = {DU, DV, DW}, to[3] = {TDU, TDV, TDW};
            if (!Sweeps(l, stages[st].pair, from, to, W, H, D, hx, hy, hz, equation_alpha, sw)) return false;
          }
          std::swap(l.buf[DU], l.buf[TDU]);
          std::swap(l.buf[DV], l.buf[TDV]);
          std::swap(l.buf[DW], l.buf[TDW]);
        }
        // as deep as what comes next reads: the sweeps of the next stage, or the weights of the next outer iteration
        const int next = !last ? (stages[st + 1].pair ? 2 : 1) : (more ? p_max + 1 : 0);
        if (next > 0) {
          if (!Exchange(D, W, H, {DU, DV, DW}, next, next)) return false;
          ++stage_exchanges_;
        }
      }
    }
    for (size_t i = per_stage ? outer_iterations_count : 0; i < outer_iterations_count;) {
      if (can_overlap && i + 1 < outer_iterations_count) {
        if (!SweepsOverlapped(locals_[0], D, W, H, K, hx, hy, hz, equation_alpha, equation_smoothness, equation_data)) return false;
        ++overlapped_iterations_;
        ++i;
        continue;
      }
      const int n = static_cast<int>(std::min<size_t>(n_ex, outer_iterations_count - i));
      for (Local& l : locals_) {
        const PlaneRange own = OwnedPlanes(D, l.rank, n_ranks_);
        if (own.empty()) continue;
        // iteration j of the n leaves the increments valid on the slab widened by g = (n-1-j)(K+1) planes
        for (int j = 0; j < n; ++j) {
          const int g = (n - 1 - j) * (K + 1);
          // the weights on the slab widened by g + K planes: all of them, or what the fused last sweep of the previous
          // iteration could not know yet (after an exchange: the planes next to the neighbours and the halo; else nothing)
          if (!CompleteWeights(l, own.lo - (g + K), own.hi + (g + K), D, W, H, hx, hy, hz, equation_smoothness, equation_data)) return false;
          // sweep s runs on the slab widened by g + K-1-s planes; a fused pair (s, s+1) is launched on the window of
          // sweep s+1 and computes sweep s on one plane more on either side by itself
          for (int s = 0; s < K;) {
            const bool pair = FusedSweepsEnabled() && s + 2 <= K;
            const int shrink = g + K - 1 - s - (pair ? 1 : 0);
            const f3d_slab sw = Window(D, l.rank, shrink, shrink);
            // the last sweep of an iteration that is not the level's last takes the next weights along (one launch)
            if (!pair && s == K - 1 && i + static_cast<size_t>(j) + 1 < outer_iterations_count && fused_weights_ &&
                FusedSweepsEnabled() && FusedPhiKsiEnabled()) {
              bool launched = false;
              if (!SweepAndNextWeights(l, {DU, DV, DW}, {TDU, TDV, TDW}, sw.z_lo, sw.z_hi, D, W, H, hx, hy, hz, equation_alpha,
                                       equation_smoothness, equation_data, launched))
                return false;
              if (launched) {
                std::swap(l.buf[DU], l.buf[TDU]);
                std::swap(l.buf[DV], l.buf[TDV]);
                std::swap(l.buf[DW], l.buf[TDW]);
                s += 1;
                continue;
              }
            }
            const Role from[3] = {DU, DV, DW}, to[3] = {TDU, TDV, TDW};
            if (!Sweeps(l, pair, from, to, W, H, D, hx, hy, hz, equation_alpha, sw)) return false;
            std::swap(l.buf[DU], l.buf[TDU]);
            std::swap(l.buf[DV], l.buf[TDV]);
            std::swap(l.buf[DW], l.buf[TDW]);
            s += pair ? 2 : 1;
          }
        }
      }
      i += n;
      if (n > 1) ++batched_exchanges_;
      if (i < outer_iterations_count) {
        const int next = static_cast<int>(std::min<size_t>(n_ex, outer_iterations_count - i)) * (K + 1);
        if (!Exchange(D, W, H, {DU, DV, DW}, next, next)) return false;
      }
    }

    // flow += increment on the slab, then the median with its own halo
    for (Local& l : locals_) {
      const f3d_slab own = Window(D, l.rank, 0, 0);
      const DevicePtr flow[3] = {l.buf[FU], l.buf[FV], l.buf[FW]}, incr[3] = {l.buf[DU], l.buf[DV], l.buf[DW]};
      if (!Check(f3d_add_n(flow, incr, 3, W, H, D, &own))) return false;
    }
    size_t radius = median_radius;
    if (radius != 1 && radius % 2 == 0) radius -= 1;
    if (radius != 1) {
      if (radius < 3 || radius > 7) {
        std::printf("Error. Wrong median raduis (%zu). Supported values: 3, 5, 7\n", radius);
        return false;
      }
      const int half = static_cast<int>(radius) / 2;
      if (!Exchange(D, W, H, {FU, FV, FW}, half, half)) return false;
      for (Local& l : locals_) {
        const f3d_slab own = Window(D, l.rank, 0, 0);
        // the three components in one launch; the increments are spent, their containers take the filtered flow
        const DevicePtr flow[3] = {l.buf[FU], l.buf[FV], l.buf[FW]}, filtered[3] = {l.buf[DU], l.buf[DV], l.buf[DW]};
        if (!Check(f3d_median_n(flow, 3, W, H, D, radius, filtered, &own))) return false;
        std::swap(l.buf[FU], l.buf[DU]);
        std::swap(l.buf[FV], l.buf[DV]);
        std::swap(l.buf[FW], l.buf[DW]);
      }
    }
    prev = cur;
    --level;
  }
  return !failed_;
}
