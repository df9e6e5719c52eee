#include "slab_plan.h"

#include <algorithm>
#include <cmath>

PlaneRange OwnedPlanes(int depth, int rank, int n_ranks)
{
  PlaneRange r;
  r.lo = static_cast<int>(static_cast<long long>(rank) * depth / n_ranks);
  r.hi = static_cast<int>(static_cast<long long>(rank + 1) * depth / n_ranks);
  return r;
}

namespace {
PlaneRange Intersect(PlaneRange a, PlaneRange b)
{
  PlaneRange r;
  r.lo = std::max(a.lo, b.lo);
  r.hi = std::min(a.hi, b.hi);
  if (r.hi < r.lo) r.hi = r.lo;
  return r;
}

// the planes below / above its slab a rank asks for
void Wanted(int depth, PlaneRange own, int need_lo, int need_hi, PlaneRange* below, PlaneRange* above)
{
  // a rank that owns nothing here still asks around its (empty) position: the z pass of a resample can need
  // source planes for a destination slab that is not empty
  below->lo = std::max(0, own.lo - need_lo);
  below->hi = own.lo;
  above->lo = own.hi;
  above->hi = std::min(depth, own.hi + need_hi);
}
}  // namespace

std::vector<HaloTransfer> PlanHaloExchange(int depth, int rank, int n_ranks, int need_lo, int need_hi)
{
  std::vector<HaloTransfer> plan;
  const PlaneRange mine = OwnedPlanes(depth, rank, n_ranks);
  PlaneRange my_below, my_above;
  Wanted(depth, mine, need_lo, need_hi, &my_below, &my_above);
  for (int q = 0; q < n_ranks; ++q) {
    if (q == rank) continue;
    const PlaneRange theirs = OwnedPlanes(depth, q, n_ranks);
    PlaneRange their_below, their_above;
    Wanted(depth, theirs, need_lo, need_hi, &their_below, &their_above);
    HaloTransfer t;
    t.peer = q;
    // a lower rank owns planes below mine; a higher rank planes above
    t.recv = Intersect(q < rank ? my_below : my_above, theirs);
    t.send = Intersect(q > rank ? their_below : their_above, mine);
    if (!t.recv.empty() || !t.send.empty()) plan.push_back(t);
  }
  return plan;
}

PlaneRange ResampleSourcePlanes(int in_depth, int out_depth, PlaneRange out)
{
  PlaneRange r;
  if (out.empty()) return r;
  const float delta = static_cast<float>(in_depth) / static_cast<float>(out_depth);
  r.lo = static_cast<int>(std::floor(static_cast<float>(out.lo) * delta));
  r.hi = static_cast<int>(std::fmin(static_cast<float>(in_depth), std::ceil(static_cast<float>(out.hi) * delta)));
  // the per-plane windows are monotone in z, so the union over `out` is [first.lo, last.hi)
  return r;
}
