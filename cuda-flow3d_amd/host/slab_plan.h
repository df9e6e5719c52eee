// z-slab decomposition plan: which global planes a rank owns at a pyramid level and which planes it must
// exchange with which peer to make a halo of a given depth valid.  Pure host arithmetic (no device), shared by
// the multi-GPU driver and the CPU tests.  Design seed: the reference's out-of-core slabs
// (src/cuda_operations/partial_data/cuda_operation_solve_p.cpp:217-243, 721-745); everything else is new.
#ifndef F3D_HOST_SLAB_PLAN_H_
#define F3D_HOST_SLAB_PLAN_H_

#include <vector>

struct PlaneRange {
  int lo = 0, hi = 0;  // global planes [lo, hi)
  int size() const { return hi > lo ? hi - lo : 0; }
  bool empty() const { return hi <= lo; }
};

// Rank r of n owns planes [floor(r * depth / n), floor((r + 1) * depth / n)) of a level of that depth.
PlaneRange OwnedPlanes(int depth, int rank, int n_ranks);

struct HaloTransfer {
  int peer;
  PlaneRange send;  // planes of mine the peer needs
  PlaneRange recv;  // planes of the peer I need
};

// Transfers that make planes [own.lo - need_lo, own.lo) and [own.hi, own.hi + need_hi) (clipped to the volume) valid
// on every rank, assuming every rank asks for the same depths.  A halo deeper than a neighbour's slab reaches
// further ranks; a rank that owns nothing at this level sends nothing but still receives around its position.
std::vector<HaloTransfer> PlanHaloExchange(int depth, int rank, int n_ranks, int need_lo, int need_hi);

// Input planes the z pass of the area resample reads to produce output planes `out` (A.1: floor(z * delta) ..
// ceil((z + 1) * delta), delta = in_depth / out_depth in float like the kernel).
PlaneRange ResampleSourcePlanes(int in_depth, int out_depth, PlaneRange out);

#endif
