// Is there a cheaper way to the SAME float as the reference's   phi = 1.f / (2.f * sqrtf(acc))   (solve_3d.cu:203-204, 259-260)?
// The expression is a function of ONE binary32 argument, so a candidate can be checked against the IEEE chain for EVERY input:
// 2^32 evaluations, a few milliseconds on the MI355X.  Candidates start from v_rsq_f32 (y ~ 1/sqrt(a), about 1 ulp) and refine
// with fused multiply-adds (explicit v_fma_f32: each a single correctly rounded operation, so the result is a deterministic
// function of the input -- the point of the exhaustive check):
//   s  = RN(sqrt(a))        s0 = a*y; h = y/2; e = fma(-s0, s0, a); s1 = fma(e, h, s0)    [; e = fma(-s1, s1, a); s2 = fma(e, h, s1)]
//   t  = 2 s                exact
//   p  = RN(1 / t)          r0 = h;   e = fma(-t, r0, 1); r1 = fma(e, r0, r0)              [; e = fma(-t, r1, 1); r2 = fma(e, r1, r1)]
// Prints, per candidate, how many inputs in [lo, hi] give other bits than the IEEE chain, and the first few of them.
// Build: make -C tools/lab bin/phi_exact_lab   (flags as the product: -ffp-contract=off, correctly rounded / and sqrt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ float ieee_chain(float a) { return 1.f / (2.f * sqrtf(a)); }

template <int NS, int NR>
__device__ __forceinline__ float candidate(float a)
{
  const float y = __builtin_amdgcn_rsqf(a);
  const float h = 0.5f * y;
  float s = a * y;
#pragma unroll
  for (int i = 0; i < NS; ++i) s = __builtin_fmaf(__builtin_fmaf(-s, s, a), h, s);
  const float t = s + s;
  float r = h;
#pragma unroll
  for (int i = 0; i < NR; ++i) r = __builtin_fmaf(__builtin_fmaf(-t, r, 1.f), r, r);
  return r;
}
// the square root alone (is the refined s the correctly rounded root?)
template <int NS>
__device__ __forceinline__ float candidate_sqrt(float a)
{
  const float y = __builtin_amdgcn_rsqf(a);
  const float h = 0.5f * y;
  float s = a * y;
#pragma unroll
  for (int i = 0; i < NS; ++i) s = __builtin_fmaf(__builtin_fmaf(-s, s, a), h, s);
  return s;
}

struct Tally { unsigned long long bad[8]; unsigned first[8][4]; };

__global__ void k_sweep_all(unsigned lo_bits, unsigned hi_bits, Tally* t)
{
  const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
  for (unsigned long long i = lo_bits + static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; i <= hi_bits; i += stride) {
    const float a = __uint_as_float(static_cast<unsigned>(i));
    const unsigned want = __float_as_uint(ieee_chain(a));
    const unsigned got[6] = {__float_as_uint(candidate<1, 1>(a)), __float_as_uint(candidate<1, 2>(a)), __float_as_uint(candidate<2, 1>(a)),
                             __float_as_uint(candidate<2, 2>(a)), __float_as_uint(candidate<1, 3>(a)), __float_as_uint(candidate<2, 3>(a))};
#pragma unroll
    for (int c = 0; c < 6; ++c)
      if (got[c] != want) {
        const unsigned long long n = atomicAdd(&t->bad[c], 1ull);
        if (n < 4) t->first[c][n] = static_cast<unsigned>(i);
      }
    const unsigned sw = __float_as_uint(sqrtf(a));
    if (__float_as_uint(candidate_sqrt<1>(a)) != sw) { const unsigned long long n = atomicAdd(&t->bad[6], 1ull); if (n < 4) t->first[6][n] = static_cast<unsigned>(i); }
    if (__float_as_uint(candidate_sqrt<2>(a)) != sw) { const unsigned long long n = atomicAdd(&t->bad[7], 1ull); if (n < 4) t->first[7][n] = static_cast<unsigned>(i); }
  }
}

int main(int argc, char** argv)
{
  // positive normal range by default: [2^-100, 2^100]; arguments: lo and hi as hex bit patterns
  unsigned lo = 0x0d800000u, hi = 0x71800000u;
  if (argc > 2) { lo = std::strtoul(argv[1], nullptr, 16); hi = std::strtoul(argv[2], nullptr, 16); }
  Tally* d; CK(hipMalloc(&d, sizeof(Tally))); CK(hipMemset(d, 0, sizeof(Tally)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_sweep_all, dim3(256 * 16), dim3(256), 0, 0, lo, hi, d);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  Tally h; CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
  std::printf("inputs 0x%08x .. 0x%08x (%llu values), %.1f ms\n", lo, hi, static_cast<unsigned long long>(hi) - lo + 1ull, ms);
  const char* names[8] = {"phi: sqrt x1, rcp x1", "phi: sqrt x1, rcp x2", "phi: sqrt x2, rcp x1", "phi: sqrt x2, rcp x2",
                          "phi: sqrt x1, rcp x3", "phi: sqrt x2, rcp x3", "sqrt alone x1", "sqrt alone x2"};
  for (int c = 0; c < 8; ++c) {
    std::printf("%-22s: %llu mismatches", names[c], h.bad[c]);
    for (unsigned k = 0; k < 4 && k < h.bad[c]; ++k) { float f; std::memcpy(&f, &h.first[c][k], 4); std::printf("  0x%08x (%g)", h.first[c][k], f); }
    std::printf("\n");
  }
  return 0;
}
