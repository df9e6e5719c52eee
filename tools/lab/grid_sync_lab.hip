// What does a grid-wide barrier cost on the MI355X?  (small levels: would ONE cooperative launch per
// outer iteration with barriers between its six stages beat six/three dependent launches?)
//   cg:     cooperative_groups::grid_group::sync()
//   own:    one device-scope atomic counter + generation word, release/acquire, one lane per workgroup spins
// Every round each workgroup publishes a word, crosses the barrier and reads the word of another workgroup (on another XCD),
// so the figure includes making the data visible across the XCDs' L2s.  Launched with hipLaunchCooperativeKernel, which
// refuses a grid that is not co-resident, so the spin cannot deadlock.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void k_cg(unsigned* buf, unsigned* out, int rounds)
{
  cg::grid_group grid = cg::this_grid();
  const unsigned n = gridDim.x, me = blockIdx.x;
  unsigned acc = 0;
  for (int r = 0; r < rounds; ++r) {
    if (threadIdx.x == 0) buf[(r & 1) * n + me] = r * 977u + me;
    grid.sync();
    acc += buf[(r & 1) * n + (me + 9) % n];
  }
  if (threadIdx.x == 0) out[me] = acc;
}

__global__ void k_own(unsigned* buf, unsigned* out, unsigned* counter, unsigned* generation, int rounds)
{
  const unsigned n = gridDim.x, me = blockIdx.x;
  unsigned acc = 0;
  for (int r = 0; r < rounds; ++r) {
    if (threadIdx.x == 0) __hip_atomic_store(&buf[(r & 1) * n + me], r * 977u + me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned gen = __hip_atomic_load(generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == n - 1) {
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(generation, gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(generation, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    acc += __hip_atomic_load(&buf[(r & 1) * n + (me + 9) % n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) out[me] = acc;
}

// flags: every workgroup publishes its arrival in a slot of its own (a plain release store, no read-modify-write), the first wave
// of workgroup 0 watches all slots with one load per poll (one lane per slot, n <= 64) and then raises the go word the others
// poll: two trips through the fabric whatever n is, no serialised atomics
__global__ void k_flags(unsigned* buf, unsigned* out, unsigned* slots, unsigned* go, int rounds)
{
  const unsigned n = gridDim.x, me = blockIdx.x;
  unsigned acc = 0;
  for (int r = 0; r < rounds; ++r) {
    const unsigned gen = r + 1;
    if (threadIdx.x == 0) __hip_atomic_store(&buf[(r & 1) * n + me], r * 977u + me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&slots[me * 16], gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (me == 0) {
      if (threadIdx.x < 64) {
        const unsigned lane = threadIdx.x;
        for (;;) {
          const unsigned v = lane < n ? __hip_atomic_load(&slots[lane * 16], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : gen;
          if (__builtin_amdgcn_ballot_w64(v != gen) == 0) break;
        }
        if (lane == 0) __hip_atomic_store(go, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (threadIdx.x == 0) {
      while (__hip_atomic_load(go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != gen) {}
    }
    __syncthreads();
    acc += __hip_atomic_load(&buf[(r & 1) * n + (me + 9) % n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) out[me] = acc;
}

__global__ void k_trivial(unsigned* buf) { if (threadIdx.x == 0) buf[blockIdx.x] += 1; }

int main()
{
  unsigned *buf, *out, *ctr;
  CK(hipMalloc(&buf, 4096 * 4)); CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&ctr, 256));
  CK(hipMemset(buf, 0, 4096 * 4)); CK(hipMemset(ctr, 0, 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 2000;
  unsigned *slots, *go;
  CK(hipMalloc(&slots, 64 * 64)); CK(hipMalloc(&go, 256));
  for (int threads : {256, 768}) for (int wgs : {8, 16, 32, 64, 128, 256}) {
    for (int which = 0; which < 3; ++which) {
      if (which == 2 && wgs > 64) continue;
      CK(hipMemset(slots, 0, 64 * 64)); CK(hipMemset(go, 0, 256));
      int r = rounds; unsigned* gen = ctr + 16;
      void* a_cg[] = {&buf, &out, &r};
      void* a_own[] = {&buf, &out, &ctr, &gen, &r};
      void* a_flags[] = {&buf, &out, &slots, &go, &r};
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (which == 0) CK(hipLaunchCooperativeKernel((void*)k_cg, dim3(wgs), dim3(threads), a_cg, 0, 0));
        else if (which == 1) CK(hipLaunchCooperativeKernel((void*)k_own, dim3(wgs), dim3(threads), a_own, 0, 0));
        else { CK(hipMemset(slots, 0, 64 * 64)); CK(hipMemset(go, 0, 256)); CK(hipEventRecord(e0));
               CK(hipLaunchCooperativeKernel((void*)k_flags, dim3(wgs), dim3(threads), a_flags, 0, 0)); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      unsigned h[256]; CK(hipMemcpy(h, out, wgs * 4, hipMemcpyDeviceToHost));
      unsigned exp = 0; for (int q = 0; q < rounds; ++q) exp += q * 977u + (0 + 9) % wgs;
      std::printf("%s  %3d workgroups x %3d lanes: %.2f us per barrier round%s\n", which == 0 ? "cg   " : which == 1 ? "own  " : "flags", wgs, threads,
                  best * 1e3f / rounds, h[0] == exp ? "" : "  (WRONG DATA)");
    }
  }
  CK(hipEventRecord(e0));
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_trivial, dim3(256), dim3(768), 0, 0, buf);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("dependent trivial launches: %.2f us each\n", ms * 1e3f / 2000);
  return 0;
}
