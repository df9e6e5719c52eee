// Lab: what does ISSUING a global_load_dword cost the issuing wave on gfx950?  One workgroup per CU, W waves, each wave
// issues 10 loads from 10 arrays back to back (row-per-wave, 256 B per load), then waits.  s_memtime around the issue
// burst and around the wait.  Variants: idle machine (1 workgroup) vs every CU busy; misses (fresh lines every step).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Args { const float* in[10]; };

__device__ __forceinline__ float gld(const float* base, unsigned byte_off)
{
  float v;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(base) : "memory");
  return v;
}

template <bool BUILTIN, int NB>
__global__ __launch_bounds__(1024) void k(Args a, float* out, unsigned long long* stamps, int steps, size_t plane_floats)
{
  const int lane = threadIdx.x, w = threadIdx.y, nw = blockDim.y;
  unsigned long long t_issue = 0, t_wait = 0;
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    const size_t row = (size_t(blockIdx.x) * steps + s) * nw + w;      // a fresh 256-B row per wave and step
    const unsigned off = unsigned(lane) * 4u + unsigned((row * 64) % plane_floats) * 4u;
    const float* b[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) b[i] = a.in[i];
    float v[10 * NB];
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (BUILTIN) {
#pragma unroll
      for (int i = 0; i < 10; ++i) v[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(b[i]) + off);
    } else {
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < 10; ++i) v[j * 10 + i] = gld(b[i], off + unsigned(j) * (1u << 22));
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!BUILTIN) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 10 * NB; ++i) asm volatile("" : "+v"(v[i]));
    }
#pragma unroll
    for (int i = 0; i < 10 * NB; ++i) acc += v[i];
    asm volatile("" ::"v"(acc));
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    t_issue += t1 - t0;
    t_wait += t2 - t1;
  }
  out[(blockIdx.x * nw + w) * 64 + lane] = acc;
  if (lane == 0 && blockIdx.x == 0) {
    stamps[w * 2] = t_issue;
    stamps[w * 2 + 1] = t_wait;
  }
}

// One step of the fused kernel's memory shape per wave: 3 dword stores then 10 dword loads (SEP), or the same bytes as
// one 12-byte store and 16 + 4 + 8 + 12 byte loads (PACKED).  Issue time only.
struct f3 { float x, y, z; };
template <bool PACKED>
__global__ __launch_bounds__(1024) void k2(Args a, float* o0, float* o1, float* o2, unsigned long long* stamps, int steps,
                                           size_t plane_floats)
{
  const int lane = threadIdx.x, w = threadIdx.y, nw = blockDim.y;
  unsigned long long t_st = 0, t_ld = 0, t_wait = 0;
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    const size_t row = (size_t(blockIdx.x) * steps + s) * nw + w;
    const size_t e = (row * 64) % (plane_floats / 4) + lane;   // element index (voxel)
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (PACKED) {
      reinterpret_cast<f3*>(o0)[e] = f3{acc, acc + 1.f, acc + 2.f};
    } else {
      o0[e] = acc; o1[e] = acc + 1.f; o2[e] = acc + 2.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float v[10];
    if (PACKED) {
      const float4 c = reinterpret_cast<const float4*>(a.in[0])[e];
      const float ww = a.in[1][e];
      const float2 p = reinterpret_cast<const float2*>(a.in[2])[e];
      const f3 d = reinterpret_cast<const f3*>(a.in[3])[e];
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      t_ld += t2 - t1;
      acc += c.x + c.y + c.z + c.w + ww + p.x + p.y + d.x + d.y + d.z;
      asm volatile("" ::"v"(acc));
      __builtin_amdgcn_sched_barrier(0);
      t_wait += __builtin_amdgcn_s_memtime() - t2;
    } else {
      asm volatile("global_load_dword %0, %10, %11\n\tglobal_load_dword %1, %10, %12\n\tglobal_load_dword %2, %10, %13\n\t"
                   "global_load_dword %3, %10, %14\n\tglobal_load_dword %4, %10, %15\n\tglobal_load_dword %5, %10, %16\n\t"
                   "global_load_dword %6, %10, %17\n\tglobal_load_dword %7, %10, %18\n\tglobal_load_dword %8, %10, %19\n\t"
                   "global_load_dword %9, %10, %20"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                     "=&v"(v[8]), "=&v"(v[9])
                   : "v"(unsigned(e) * 4u), "s"(a.in[0]), "s"(a.in[1]), "s"(a.in[2]), "s"(a.in[3]), "s"(a.in[4]), "s"(a.in[5]),
                     "s"(a.in[6]), "s"(a.in[7]), "s"(a.in[8]), "s"(a.in[9])
                   : "memory");
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      t_ld += t2 - t1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 10; ++i) { asm volatile("" : "+v"(v[i])); acc += v[i]; }
      asm volatile("" ::"v"(acc));
      __builtin_amdgcn_sched_barrier(0);
      t_wait += __builtin_amdgcn_s_memtime() - t2;
    }
    t_st += t1 - t0;
  }
  o1[(blockIdx.x * nw + w) * 64 + lane + plane_floats / 2] = acc;
  if (lane == 0 && blockIdx.x == 0) {
    stamps[w * 4] = t_st; stamps[w * 4 + 1] = t_ld; stamps[w * 4 + 2] = t_wait;
  }
}

int main()
{
  const size_t plane = size_t(64) << 20;  // 64 M floats = 256 MB per array
  Args a;
  for (int i = 0; i < 10; ++i) { float* p; CK(hipMalloc(&p, plane * 4)); CK(hipMemset(p, 0, plane * 4)); a.in[i] = p; }
  float* out; CK(hipMalloc(&out, 1024 * 1024 * 4));
  unsigned long long* st; CK(hipMalloc(&st, 64 * 8));
  for (int nb = 1; nb <= 3; ++nb)
  for (int builtin = 0; builtin < 1; ++builtin)
    for (int blocks : {1, 256})
      for (int nw : {1, 4, 12}) {
        const int steps = 200;
        if (builtin) k<true, 1><<<blocks, dim3(64, nw)>>>(a, out, st, steps, plane);
        else if (nb == 1) k<false, 1><<<blocks, dim3(64, nw)>>>(a, out, st, steps, plane);
        else if (nb == 2) k<false, 2><<<blocks, dim3(64, nw)>>>(a, out, st, steps, plane);
        else k<false, 3><<<blocks, dim3(64, nw)>>>(a, out, st, steps, plane);
        CK(hipDeviceSynchronize());
        unsigned long long h[64];
        CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
        double is = 0, wt = 0;
        for (int w = 0; w < nw; ++w) { is += h[w * 2]; wt += h[w * 2 + 1]; }
        printf("%d loads per burst, %3d workgroups x %2d waves: issue %7.1f cycles/burst/wave (%5.1f per load), wait %8.1f\n",
               10 * nb, blocks, nw, is / nw / steps, is / nw / steps / (10 * nb), wt / nw / steps);
      }
  {
    float *o0, *o1, *o2;
    CK(hipMalloc(&o0, plane * 4)); CK(hipMalloc(&o1, plane * 4)); CK(hipMalloc(&o2, plane * 4));
    for (int packed = 0; packed < 2; ++packed)
      for (int blocks : {1, 256})
        for (int nw : {4, 12}) {
          const int steps = 200;
          if (packed) k2<true><<<blocks, dim3(64, nw)>>>(a, o0, o1, o2, st, steps, plane);
          else k2<false><<<blocks, dim3(64, nw)>>>(a, o0, o1, o2, st, steps, plane);
          CK(hipDeviceSynchronize());
          unsigned long long h[64];
          CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
          double ts = 0, tl = 0, tw = 0;
          for (int w = 0; w < nw; ++w) { ts += h[w * 4]; tl += h[w * 4 + 1]; tw += h[w * 4 + 2]; }
          printf("%s, %3d workgroups x %2d waves: stores %7.1f, loads %7.1f, wait %8.1f cycles/step/wave\n",
                 packed ? "packed (1 store, 4 loads) " : "separate (3 stores, 10 loads)", blocks, nw, ts / nw / steps, tl / nw / steps,
                 tw / nw / steps);
        }
  }
  return 0;
}
