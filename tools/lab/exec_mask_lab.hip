// Micro-benchmark: does a vector instruction cost less when part of the wave is masked off?
// A wave64 instruction runs over four passes of 16 lanes on gfx9; the question behind the remainder tile column of k_pair8 (the last
// tile column of a level whose width is not a multiple of 64 computes lanes beyond the volume) is whether passes whose 16 lanes are all
// inactive are skipped.  Chains of independent mul + add (contraction off) under `if (lane < active)`, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, int active)
{
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x * 1e-3f;
  const float b = seed * 0.999f, c = seed * 1e-3f;
  if (static_cast<int>(threadIdx.x & 63) < active) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = a[i] * b + c;
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
  const int iters = 20000, waves_per_simd = 4;
  const int blocks = 256 * waves_per_simd;
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int active : {64, 49, 48, 33, 32, 17, 16, 8, 1}) {
    k<<<blocks, 256>>>(out, 100, 1.0f, active);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<blocks, 256>>>(out, iters, 1.0f, active);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = double(blocks) * 4 * iters * 16.0 / 1024.0;
    printf("active lanes %2d: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", active, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  }
  hipFree(out);
  return 0;
}
