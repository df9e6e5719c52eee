// Micro-benchmark: VALU issue rate on gfx950 for plain f32, packed f32 and IEEE division, at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
  float a[8];
  float2v pa[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; pa[i] = float2v{a[i], a[i] + 1.f}; }
  const float b = seed * 0.999f, c = seed * 1e-3f;
  const double rd = 1.0 / static_cast<double>(seed * 3.3f);
  bool tiny = false;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = a[i] * b + c;           // mul + add (contraction off): 2 instr
      if (MODE == 1) pa[i] = pa[i] * b + c;         // pk_mul + pk_add: 2 instr, 4 flops
      if (MODE == 2) a[i] = c / (a[i] + b);         // IEEE divide
      if (MODE == 3) a[i] = sqrtf(a[i] + b);        // IEEE sqrt
      if (MODE == 4) {                              // exact division by a uniform divisor through double: cvt, mul_f64, cvt
        const float x = a[i] + b;
        a[i] = static_cast<float>(static_cast<double>(x) * rd);
      }
      if (MODE == 5) {                              // ... plus the guard for tiny non-zero numerators (3 integer ops)
        const float x = a[i] + b;
        const unsigned u = __float_as_uint(x) & 0x7fffffffu;
        tiny |= (u - 1u) < 0x0d800000u - 1u;
        a[i] = static_cast<float>(static_cast<double>(x) * rd);
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + pa[i].x + pa[i].y;
  if (tiny) s += 1.f;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int instr_per_it, int waves_per_simd)
{
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;  // 256-thread blocks = 1 wave per SIMD each
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double wave_instr = double(blocks) * 4 * iters * 8.0 * instr_per_it;   // per-wave instruction count
  double per_simd = wave_instr / 1024.0;                                  // instr per SIMD
  printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, waves_per_simd,
         ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  hipFree(out);
}

int main()
{
  for (int w : {1, 2, 4}) {
    run<0>("mul+add", 2, w);
    run<1>("pk mul+add", 2, w);
    run<2>("div(+add)", 1, w);
    run<3>("sqrt(+add)", 1, w);
    run<4>("udiv f64(+add)", 1, w);
    run<5>("udiv f64+guard", 1, w);
  }
  return 0;
}
