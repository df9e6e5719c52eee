// Probe: does an LDS-DMA (global_load_lds_dword, M0 = LDS address) reach LDS addresses above 64 KiB on gfx950?
// A 128 KiB LDS array is filled with -1, one wave DMAs a known row to byte address T for T = 1 KiB, 60 KiB, 66 KiB, 100 KiB,
// 127 KiB, and the whole array is scanned for where the 64 values landed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) float LdsFloat;
constexpr int N = 32 * 1024;  // floats = 128 KiB
__global__ __launch_bounds__(64) void k(const float* a, int* where, int target_float, int variant)
{
  __shared__ float lds[N];
  const unsigned lane = threadIdx.x;
  for (int i = lane; i < N; i += 64) lds[i] = -1.f;
  __syncthreads();
  const unsigned m0v = static_cast<unsigned>(reinterpret_cast<unsigned long>((LdsFloat*)&lds[target_float]));
  if (variant == 0) {
    asm volatile("s_nop 4\n\tglobal_load_lds_dword %0, %1" ::"v"(lane * 4u), "s"(a), "{m0}"(m0v) : "memory");
  } else {
    const float* p = a + lane;
    asm volatile("s_nop 4\n\tglobal_load_lds_dword %0, off" ::"v"(p), "{m0}"(m0v) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lane == 0) {
    int first = -1, count = 0;
    for (int i = 0; i < N; ++i)
      if (lds[i] != -1.f) { if (first < 0) first = i; ++count; }
    where[0] = first; where[1] = count; where[2] = (int)m0v;
    where[3] = first >= 0 ? (int)lds[first] : -1;
  }
}
int main()
{
  std::vector<float> h(64);
  for (int i = 0; i < 64; ++i) h[i] = 1000.f + i;
  float* a; int* w;
  hipMalloc(&a, 256); hipMalloc(&w, 16);
  hipMemcpy(a, h.data(), 256, hipMemcpyHostToDevice);
  for (int variant = 0; variant < 2; ++variant)
    for (int kb : {1, 60, 66, 100, 127}) {
      k<<<1, 64>>>(a, w, kb * 256, variant);
      int r[4];
      hipMemcpy(r, w, 16, hipMemcpyDeviceToHost);
      printf("variant %d target byte %6d (m0 %6d): %d values landed starting at byte %d (first value %d)\n", variant, kb * 1024, r[2],
             r[1], r[0] * 4, r[3]);
    }
  return 0;
}
