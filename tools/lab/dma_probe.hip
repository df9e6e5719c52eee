// Probe: semantics of buffer_load_dword ... lds (LDS-DMA) on gfx950 -- full-wave rows and an EXEC-masked pair of lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* a, float* out, int n)
{
  __shared__ float row[64];
  __shared__ float col[64];
  auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a), 0, n * 4, 0x00020000);
  const unsigned lane = threadIdx.x;
  row[lane] = -1.f;
  col[lane] = -1.f;
  __syncthreads();
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, &row[0], 4, lane * 4u, 1000u * 4u, 0, 0);   // a[1000 + lane]
  const unsigned voff = (lane < 32 ? 7u : 9u) * 4u;
  if (lane == 0 || lane == 32)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, &col[5], 4, voff, 2000u * 4u, 0, 0);        // a[2007], a[2009]
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  out[lane] = row[lane];
  out[64 + lane] = col[lane];
}
int main()
{
  const int n = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = float(i);
  float *a, *o;
  hipMalloc(&a, n * 4); hipMalloc(&o, 128 * 4);
  hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(a, o, n);
  std::vector<float> r(128);
  hipMemcpy(r.data(), o, 128 * 4, hipMemcpyDeviceToHost);
  bool ok = true;
  for (int i = 0; i < 64; ++i) ok &= r[i] == 1000.f + i;
  printf("row DMA %s\n", ok ? "ok" : "BAD");
  printf("masked DMA landed at:");
  for (int i = 0; i < 64; ++i) if (r[64 + i] != -1.f) printf(" col[%d]=%g", i, r[64 + i]);
  printf("\n");
  return 0;
}
