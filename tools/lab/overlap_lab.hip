// Lab: does arithmetic overlap with the streaming loads in a z-march kernel?  10 loads + 3 stores per voxel-step,
// synthetic VALU work per step (NF mul+add pairs on 8 independent chains + ND IEEE divisions), prefetch depth P.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Args { const float* in[10]; float* out[3]; };

template <int TY, int P, int NF, int ND, int WPS, int LX = 0>
__global__ __launch_bounds__(64 * TY, WPS) void k_march(Args a, int W, int H, int D, int pitch, int zchunk)
{
  const int lane = threadIdx.x;
  const int y = __builtin_amdgcn_readfirstlane(int(blockIdx.y) * TY + int(threadIdx.y));
  const int x = blockIdx.x * 64 + lane;
  const int z0 = blockIdx.z * zchunk, z1 = min(z0 + zchunk, D);
  if (y >= H) return;
  float c[10], q[P][10];
  __shared__ float sh[LX == 1 ? 2 : 1][9][TY][64];
  const int r = threadIdx.y;
  auto row = [&](int zz) { return (size_t(min(zz, D - 1)) * H + y) * pitch + x; };
#pragma unroll
  for (int i = 0; i < 10; ++i) c[i] = a.in[i][row(z0)];
#pragma unroll
  for (int d = 0; d < P - 1; ++d)
#pragma unroll
    for (int i = 0; i < 10; ++i) q[d][i] = a.in[i][row(z0 + 1 + d)];
  for (int z = z0; z < z1; ++z) {
#pragma unroll
    for (int i = 0; i < 10; ++i) q[P - 1][i] = a.in[i][row(z + P)];
    float yn = 0.f;
    if (LX == 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sh[z & 1][i][r][lane] = c[i];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 9; ++i) yn += sh[z & 1][i][(r + 1) % TY][lane] - sh[z & 1][i][(r + TY - 1) % TY][lane];
    }
    if (LX == 2) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 9; ++i) sh[0][i][r][lane] = c[i];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 9; ++i) yn += sh[0][i][(r + 1) % TY][lane] - sh[0][i][(r + TY - 1) % TY][lane];
    }
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = c[i];
#pragma unroll
    for (int it = 0; it < NF; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = acc[i] * c[8] + c[9];
    float s = yn;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
#pragma unroll
    for (int d = 0; d < ND; ++d) s = (s + c[d]) / (acc[d] + 3.f);
    const size_t o = row(z);
    a.out[0][o] = s; a.out[1][o] = s * 2.f; a.out[2][o] = s * 3.f;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      c[i] = q[0][i];
#pragma unroll
      for (int d = 0; d + 1 < P; ++d) q[d][i] = q[d + 1][i];
    }
  }
}

int main()
{
  const int W = 512, H = 512, D = 512, pitch = 512;
  const size_t n = size_t(W) * H * D;
  Args a;
  std::vector<float> host(n);
  for (size_t i = 0; i < n; ++i) host[i] = 0.5f + float(i % 977) * 1e-3f;
  for (int i = 0; i < 10; ++i) { float* p; CK(hipMalloc(&p, n * 4 + 256)); CK(hipMemcpy(p, host.data(), n * 4, hipMemcpyHostToDevice)); a.in[i] = p; }
  for (int i = 0; i < 3; ++i) { float* p; CK(hipMalloc(&p, n * 4 + 256)); a.out[i] = p; }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-52s %8.3f ms\n", name, ms);
    return 0;
  };
  const int zc = 64;
  dim3 g4(W / 64, H / 4, D / zc), b4(64, 4);
  dim3 g8(W / 64, H / 8, D / zc), b8(64, 8);
#define RUN(TY, P, NF, ND, WPS, LX) time("TY=" #TY " P=" #P " NF=" #NF " ND=" #ND " waves/SIMD>=" #WPS " LX=" #LX, [&] { k_march<TY, P, NF, ND, WPS, LX><<<(TY == 4 ? g4 : g8), (TY == 4 ? b4 : b8)>>>(a, W, H, D, pitch, zc); })
  RUN(4, 1, 0, 0, 4, 0);
  RUN(4, 1, 16, 6, 4, 0);
  RUN(4, 1, 16, 6, 4, 1);
  RUN(4, 1, 16, 6, 4, 2);
  RUN(8, 1, 16, 6, 4, 0);
  RUN(8, 1, 16, 6, 4, 1);
  RUN(8, 1, 16, 6, 4, 2);
  RUN(8, 2, 16, 6, 4, 1);
  RUN(8, 1, 24, 6, 4, 1);
  RUN(8, 1, 0, 0, 4, 1);
  return 0;
}
