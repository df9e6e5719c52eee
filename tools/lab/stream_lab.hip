// Lab: what does the MI355X memory system deliver for the sweep's access shape (10 input + 3 output streams, 512^3)?
//  A: flat grid-stride float4 copy-sum (ceiling)            B: flat dword
//  C: z-march, wave = 64-float row, WG = TY rows, 10 loads + 3 stores / voxel / step (no neighbours), prefetch 1 plane
//  D: C + the y-neighbour row loads of 9 arrays (28 loads), like k_solver v1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Args { const float* in[10]; float* out[3]; };

template <typename T>
__global__ __launch_bounds__(256) void k_flat(Args a, size_t n)
{
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    T s = reinterpret_cast<const T*>(a.in[0])[i];
#pragma unroll
    for (int k = 1; k < 10; ++k) s += reinterpret_cast<const T*>(a.in[k])[i];
    reinterpret_cast<T*>(a.out[0])[i] = s;
    reinterpret_cast<T*>(a.out[1])[i] = s * 2.f;
    reinterpret_cast<T*>(a.out[2])[i] = s * 3.f;
  }
}

template <int TY, bool YN>
__global__ __launch_bounds__(64 * TY) void k_march(Args a, int W, int H, int D, int pitch, int zchunk)
{
  const int lane = threadIdx.x;
  const int y = __builtin_amdgcn_readfirstlane(int(blockIdx.y) * TY + int(threadIdx.y));
  const int x = blockIdx.x * 64 + lane;
  const int z0 = blockIdx.z * zchunk, z1 = min(z0 + zchunk, D);
  if (y >= H || x >= W) return;
  const int yl = y == 0 ? 1 : y - 1, yh = y == H - 1 ? H - 2 : y + 1;
  float c[10], q[10], l[9], h[9], nl[9], nh[9];
  auto row = [&](int yy, int zz) { return (size_t(zz) * H + yy) * pitch + x; };
#pragma unroll
  for (int i = 0; i < 10; ++i) c[i] = a.in[i][row(y, z0)];
  if (YN) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { l[i] = a.in[i][row(yl, z0)]; h[i] = a.in[i][row(yh, z0)]; }
  }
  for (int z = z0; z < z1; ++z) {
    const bool more = z + 1 < z1;
    if (more) {
#pragma unroll
      for (int i = 0; i < 10; ++i) q[i] = a.in[i][row(y, z + 1)];
      if (YN) {
#pragma unroll
        for (int i = 0; i < 9; ++i) { nl[i] = a.in[i][row(yl, z + 1)]; nh[i] = a.in[i][row(yh, z + 1)]; }
      }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) s += c[i];
    if (YN) {
#pragma unroll
      for (int i = 0; i < 9; ++i) s += l[i] - h[i];
    }
    const size_t o = row(y, z);
    a.out[0][o] = s; a.out[1][o] = s * 2.f; a.out[2][o] = s * 3.f;
    if (more) {
#pragma unroll
      for (int i = 0; i < 10; ++i) c[i] = q[i];
      if (YN) {
#pragma unroll
        for (int i = 0; i < 9; ++i) { l[i] = nl[i]; h[i] = nh[i]; }
      }
    }
  }
}

// E: the same 52 B per voxel as C, but packed: (f0,f1,u,v) as float4, w as float, (phi,ksi) as float2, (du,dv,dw) as 3 floats
// in, 3 floats out -> 4 loads + 1 store per voxel step instead of 10 + 3
struct PArgs { const float4* c4; const float* w; const float2* p2; const float* d3; float* o3; };
struct f3 { float x, y, z; };
template <int TY>
__global__ __launch_bounds__(64 * TY) void k_march_packed(PArgs a, int W, int H, int D, int pitch, int zchunk)
{
  const int lane = threadIdx.x;
  const int y = __builtin_amdgcn_readfirstlane(int(blockIdx.y) * TY + int(threadIdx.y));
  const int x = blockIdx.x * 64 + lane;
  const int z0 = blockIdx.z * zchunk, z1 = min(z0 + zchunk, D);
  if (y >= H || x >= W) return;
  auto row = [&](int yy, int zz) { return (size_t(zz) * H + yy) * pitch + x; };
  float4 c = a.c4[row(y, z0)];
  float wv = a.w[row(y, z0)];
  float2 p = a.p2[row(y, z0)];
  f3 d = reinterpret_cast<const f3*>(a.d3)[row(y, z0)];
  for (int z = z0; z < z1; ++z) {
    const bool more = z + 1 < z1;
    float4 nc = c; float nw = wv; float2 np = p; f3 nd = d;
    if (more) {
      nc = a.c4[row(y, z + 1)];
      nw = a.w[row(y, z + 1)];
      np = a.p2[row(y, z + 1)];
      nd = reinterpret_cast<const f3*>(a.d3)[row(y, z + 1)];
    }
    const float s = c.x + c.y + c.z + c.w + wv + p.x + p.y + d.x + d.y + d.z;
    f3 o = {s, s * 2.f, s * 3.f};
    reinterpret_cast<f3*>(a.o3)[row(y, z)] = o;
    c = nc; wv = nw; p = np; d = nd;
  }
}

int main(int argc, char** argv)
{
  const size_t skew = argc > 1 ? strtoull(argv[1], nullptr, 0) : 0;  // bytes between the bases of successive arrays (mod the array size)
  const int nstream = argc > 2 ? atoi(argv[2]) : 0;
  printf("skew %zu bytes\n", skew);
  const int W = 512, H = 512, D = 512, pitch = 512;
  const size_t n = size_t(W) * H * D;
  Args a;
  std::vector<float> host(n);
  for (size_t i = 0; i < n; ++i) host[i] = float(i % 977) * 1e-3f;
  for (int i = 0; i < 10; ++i) { char* p; CK(hipMalloc(&p, n * 4 + 16 * skew + 256)); p += i * skew; CK(hipMemcpy(p, host.data(), n * 4, hipMemcpyHostToDevice)); a.in[i] = (float*)p; }
  for (int i = 0; i < 3; ++i) { char* p; CK(hipMalloc(&p, n * 4 + 16 * skew + 256)); p += (10 + i) * skew; a.out[i] = (float*)p; }
  (void)nstream;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-44s %8.3f ms  %7.1f GB/s (52 B/voxel)\n", name, ms, 52.0 * n / (ms * 1e-3) / 1e9);
    return 0;
  };
  time("A flat float4, 2048 blocks", [&] { k_flat<float4><<<2048, 256>>>(a, n / 4); });
  time("A flat float4, 8192 blocks", [&] { k_flat<float4><<<8192, 256>>>(a, n / 4); });
  time("B flat dword, 4096 blocks", [&] { k_flat<float><<<4096, 256>>>(a, n); });
  {
    PArgs pa;
    float* q;
    CK(hipMalloc(&q, n * 16 + 256)); CK(hipMemset(q, 0, n * 16)); pa.c4 = (const float4*)q;
    CK(hipMalloc(&q, n * 4 + 256)); CK(hipMemset(q, 0, n * 4)); pa.w = q;
    CK(hipMalloc(&q, n * 8 + 256)); CK(hipMemset(q, 0, n * 8)); pa.p2 = (const float2*)q;
    CK(hipMalloc(&q, n * 12 + 256)); CK(hipMemset(q, 0, n * 12)); pa.d3 = q;
    CK(hipMalloc(&q, n * 12 + 256)); pa.o3 = q;
    for (int zc : {128, 64}) {
      char nm[64];
      snprintf(nm, 64, "E packed march TY=8 zchunk=%d", zc);
      time(nm, [&] { k_march_packed<8><<<dim3(W / 64, H / 8, D / zc), dim3(64, 8)>>>(pa, W, H, D, pitch, zc); });
      snprintf(nm, 64, "E packed march TY=4 zchunk=%d", zc);
      time(nm, [&] { k_march_packed<4><<<dim3(W / 64, H / 4, D / zc), dim3(64, 4)>>>(pa, W, H, D, pitch, zc); });
    }
  }
  for (int zc : {128}) {
    char nm[64];
    snprintf(nm, 64, "C march TY=4 zchunk=%d", zc);
    time(nm, [&] { k_march<4, false><<<dim3(W / 64, H / 4, D / zc), dim3(64, 4)>>>(a, W, H, D, pitch, zc); });
    snprintf(nm, 64, "C march TY=8 zchunk=%d", zc);
    time(nm, [&] { k_march<8, false><<<dim3(W / 64, H / 8, D / zc), dim3(64, 8)>>>(a, W, H, D, pitch, zc); });
    snprintf(nm, 64, "D march+yn TY=4 zchunk=%d", zc);
    time(nm, [&] { k_march<4, true><<<dim3(W / 64, H / 4, D / zc), dim3(64, 4)>>>(a, W, H, D, pitch, zc); });
    snprintf(nm, 64, "D march+yn TY=8 zchunk=%d", zc);
    time(nm, [&] { k_march<8, true><<<dim3(W / 64, H / 8, D / zc), dim3(64, 8)>>>(a, W, H, D, pitch, zc); });
  }
  return 0;
}
