// Lab: memory-side cost of the sweep's tiling choices with a synthetic VALU load (NF=16 mul+add x8, ND=6 div).
//  XOV  : 62-wide overlapping x tiles (unaligned 256-B row loads) vs aligned 64-wide
//  HALO : the two edge waves of the 8-wave workgroup stream one extra row each (y halo), 9 arrays
//  XCD  : XCD-contiguous tile order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Args { const float* in[10]; float* out[3]; };
constexpr int TY = 8;

template <bool XOV, bool HALO, bool XCD, int NF, int ND, int XH = 0>
__global__ __launch_bounds__(64 * TY, 4) void k(Args a, int W, int H, int D, int pitch, int zchunk, int ntx, int nty, int ntiles)
{
  __shared__ float sh[2][9][TY + 2][64];
  int tile = blockIdx.x;
  if (XCD) { const int per = (ntiles + 7) / 8; tile = (tile % 8) * per + tile / 8; }
  if (tile >= ntiles) return;
  const int tx = tile % ntx, ty = (tile / ntx) % nty, tz = tile / (ntx * nty);
  const int lane = threadIdx.x;
  const int r = __builtin_amdgcn_readfirstlane(int(threadIdx.y));
  const int y = min(ty * TY + r, H - 1);
  int x = XOV ? tx * 62 - 1 + lane : tx * 64 + lane;
  x = min(max(x, 0), W - 1);
  const bool owner = XOV ? (lane >= 1 && lane <= 62) : true;
  const int z0 = tz * zchunk, z1 = min(z0 + zchunk, D);
  const bool edge = HALO && (r == 0 || r == TY - 1);
  const int yh = min(max(r == 0 ? ty * TY - 1 : ty * TY + TY, 0), H - 1);
  const int hslot = r == 0 ? 0 : TY + 1;
  float c[10], q[10], hc[9], hq[9], xc[9], xq[9];
  const int xh = min(max(lane < 32 ? tx * 64 - 1 : tx * 64 + 64, 0), W - 1);
  auto rowh = [&](int yy, int zz) { return (size_t(min(zz, D - 1)) * H + yy) * pitch + xh; };
  auto row = [&](int yy, int zz) { return (size_t(min(zz, D - 1)) * H + yy) * pitch + x; };
#pragma unroll
  for (int i = 0; i < 10; ++i) c[i] = a.in[i][row(y, z0)];
  if (edge) {
#pragma unroll
    for (int i = 0; i < 9; ++i) hc[i] = a.in[i][row(yh, z0)];
  }
  if (XH == 1) {
#pragma unroll
    for (int i = 0; i < 9; ++i) xc[i] = a.in[i][rowh(y, z0)];
  }
  for (int z = z0; z < z1; ++z) {
#pragma unroll
    for (int i = 0; i < 10; ++i) q[i] = a.in[i][row(y, z + 1)];
    if (XH == 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) xq[i] = a.in[i][rowh(y, z + 1)];
    }
    if (XH == 2) {
#pragma unroll
      for (int i = 0; i < 9; ++i) xc[i] = a.in[i][rowh(y, z)];
    }
    if (edge) {
#pragma unroll
      for (int i = 0; i < 9; ++i) hq[i] = a.in[i][row(yh, z + 1)];
    }
    const int b = z & 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) sh[b][i][r + 1][lane] = c[i];
    if (edge) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sh[b][i][hslot][lane] = hc[i];
    }
    __syncthreads();
    float yn = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) yn += sh[b][i][r][lane] - sh[b][i][r + 2][lane];
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = c[i];
#pragma unroll
    for (int it = 0; it < NF; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = acc[i] * c[8] + c[9];
    float s = yn;
    if (XH) {
#pragma unroll
      for (int i = 0; i < 9; ++i) s += xc[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
#pragma unroll
    for (int d = 0; d < ND; ++d) s = (s + c[d]) / (acc[d] + 3.f);
    if (owner) {
      const size_t o = row(y, z);
      a.out[0][o] = s; a.out[1][o] = s * 2.f; a.out[2][o] = s * 3.f;
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) c[i] = q[i];
    if (XH == 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) xc[i] = xq[i];
    }
    if (edge) {
#pragma unroll
      for (int i = 0; i < 9; ++i) hc[i] = hq[i];
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
    printf("%-40s %8.3f ms\n", name, ms);
    return 0;
  };
  for (int zc : {32, 64}) {
    printf("zchunk %d\n", zc);
#define RUN(XOV, HALO, XCD, NF, ND, XH) { const int ntx = XOV ? (W + 61) / 62 : W / 64, nty = H / TY, nz = D / zc, nt = ntx * nty * nz; \
    const int blocks = XCD ? ((nt + 7) / 8) * 8 : nt; \
    time("XOV=" #XOV " HALO=" #HALO " XCD=" #XCD " NF=" #NF " ND=" #ND " XH=" #XH, [&] { k<XOV, HALO, XCD, NF, ND, XH><<<blocks, dim3(64, TY)>>>(a, W, H, D, pitch, zc, ntx, nty, nt); }); }
    RUN(false, false, true, 16, 6, 0);
    RUN(false, true, true, 16, 6, 0);
    RUN(false, true, true, 16, 6, 1);
    RUN(false, true, true, 16, 6, 2);
    RUN(false, true, false, 16, 6, 1);
    RUN(false, true, true, 24, 6, 1);
    RUN(false, true, true, 0, 0, 1);
  }
  return 0;
}
