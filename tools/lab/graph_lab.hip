// Does a captured graph shorten the distance between dependent small kernels?  (BASELINE configs 2 and 3 are chains of ~4 800 and
// ~1 200 dependent launches of 6 - 30 us: DESIGN.md section 6.)  N dependent launches of a kernel that keeps W workgroups busy for
// roughly T microseconds, (a) issued into a stream one by one, (b) captured once into a graph and replayed.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      std::fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      std::exit(1);                                                                     \
    }                                                                                   \
  } while (0)

__global__ __launch_bounds__(512) void k_link(const float* __restrict__ in, float* __restrict__ out, int n, int spin)
{
  // every workgroup reads what the previous launch wrote (a real dependency) and works for `spin` dependent steps
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = in[i % n];
  for (int k = 0; k < spin; ++k) v = v * 1.0000001f + 0.5f;
  out[i % n] = v;
}

static double run_stream(hipStream_t s, float* a, float* b, int n, int wgs, int spin, int launches)
{
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < launches; ++i) {
    hipLaunchKernelGGL(k_link, dim3(wgs), dim3(512), 0, s, i & 1 ? b : a, i & 1 ? a : b, n, spin);
  }
  CHECK(hipStreamSynchronize(s));
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches;
}

int main()
{
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int n = 1 << 20;
  float *a, *b;
  CHECK(hipMalloc(&a, n * sizeof(float)));
  CHECK(hipMalloc(&b, n * sizeof(float)));
  CHECK(hipMemset(a, 0, n * sizeof(float)));
  CHECK(hipMemset(b, 0, n * sizeof(float)));
  const int launches = 1000;
  for (int wgs : {8, 64, 256}) {
    for (int spin : {0, 400, 2000}) {
      run_stream(s, a, b, n, wgs, spin, 50);
      const double one = run_stream(s, a, b, n, wgs, 1, 1);   // (warm)
      (void)one;
      const double t_stream = run_stream(s, a, b, n, wgs, spin, launches);
      // the same chain as a graph
      hipGraph_t graph;
      hipGraphExec_t exec;
      CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < launches; ++i)
        hipLaunchKernelGGL(k_link, dim3(wgs), dim3(512), 0, s, i & 1 ? b : a, i & 1 ? a : b, n, spin);
      CHECK(hipStreamEndCapture(s, &graph));
      auto i0 = std::chrono::steady_clock::now();
      CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      const double t_inst = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - i0).count();
      CHECK(hipGraphLaunch(exec, s));
      CHECK(hipStreamSynchronize(s));
      double best = 1e30;
      for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        CHECK(hipGraphLaunch(exec, s));
        CHECK(hipStreamSynchronize(s));
        const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches;
        if (t < best) best = t;
      }
      std::printf("%3d workgroups, %4d steps: stream %6.2f us per launch   graph %6.2f us per launch   (instantiate %.0f us for %d nodes)\n",
                  wgs, spin, t_stream, best, t_inst, launches);
      CHECK(hipGraphExecDestroy(exec));
      CHECK(hipGraphDestroy(graph));
    }
  }
  return 0;
}
