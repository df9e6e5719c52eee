// Generator of the translated-Gaussian pair used by configs C4/C5: splitmix64, seed 20241003,
// uniform doubles from the top 53 bits drawn in the order (cx, cy, cz, sigma, a) per blob; centres in
// [0.15, 0.85] of each axis, sigma in [m/32, m/12] with m the mean edge length, amplitude in [0.3, 1.0].
// The Gaussian is evaluated separably (three 1-D tables per blob), so a 512^3 pair takes seconds on the host.
#include "synth.h"

#include <cmath>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace f3d_synth {

namespace {

struct SplitMix64 {
  unsigned long long state;
  unsigned long long Next()
  {
    unsigned long long z = (state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
  double Uniform() { return static_cast<double>(Next() >> 11) * (1.0 / 9007199254740992.0); }
  double Uniform(double lo, double hi) { return lo + (hi - lo) * Uniform(); }
};

struct Blob {
  double c[3];
  double sigma;
  double amp;
};

// out[k * n + i] = exp(-(i - centre_k)^2 / (2 sigma_k^2))
std::vector<double> AxisTable(const std::vector<Blob>& blobs, int axis, size_t n, double shift)
{
  std::vector<double> t(blobs.size() * n);
  for (size_t k = 0; k < blobs.size(); ++k) {
    const double c = blobs[k].c[axis] + shift;
    const double inv = 1.0 / (2.0 * blobs[k].sigma * blobs[k].sigma);
    for (size_t i = 0; i < n; ++i) {
      const double d = static_cast<double>(i) - c;
      t[k * n + i] = std::exp(-d * d * inv);
    }
  }
  return t;
}

// unscaled field; returns its maximum
float Render(const std::vector<Blob>& blobs, size_t w, size_t h, size_t d, const double shift[3], float* out, size_t z_lo,
             size_t z_hi)
{
  const std::vector<double> tx = AxisTable(blobs, 0, w, shift[0]);
  const std::vector<double> ty = AxisTable(blobs, 1, h, shift[1]);
  const std::vector<double> tz = AxisTable(blobs, 2, d, shift[2]);
  const size_t nb = blobs.size();
  float vmax = 0.f;
#pragma omp parallel for schedule(static) reduction(max : vmax)
  for (long long row = static_cast<long long>(z_lo * h); row < static_cast<long long>(z_hi * h); ++row) {
    const size_t y = static_cast<size_t>(row) % h, z = static_cast<size_t>(row) / h;
    std::vector<double> acc(w, 0.0);
    for (size_t k = 0; k < nb; ++k) {
      const double coef = blobs[k].amp * ty[k * h + y] * tz[k * d + z];
      const double* ex = &tx[k * w];
      for (size_t x = 0; x < w; ++x) acc[x] += coef * ex[x];
    }
    float* dst = out + static_cast<size_t>(row) * w;
    for (size_t x = 0; x < w; ++x) {
      dst[x] = static_cast<float>(acc[x]);
      if (dst[x] > vmax) vmax = dst[x];
    }
  }
  return vmax;
}

}  // namespace

namespace {
std::vector<Blob> MakeBlobs(size_t w, size_t h, size_t d)
{
  SplitMix64 rng{kSeed};
  const double dims[3] = {static_cast<double>(w), static_cast<double>(h), static_cast<double>(d)};
  const double m = (dims[0] + dims[1] + dims[2]) / 3.0;
  std::vector<Blob> blobs(kBlobs);
  for (Blob& b : blobs) {
    for (int a = 0; a < 3; ++a) b.c[a] = rng.Uniform(0.15, 0.85) * dims[a];
    b.sigma = rng.Uniform(m / 32.0, m / 12.0);
    b.amp = rng.Uniform(0.3, 1.0);
  }
  return blobs;
}

void LimitThreads()
{
#ifdef _OPENMP
  if (omp_get_max_threads() > 16) omp_set_num_threads(16);  // stay within the per-GPU CPU share of a shared box
#endif
}
}  // namespace

float TranslatedGaussianPlanes(size_t w, size_t h, size_t d, size_t z_lo, size_t z_hi, float* frame_0, float* frame_1)
{
  LimitThreads();
  const std::vector<Blob> blobs = MakeBlobs(w, h, d);
  const double none[3] = {0.0, 0.0, 0.0};
  const float vmax = Render(blobs, w, h, d, none, frame_0, z_lo, z_hi);
  Render(blobs, w, h, d, kShift, frame_1, z_lo, z_hi);
  return vmax;
}

void TranslatedGaussianPair(size_t w, size_t h, size_t d, float* frame_0, float* frame_1)
{
  const float vmax = TranslatedGaussianPlanes(w, h, d, 0, d, frame_0, frame_1);
  const float s = vmax > 0.f ? 255.f / vmax : 1.f;
  const size_t count = w * h * d;
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < static_cast<long long>(count); ++i) {
    frame_0[i] *= s;
    frame_1[i] *= s;
  }
}

}  // namespace f3d_synth
