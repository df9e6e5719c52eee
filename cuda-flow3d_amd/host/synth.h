// Deterministic synthetic benchmark input: a pair of smooth volumes related by a known sub-voxel translation.
#ifndef F3D_HOST_SYNTH_H_
#define F3D_HOST_SYNTH_H_

#include <cstddef>

namespace f3d_synth {

constexpr int kBlobs = 64;
constexpr unsigned long long kSeed = 20241003ULL;
constexpr double kShift[3] = {2.0, -1.0, 0.5};  // frame_1(p) = frame_0(p - t); expected flow (u, v, w) ~ +t

// Fills two dense width*height*depth float volumes (x fastest).  frame_0 is a sum of kBlobs Gaussian blobs
// scaled so that its maximum is 255; frame_1 is the same field evaluated with the centres moved by kShift.
void TranslatedGaussianPair(size_t width, size_t height, size_t depth, float* frame_0, float* frame_1);

// Slab form for multi-process runs: renders only planes [z_lo, z_hi) of both frames, UNSCALED, into the full-size
// arrays and returns the maximum of frame_0 over those planes; the caller reduces the maxima over all slabs and
// multiplies by 255 / max (exactly what the whole-volume function does).
float TranslatedGaussianPlanes(size_t width, size_t height, size_t depth, size_t z_lo, size_t z_hi, float* frame_0,
                               float* frame_1);

}  // namespace f3d_synth

#endif
