// Host volume container and RAW / VTK file formats; behaviour of src/data_types/data3d.cpp
// (U8 reader :95-140, F32 reader :142-181, U8 writer :183-212, F32 writer :214-237, VTK :239-264).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "data_types.h"

Data3D::Data3D(size_t width, size_t height, size_t depth) { Allocate(width, height, depth); }

Data3D::~Data3D() { Release(); }

void Data3D::Release()
{
  if (owns_) delete[] data_;
  owns_ = true;
  data_ = nullptr;
  width_ = height_ = depth_ = 0;
}

bool Data3D::Allocate(size_t width, size_t height, size_t depth)
{
  // same extent, own storage: keep it (a frame sequence reads file after file into the same page-locked buffer)
  if (data_ && owns_ && width == width_ && height == height_ && depth == depth_) return true;
  Release();
  data_ = new (std::nothrow) float[width * height * depth];
  if (!data_) {
    std::printf("Error. Cannot allocate memory on the host.\n");
    return false;
  }
  width_ = width;
  height_ = height;
  depth_ = depth;
  return true;
}

void Data3D::Swap(Data3D& other)
{
  if (width_ == other.width_ && height_ == other.height_ && depth_ == other.depth_) {
    std::swap(data_, other.data_);
    std::swap(owns_, other.owns_);
  } else {
    std::printf("Error. Cannot swap two Data3D objects (wrong dimensions).\n");
  }
}

void Data3D::ZeroData()
{
  if (data_) std::memset(data_, 0, width_ * height_ * depth_ * sizeof(float));
}

namespace {
struct FileCloser {
  void operator()(std::FILE* f) const { if (f) std::fclose(f); }
};
using File = std::unique_ptr<std::FILE, FileCloser>;

// the file must hold exactly the volume: one more byte is "wrong dimensions" (data3d.cpp:124-131)
bool at_end(std::FILE* f)
{
  unsigned char probe;
  return std::fread(&probe, 1, 1, f) == 0;
}
}  // namespace

bool Data3D::ReadRAWFromFileU8(const char* filename, size_t width, size_t height, size_t depth)
{
  File file(std::fopen(filename, "rb"));
  if (!file) {
    std::printf("Cannot open file '%s'.\n", filename);
    return false;
  }
  if (!Allocate(width, height, depth)) return false;
  std::vector<unsigned char> row(width);
  bool ok = true;
  for (size_t r = 0; ok && r < height * depth; ++r) {
    ok = std::fread(row.data(), 1, width, file.get()) == width;
    if (ok) {
      float* dst = data_ + r * width;
      for (size_t x = 0; x < width; ++x) dst[x] = static_cast<float>(row[x]);
    }
  }
  if (!ok || !at_end(file.get())) {
    std::printf("Error reading RAW data from file '%s': wrong dimensions.", filename);
    Release();
    return false;
  }
  return true;
}

bool Data3D::ReadRAWFromFileF32(const char* filename, size_t width, size_t height, size_t depth)
{
  File file(std::fopen(filename, "rb"));
  if (!file) {
    std::printf("Cannot open file '%s'.\n", filename);
    return false;
  }
  if (!Allocate(width, height, depth)) return false;
  const size_t count = width * height * depth;
  if (std::fread(data_, sizeof(float), count, file.get()) != count || !at_end(file.get())) {
    std::printf("Error reading RAW data from file '%s': wrong dimensions.", filename);
    Release();
    return false;
  }
  return true;
}

bool Data3D::WriteRAWToFileU8(const char* filename) const
{
  File file(std::fopen(filename, "wb"));
  if (!file) {
    std::printf("Cannot open file '%s'.\n", filename);
    return false;
  }
  std::vector<unsigned char> row(width_);
  for (size_t r = 0; r < height_ * depth_; ++r) {
    const float* src = data_ + r * width_;
    for (size_t x = 0; x < width_; ++x)  // clamp to [0, 255], truncate (data3d.cpp:192-193)
      row[x] = static_cast<unsigned char>(std::min(255.f, std::max(0.f, src[x])));
    if (std::fwrite(row.data(), 1, width_, file.get()) != width_) {
      std::printf("Error writing RAW data to file '%s'.", filename);
      return false;
    }
  }
  return true;
}

bool Data3D::WriteRAWToFileF32(const char* filename) const
{
  File file(std::fopen(filename, "wb"));
  if (!file) {
    std::printf("Cannot open file '%s'.\n", filename);
    return false;
  }
  const size_t count = width_ * height_ * depth_;
  if (std::fwrite(data_, sizeof(float), count, file.get()) != count) {
    std::printf("Error writing RAW data to file '%s'.", filename);
    return false;
  }
  return true;
}

bool Data3D::WriteFlowToFileVTK(const char* filename, const Data3D& u, const Data3D& v, const Data3D& w)
{
  File file(std::fopen(filename, "wb"));
  if (!file) {
    std::printf("Cannot open file '%s'.\n", filename);
    return false;
  }
  std::FILE* f = file.get();
  const size_t count = u.width_ * u.height_ * u.depth_;
  // header text of data3d.cpp:243-251 (host-endian binary payload, interleaved u v w)
  std::fprintf(f, "# vtk DataFile Version 2.0\n");
  std::fprintf(f, "3D Vector field computed by GpuFlow3D\n");
  std::fprintf(f, "BINARY\n");
  std::fprintf(f, "DATASET STRUCTURED_POINTS\n");
  std::fprintf(f, "DIMENSIONS %zu %zu %zu\n", u.width_, u.height_, u.depth_);
  std::fprintf(f, "ORIGIN 0 0 0\n");
  std::fprintf(f, "SPACING 1 1 1\n");
  std::fprintf(f, "POINT_DATA %zu\n", count);
  std::fprintf(f, "VECTORS vectors float\n");
  std::vector<float> line(3 * u.width_);
  for (size_t r = 0; r < u.height_ * u.depth_; ++r) {
    for (size_t x = 0; x < u.width_; ++x) {
      line[3 * x + 0] = u.data_[r * u.width_ + x];
      line[3 * x + 1] = v.data_[r * u.width_ + x];
      line[3 * x + 2] = w.data_[r * u.width_ + x];
    }
    if (std::fwrite(line.data(), sizeof(float), line.size(), f) != line.size()) return false;
  }
  return true;
}
