// Host data types of the drop-in surface, re-implemented from the behaviour of the reference's
// src/data_types/{data_structs.h, operation_parameters.{h,cpp}, data3d.{h,cpp}}.
#ifndef F3D_HOST_DATA_TYPES_H_
#define F3D_HOST_DATA_TYPES_H_

#include <cstddef>
#include <string>
#include <unordered_map>

#include "f3d.h"

// Layout-identical to the reference struct (src/data_types/data_structs.h:20-25) and to f3d_size4.
struct DataSize4 {
  size_t width;
  size_t height;
  size_t depth;
  size_t pitch;
};
static_assert(sizeof(DataSize4) == sizeof(f3d_size4), "DataSize4 must stay a 4 x size_t record");

struct Stat3 {  // src/data_types/data_structs.h:29-33
  float min;
  float max;
  float avg;
};

// Device pointers travel through the parameter bag as 64-bit integers (the reference's CUdeviceptr).
typedef f3d_devptr DevicePtr;

// String-keyed bag of NON-OWNING pointers to caller variables (src/data_types/operation_parameters.{h,cpp}):
// the first push under a key wins, a missing key reads as nullptr, values are typed at the read site.
class OperationParameters {
 public:
  OperationParameters() = default;
  bool PushValuePtr(std::string key, void* value_ptr) { return map_.emplace(std::move(key), value_ptr).second; }
  void* GetValuePtr(std::string key) const
  {
    auto it = map_.find(key);
    return it == map_.end() ? nullptr : it->second;
  }
  void Clear() { map_.clear(); }

 private:
  std::unordered_map<std::string, void*> map_;
};

// Dense host volume, x fastest then y then z (src/data_types/data3d.h:22-62).
class Data3D {
 public:
  Data3D() = default;
  Data3D(size_t width, size_t height, size_t depth);
  // non-owning view of caller memory (language bindings hand their own buffers to ComputeFlow)
  Data3D(float* external, size_t width, size_t height, size_t depth)
      : data_(external), width_(width), height_(height), depth_(depth), owns_(false) {}
  Data3D(const Data3D&) = delete;
  Data3D& operator=(const Data3D&) = delete;
  ~Data3D();

  size_t Width() const { return width_; }
  size_t Height() const { return height_; }
  size_t Depth() const { return depth_; }
  float* DataPtr() { return data_; }
  const float* DataPtr() const { return data_; }
  float& Data(size_t x, size_t y, size_t z) { return data_[(z * height_ + y) * width_ + x]; }

  bool Allocate(size_t width, size_t height, size_t depth);  // (re)allocate owned storage
  void Swap(Data3D& other);
  void ZeroData();

  bool ReadRAWFromFileU8(const char* filename, size_t width, size_t height, size_t depth);
  bool ReadRAWFromFileF32(const char* filename, size_t width, size_t height, size_t depth);
  bool WriteRAWToFileU8(const char* filename) const;
  bool WriteRAWToFileF32(const char* filename) const;
  static bool WriteFlowToFileVTK(const char* filename, const Data3D& flow_u, const Data3D& flow_v, const Data3D& flow_w);

 private:
  void Release();

  float* data_ = nullptr;
  size_t width_ = 0;
  size_t height_ = 0;
  size_t depth_ = 0;
  bool owns_ = true;
};

#endif
