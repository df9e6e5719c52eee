// Parameter-bag read macros and console helpers (behaviour of src/utils/common_utils.{h,cpp}).
#ifndef F3D_HOST_COMMON_UTILS_H_
#define F3D_HOST_COMMON_UTILS_H_

#include <cstddef>
#include <cstdio>
#include <initializer_list>

namespace Utils {
void GetExecutablePath(char* path, size_t size);  // directory of /proc/self/exe (common_utils.cpp:30-47)
void PrintProgressBar(float complete);            // "\r[====    ]", 40 cells (common_utils.cpp:49-63)
}  // namespace Utils

// Typed read from the bag; on a missing key print the reference's message and return
// (src/utils/common_utils.h:29-60).  They expand inside members that have GetName().
#define F3D_PARAM_LOOKUP_(P, N, ON_MISS)                                              \
  void* v_ptr = (P).GetValuePtr((N));                                                 \
  if (!v_ptr) {                                                                       \
    std::printf("Operation: '%s'. Missing parameter '%s'.\n", GetName(), (N));        \
    ON_MISS;                                                                          \
  }

#define GET_PARAM_OR_RETURN(P, T, V, N)            \
  do {                                             \
    F3D_PARAM_LOOKUP_(P, N, return)                \
    (V) = *static_cast<T*>(v_ptr);                 \
  } while (0)

#define GET_PARAM_OR_RETURN_VALUE(P, T, V, N, R)   \
  do {                                             \
    F3D_PARAM_LOOKUP_(P, N, return (R))            \
    (V) = *static_cast<T*>(v_ptr);                 \
  } while (0)

#define GET_PARAM_PTR_OR_RETURN(P, T, PTR, N)      \
  do {                                             \
    F3D_PARAM_LOOKUP_(P, N, return)                \
    (PTR) = static_cast<T*>(v_ptr);                \
  } while (0)

// An operator's interface is its key strings.  The drivers describe every call as a LIST of {key, pointer to a
// variable of theirs}; FillBag empties the bag and pushes the list in order ("first push wins", operation_parameters.cpp:23-30).
// The bag owns nothing: the variables must outlive the Execute() call.
struct BagEntry {
  const char* key;
  void* value;
};
template <typename Bag>
inline Bag& FillBag(Bag& bag, std::initializer_list<BagEntry> entries)
{
  bag.Clear();
  for (const BagEntry& e : entries) bag.PushValuePtr(e.key, e.value);
  return bag;
}

#endif
