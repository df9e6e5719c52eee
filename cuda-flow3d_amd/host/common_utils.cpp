#include "common_utils.h"

#include <unistd.h>

#include <cstring>

namespace Utils {

void GetExecutablePath(char* path, size_t size)
{
  if (size == 0) return;
  ssize_t n = readlink("/proc/self/exe", path, size - 1);
  if (n < 0) n = 0;
  path[n] = '\0';
  if (char* slash = std::strrchr(path, '/')) *slash = '\0';
}

void PrintProgressBar(float complete)
{
  constexpr int kCells = 40;
  char bar[kCells + 3];
  bar[0] = '[';
  for (int i = 0; i < kCells; ++i) bar[1 + i] = (i / static_cast<float>(kCells) < complete) ? '=' : ' ';
  bar[kCells + 1] = ']';
  bar[kCells + 2] = '\0';
  std::printf("\r%s", bar);
}

}  // namespace Utils
