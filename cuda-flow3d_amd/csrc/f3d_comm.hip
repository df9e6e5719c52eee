// z-slab communication layer: plane pack/unpack kernels and neighbour exchange on RCCL (ncclSend/ncclRecv over
// xGMI).  The reference is single-GPU; this is new design.  librccl.so.1 is opened lazily with
// dlopen so that single-GPU users never load it.  All traffic runs on the library stream, in order with the kernels.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "f3d_internal.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1;
  float* d_scalar = nullptr;
} R;

int load_rccl()
{
  if (R.handle) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    R.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (R.handle) break;
  }
  if (!R.handle) return f3d::fail("f3d_comm: cannot load librccl (%s)", dlerror());
#define F3D_SYM(field, name)                                                       \
  R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.handle, name));            \
  if (!R.field) return f3d::fail("f3d_comm: librccl lacks %s", name)
  F3D_SYM(GetUniqueId, "ncclGetUniqueId");
  F3D_SYM(CommInitRank, "ncclCommInitRank");
  F3D_SYM(CommDestroy, "ncclCommDestroy");
  F3D_SYM(GroupStart, "ncclGroupStart");
  F3D_SYM(GroupEnd, "ncclGroupEnd");
  F3D_SYM(Send, "ncclSend");
  F3D_SYM(Recv, "ncclRecv");
  F3D_SYM(AllReduce, "ncclAllReduce");
  F3D_SYM(GetErrorString, "ncclGetErrorString");
#undef F3D_SYM
  return 0;
}

#define F3D_NCCL(call)                                                                                   \
  do {                                                                                                   \
    ncclResult_t r_ = (call);                                                                            \
    if (r_ != ncclSuccess) return f3d::fail("RCCL error %d (%s) in %s", static_cast<int>(r_), R.GetErrorString(r_), #call); \
  } while (0)

// dense[(p * height + y) * width + x] <-> container[((plane0 + p) * Hc + y) * pitch + x]
template <bool PACK>
__global__ __launch_bounds__(256) void k_planes(float* __restrict__ field, float* __restrict__ dense, int plane0, int width,
                                                int height, int Hc, int pitch)
{
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int p = blockIdx.z;
  if (x >= width || y >= height) return;
  const size_t c = (static_cast<size_t>(plane0 + p) * Hc + y) * pitch + x;
  const size_t d = (static_cast<size_t>(p) * height + y) * width + x;
  if (PACK) dense[d] = field[c];
  else field[c] = dense[d];
}

// One launch for a whole exchange: up to kMaxSeg (field, plane range) segments, each packed densely at its offset.
constexpr int kMaxSeg = 32;
struct SegTable {
  float* field[kMaxSeg];
  int plane0[kMaxSeg];
  int count[kMaxSeg];
  unsigned long long offset[kMaxSeg];  // floats into the dense buffer
  int n;
};

template <bool PACK>
__global__ __launch_bounds__(256) void k_planes_batched(SegTable t, float* __restrict__ dense, int width, int height, int Hc,
                                                        int pitch, int max_count)
{
  const int seg = blockIdx.z / max_count;
  const int p = blockIdx.z - seg * max_count;
  if (p >= t.count[seg]) return;
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  if (x >= width || y >= height) return;
  float* field = t.field[seg];
  const size_t c = (static_cast<size_t>(t.plane0[seg] + p) * Hc + y) * pitch + x;
  const size_t d = t.offset[seg] + (static_cast<size_t>(p) * height + y) * width + x;
  if (PACK) dense[d] = field[c];
  else field[c] = dense[d];
}

__global__ __launch_bounds__(256) void k_copy_planes(float* __restrict__ dst, const float* __restrict__ src, int dst_plane0,
                                                     int src_plane0, int width, int height, int Hc, int pitch)
{
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int p = blockIdx.z;
  if (x >= width || y >= height) return;
  const size_t row = static_cast<size_t>(y) * pitch + x;
  const size_t plane = static_cast<size_t>(Hc) * pitch;
  dst[(dst_plane0 + p) * plane + row] = src[(src_plane0 + p) * plane + row];
}

int check_planes(int plane0, int count, size_t width, size_t height, const char* who)
{
  const f3d_size4& c = f3d::container();
  if (c.pitch == 0) return f3d::fail("%s: f3d_set_container() has not been called", who);
  if (count < 0 || plane0 < 0 || static_cast<size_t>(plane0 + count) > c.depth || width > c.width || height > c.height)
    return f3d::fail("%s: planes [%d,%d) x %zux%zu outside the %zux%zux%zu container", who, plane0, plane0 + count, width,
                     height, c.width, c.height, c.depth);
  return 0;
}

}  // namespace

extern "C" {

int f3d_comm_unique_id(void* id128)
{
  if (!id128) return f3d::fail("f3d_comm_unique_id: null argument");
  if (load_rccl()) return 1;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclUniqueId id;
  F3D_NCCL(R.GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}

int f3d_comm_init(const void* id128, int rank, int n_ranks)
{
  F3D_REQUIRE_READY("f3d_comm_init");
  if (!id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return f3d::fail("f3d_comm_init: bad arguments");
  if (R.comm) return f3d::fail("f3d_comm_init: communicator already initialised");
  if (load_rccl()) return 1;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  F3D_NCCL(R.CommInitRank(&R.comm, n_ranks, id, rank));
  R.rank = rank;
  R.n_ranks = n_ranks;
  F3D_HIP(hipMalloc(reinterpret_cast<void**>(&R.d_scalar), sizeof(float)));
  return 0;
}

int f3d_comm_destroy(void)
{
  if (R.comm) {
    (void)hipStreamSynchronize(f3d::stream());
    (void)R.CommDestroy(R.comm);
    R.comm = nullptr;
  }
  if (R.d_scalar) {
    (void)hipFree(R.d_scalar);
    R.d_scalar = nullptr;
  }
  R.rank = 0;
  R.n_ranks = 1;
  return 0;
}

int f3d_comm_rank(int* rank, int* n_ranks)
{
  if (rank) *rank = R.rank;
  if (n_ranks) *n_ranks = R.n_ranks;
  return 0;
}

int f3d_pack_planes(f3d_devptr field, int plane0, int count, size_t width, size_t height, f3d_devptr staging,
                    size_t offset_floats)
{
  F3D_REQUIRE_READY("f3d_pack_planes");
  if (check_planes(plane0, count, width, height, "f3d_pack_planes")) return 1;
  if (count == 0) return 0;
  const f3d_size4& c = f3d::container();
  const dim3 grid((width + 63) / 64, (height + 3) / 4, count), block(64, 4, 1);
  hipLaunchKernelGGL(k_planes<true>, grid, block, 0, f3d::stream(), f3d_ptr<float>(field),
                     f3d_ptr<float>(staging) + offset_floats, plane0, static_cast<int>(width), static_cast<int>(height),
                     static_cast<int>(c.height), static_cast<int>(c.pitch / sizeof(float)));
  F3D_HIP(hipGetLastError());
  return 0;
}

int f3d_unpack_planes(f3d_devptr field, int plane0, int count, size_t width, size_t height, f3d_devptr staging,
                      size_t offset_floats)
{
  F3D_REQUIRE_READY("f3d_unpack_planes");
  if (check_planes(plane0, count, width, height, "f3d_unpack_planes")) return 1;
  if (count == 0) return 0;
  const f3d_size4& c = f3d::container();
  const dim3 grid((width + 63) / 64, (height + 3) / 4, count), block(64, 4, 1);
  hipLaunchKernelGGL(k_planes<false>, grid, block, 0, f3d::stream(), f3d_ptr<float>(field),
                     f3d_ptr<float>(staging) + offset_floats, plane0, static_cast<int>(width), static_cast<int>(height),
                     static_cast<int>(c.height), static_cast<int>(c.pitch / sizeof(float)));
  F3D_HIP(hipGetLastError());
  return 0;
}

static int planes_batched(bool pack, const f3d_devptr* fields, const int* plane0, const int* count, const size_t* offset,
                          int n_segments, size_t width, size_t height, f3d_devptr staging, const char* who)
{
  F3D_REQUIRE_READY(who);
  if (n_segments < 0 || n_segments > kMaxSeg) return f3d::fail("%s: at most %d segments per call", who, kMaxSeg);
  SegTable t;
  t.n = n_segments;
  int max_count = 0;
  for (int i = 0; i < n_segments; ++i) {
    if (check_planes(plane0[i], count[i], width, height, who)) return 1;
    t.field[i] = f3d_ptr<float>(fields[i]);
    t.plane0[i] = plane0[i];
    t.count[i] = count[i];
    t.offset[i] = offset[i];
    if (count[i] > max_count) max_count = count[i];
  }
  if (n_segments == 0 || max_count == 0) return 0;
  const f3d_size4& c = f3d::container();
  const dim3 grid((width + 63) / 64, (height + 3) / 4, n_segments * max_count), block(64, 4, 1);
  float* dense = f3d_ptr<float>(staging);
  const int w = static_cast<int>(width), h = static_cast<int>(height), hc = static_cast<int>(c.height),
            pf = static_cast<int>(c.pitch / sizeof(float));
  if (pack) hipLaunchKernelGGL(k_planes_batched<true>, grid, block, 0, f3d::stream(), t, dense, w, h, hc, pf, max_count);
  else hipLaunchKernelGGL(k_planes_batched<false>, grid, block, 0, f3d::stream(), t, dense, w, h, hc, pf, max_count);
  F3D_HIP(hipGetLastError());
  return 0;
}

int f3d_pack_segments(const f3d_devptr* fields, const int* plane0, const int* count, const size_t* offset_floats,
                      int n_segments, size_t width, size_t height, f3d_devptr staging)
{
  return planes_batched(true, fields, plane0, count, offset_floats, n_segments, width, height, staging, "f3d_pack_segments");
}

int f3d_unpack_segments(const f3d_devptr* fields, const int* plane0, const int* count, const size_t* offset_floats,
                        int n_segments, size_t width, size_t height, f3d_devptr staging)
{
  return planes_batched(false, fields, plane0, count, offset_floats, n_segments, width, height, staging, "f3d_unpack_segments");
}

int f3d_copy_planes(f3d_devptr dst, int dst_plane0, f3d_devptr src, int src_plane0, int count, size_t width, size_t height)
{
  F3D_REQUIRE_READY("f3d_copy_planes");
  if (check_planes(dst_plane0, count, width, height, "f3d_copy_planes") ||
      check_planes(src_plane0, count, width, height, "f3d_copy_planes"))
    return 1;
  if (count == 0) return 0;
  const f3d_size4& c = f3d::container();
  const dim3 grid((width + 63) / 64, (height + 3) / 4, count), block(64, 4, 1);
  hipLaunchKernelGGL(k_copy_planes, grid, block, 0, f3d::stream(), f3d_ptr<float>(dst), f3d_ptr<const float>(src), dst_plane0,
                     src_plane0, static_cast<int>(width), static_cast<int>(height), static_cast<int>(c.height),
                     static_cast<int>(c.pitch / sizeof(float)));
  F3D_HIP(hipGetLastError());
  return 0;
}

int f3d_comm_sendrecv(f3d_devptr send_buf, const size_t* send_offset, const size_t* send_count, f3d_devptr recv_buf,
                      const size_t* recv_offset, const size_t* recv_count, const int* peers, int n_peers)
{
  F3D_REQUIRE_READY("f3d_comm_sendrecv");
  if (!R.comm) return f3d::fail("f3d_comm_sendrecv: f3d_comm_init() has not been called");
  F3D_NCCL(R.GroupStart());
  for (int i = 0; i < n_peers; ++i) {
    if (peers[i] < 0 || peers[i] >= R.n_ranks) {  // the own rank is a legal peer: RCCL pairs the send with the recv locally
      (void)R.GroupEnd();
      return f3d::fail("f3d_comm_sendrecv: bad peer %d", peers[i]);
    }
    if (send_count[i])
      F3D_NCCL(R.Send(f3d_ptr<const float>(send_buf) + send_offset[i], send_count[i], ncclFloat, peers[i], R.comm, f3d::stream()));
    if (recv_count[i])
      F3D_NCCL(R.Recv(f3d_ptr<float>(recv_buf) + recv_offset[i], recv_count[i], ncclFloat, peers[i], R.comm, f3d::stream()));
  }
  F3D_NCCL(R.GroupEnd());
  return 0;
}

int f3d_comm_allreduce_max_f32(float* value)
{
  F3D_REQUIRE_READY("f3d_comm_allreduce_max_f32");
  if (!value) return f3d::fail("f3d_comm_allreduce_max_f32: null argument");
  if (!R.comm || R.n_ranks == 1) return 0;
  F3D_HIP(hipMemcpyAsync(R.d_scalar, value, sizeof(float), hipMemcpyHostToDevice, f3d::stream()));
  F3D_NCCL(R.AllReduce(R.d_scalar, R.d_scalar, 1, ncclFloat, ncclMax, R.comm, f3d::stream()));
  F3D_HIP(hipMemcpyAsync(value, R.d_scalar, sizeof(float), hipMemcpyDeviceToHost, f3d::stream()));
  F3D_HIP(hipStreamSynchronize(f3d::stream()));
  return 0;
}

}  // extern "C"
