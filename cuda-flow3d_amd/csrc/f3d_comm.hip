// z-slab communication layer: plane pack/unpack kernels and neighbour exchange on RCCL (ncclSend/ncclRecv over
// xGMI).  The reference is single-GPU; this is new design.  librccl.so.1 is opened lazily with
// dlopen so that single-GPU users never load it.  All traffic runs on the library stream, in order with the kernels.
//
// Rehearsal transport (F3D_COMM_BACKEND=shm, never the default): RCCL refuses two ranks on one device, so a box with
// fewer GPUs than ranks cannot run the one-process-per-rank driver at all.  With this backend the same calls move the
// packed halos through POSIX shared memory (device -> host mailbox of the sender, mailbox -> device of the receiver), which
// lets N processes share one GPU: it exists to test the multi-process orchestration, not to be fast.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <thread>

#include "f3d_internal.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional: older libraries may lack it
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;     // optional, f3d_comm_info only
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;  // optional
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;  // optional
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1;
  float* d_scalar = nullptr;
  unsigned long long sent_bytes = 0, exchanges = 0;  // what this rank has handed to the transport (f3d_comm_info)
  // split exchange (f3d_comm_sendrecv_begin / _end): the transfer runs on its own stream beside the kernels
  hipStream_t side = nullptr;
  hipEvent_t packed = nullptr, arrived = nullptr;
  bool open = false;
} R;

// Exchange timing (f3d_comm_timing*): HIP events on the stream the work runs on.  Three classes of interval:
//   0  a whole blocking exchange on the library stream -- f3d_comm_mark(0) before the pack launch ... f3d_comm_mark(1) after the unpack
//   1  a whole exchange whose transfer runs beside kernels (the overlapped order): the interval includes the interior it hides behind
//   2  the grouped send / recv alone, on whichever stream it was posted to (recorded by grouped_sendrecv itself)
// Off by default and outside every timed region: bench.py switches it on for one extra, untimed solve.
struct ExchangeTimer {
  bool on = false;
  std::vector<hipEvent_t> pool;
  struct Span { hipEvent_t a, b; int cls; unsigned long long bytes; };
  std::vector<Span> spans;
  hipEvent_t open_a = nullptr;
  int open_cls = 0;
  unsigned long long open_bytes0 = 0;
  struct Sum { double us = 0, min_us = 0, max_us = 0; unsigned long long n = 0, bytes = 0; } sum[3];
  hipEvent_t get()
  {
    hipEvent_t e = nullptr;
    if (!pool.empty()) {
      e = pool.back();
      pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
      e = nullptr;
    }
    return e;
  }
} T;

// the shared-memory backend has no second engine: _begin records the request, _end performs it
struct Pending {
  std::vector<size_t> send_offset, send_count, recv_offset, recv_count;
  std::vector<int> peers;
  const float* send_buf = nullptr;
  float* recv_buf = nullptr;
} Q;

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
  R.CommAbort = reinterpret_cast<decltype(R.CommAbort)>(dlsym(R.handle, "ncclCommAbort"));
  R.CommCount = reinterpret_cast<decltype(R.CommCount)>(dlsym(R.handle, "ncclCommCount"));
  R.CommCuDevice = reinterpret_cast<decltype(R.CommCuDevice)>(dlsym(R.handle, "ncclCommCuDevice"));
  R.CommUserRank = reinterpret_cast<decltype(R.CommUserRank)>(dlsym(R.handle, "ncclCommUserRank"));
  return 0;
}

// ---- shared-memory rehearsal transport ---------------------------------------------------------------------------------
constexpr int kShmMaxRanks = 16;
struct ShmHeader {
  std::atomic<unsigned long long> posted[kShmMaxRanks];  // messages this rank has put into its outbox for rank d
  std::atomic<unsigned long long> taken[kShmMaxRanks];   // ... and how many of them rank d has copied out
  std::atomic<unsigned long long> red_seq;               // all-reduce rounds this rank has entered
  float red_val[2];                                      // its contribution, by round parity
};
struct Shm {
  bool active = false;
  std::string session;
  int rank = 0, n_ranks = 1;
  size_t cap_floats = 0;       // capacity of one outbox
  size_t bytes = 0;            // size of one rank's segment
  char* seg[kShmMaxRanks] = {};
  unsigned long long red_round = 0;
} M;

std::string shm_name(const std::string& session, int rank) { return "/f3d_" + session + "_" + std::to_string(rank); }
ShmHeader* shm_header(int rank) { return reinterpret_cast<ShmHeader*>(M.seg[rank]); }
float* shm_outbox(int owner, int dest)
{
  return reinterpret_cast<float*>(M.seg[owner] + 4096) + static_cast<size_t>(dest) * M.cap_floats;
}

int shm_attach(int rank)  // maps rank's segment, waiting for its owner to create it
{
  if (M.seg[rank]) return 0;
  const std::string name = shm_name(M.session, rank);
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(60);
  for (;;) {
    const int fd = shm_open(name.c_str(), O_RDWR, 0600);
    if (fd >= 0) {
      off_t size = lseek(fd, 0, SEEK_END);
      if (size >= static_cast<off_t>(M.bytes)) {
        void* p = mmap(nullptr, M.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return f3d::fail("f3d_comm (shm): cannot map %s", name.c_str());
        M.seg[rank] = static_cast<char*>(p);
        return 0;
      }
      close(fd);
    }
    if (std::chrono::steady_clock::now() > deadline) return f3d::fail("f3d_comm (shm): rank %d never created %s", rank, name.c_str());
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
  }
}

template <typename Pred>
int shm_wait(Pred done, const char* what, int peer)
{
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
  for (unsigned spin = 0; !done(); ++spin) {
    if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spin & 0xfff) == 0 && std::chrono::steady_clock::now() > deadline)
      return f3d::fail("f3d_comm (shm): rank %d timed out waiting for rank %d (%s)", M.rank, peer, what);
  }
  return 0;
}

int shm_init(const char* id128, int rank, int n_ranks)
{
  if (n_ranks > kShmMaxRanks) return f3d::fail("f3d_comm (shm): at most %d ranks", kShmMaxRanks);
  M.session.assign(id128 + 7, strnlen(id128 + 7, 100));
  M.rank = rank;
  M.n_ranks = n_ranks;
  const char* cap = std::getenv("F3D_SHM_CAP_MB");
  M.cap_floats = static_cast<size_t>(cap ? std::atol(cap) : 64) * (1u << 20) / sizeof(float);
  M.bytes = 4096 + static_cast<size_t>(n_ranks) * M.cap_floats * sizeof(float);  // pages are committed on first touch
  const std::string name = shm_name(M.session, rank);
  shm_unlink(name.c_str());
  const int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return f3d::fail("f3d_comm (shm): cannot create %s", name.c_str());
  // a peer maps the segment only once it has its final size; a fresh tmpfs file reads as zeros, so every counter in
  // the header starts at 0 without this rank writing it
  if (ftruncate(fd, static_cast<off_t>(M.bytes)) != 0) {
    close(fd);
    return f3d::fail("f3d_comm (shm): cannot size %s to %zu bytes", name.c_str(), M.bytes);
  }
  void* p = mmap(nullptr, M.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return f3d::fail("f3d_comm (shm): cannot map %s", name.c_str());
  M.seg[rank] = static_cast<char*>(p);  // a fresh tmpfs file reads as zeros: all counters start at 0
  M.active = true;
  M.red_round = 0;
  return 0;
}

void shm_destroy()
{
  if (!M.active) return;
  for (int r = 0; r < kShmMaxRanks; ++r)
    if (M.seg[r]) {
      munmap(M.seg[r], M.bytes);
      M.seg[r] = nullptr;
    }
  shm_unlink(shm_name(M.session, M.rank).c_str());
  M.active = false;
}

int shm_sendrecv(const float* send_buf, const size_t* send_offset, const size_t* send_count, float* recv_buf,
                 const size_t* recv_offset, const size_t* recv_count, const int* peers, int n_peers)
{
  F3D_HIP(hipStreamSynchronize(f3d::stream()));  // the pack kernel has filled the staging buffer
  ShmHeader* mine = shm_header(M.rank);
  for (int i = 0; i < n_peers; ++i) {
    const int d = peers[i];
    if (d < 0 || d >= M.n_ranks) return f3d::fail("f3d_comm_sendrecv: bad peer %d", d);
    if (!send_count[i]) continue;
    R.sent_bytes += send_count[i] * sizeof(float);
    if (send_count[i] > M.cap_floats)
      return f3d::fail("f3d_comm (shm): message of %zu floats exceeds the outbox (%zu); raise F3D_SHM_CAP_MB", send_count[i], M.cap_floats);
    if (shm_wait([&] { return mine->taken[d].load(std::memory_order_acquire) == mine->posted[d].load(std::memory_order_relaxed); },
                 "outbox free", d))
      return 1;
    F3D_HIP(hipMemcpy(shm_outbox(M.rank, d), send_buf + send_offset[i], send_count[i] * sizeof(float), hipMemcpyDeviceToHost));
    mine->posted[d].fetch_add(1, std::memory_order_release);
  }
  for (int i = 0; i < n_peers; ++i) {
    const int s = peers[i];
    if (!recv_count[i]) continue;
    if (shm_attach(s)) return 1;
    ShmHeader* theirs = shm_header(s);
    if (shm_wait([&] { return theirs->posted[M.rank].load(std::memory_order_acquire) > theirs->taken[M.rank].load(std::memory_order_relaxed); },
                 "message", s))
      return 1;
    F3D_HIP(hipMemcpy(recv_buf + recv_offset[i], shm_outbox(s, M.rank), recv_count[i] * sizeof(float), hipMemcpyHostToDevice));
    theirs->taken[M.rank].fetch_add(1, std::memory_order_release);
  }
  ++R.exchanges;
  return 0;
}

int shm_allreduce_max(float* value)
{
  ShmHeader* mine = shm_header(M.rank);
  const unsigned long long round = ++M.red_round;
  mine->red_val[round & 1] = *value;
  mine->red_seq.store(round, std::memory_order_release);
  float m = *value;
  for (int r = 0; r < M.n_ranks; ++r) {
    if (r == M.rank) continue;
    if (shm_attach(r)) return 1;
    ShmHeader* theirs = shm_header(r);
    // a rank can be at most one round ahead (it needs everybody's value of this round to leave it), so the slot of this
    // round's parity still holds this round's value when we read it
    if (shm_wait([&] { return theirs->red_seq.load(std::memory_order_acquire) >= round; }, "all-reduce", r)) return 1;
    const float v = theirs->red_val[round & 1];
    if (v > m) m = v;
  }
  *value = m;
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

// the same for up to kMaxSeg (destination, source, plane range) triples in one launch: a whole in-process exchange
struct CopyTable {
  float* dst[kMaxSeg];
  const float* src[kMaxSeg];
  int dst_plane0[kMaxSeg], src_plane0[kMaxSeg], count[kMaxSeg];
};
__global__ __launch_bounds__(256) void k_copy_planes_batched(CopyTable t, int width, int height, int Hc, int pitch, int max_count)
{
  const int seg = blockIdx.z / max_count;
  const int p = blockIdx.z - seg * max_count;
  if (p >= t.count[seg]) return;
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  if (x >= width || y >= height) return;
  const size_t row = static_cast<size_t>(y) * pitch + x;
  const size_t plane = static_cast<size_t>(Hc) * pitch;
  t.dst[seg][(t.dst_plane0[seg] + p) * plane + row] = t.src[seg][(t.src_plane0[seg] + p) * plane + row];
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

bool want_shm()
{
  const char* e = std::getenv("F3D_COMM_BACKEND");
  return e && std::strcmp(e, "shm") == 0;
}

}  // namespace

namespace {

// One grouped exchange.  Peers are validated BEFORE the group is opened, and once it is open ncclGroupEnd is always
// called: a Send/Recv that fails is remembered, the group is closed, then the error is returned -- a communicator is
// never left with an open group.  A rank that cannot post its part aborts the communicator (ncclCommAbort) so that the
// peers already blocked in their grouped recv fail instead of waiting for ever.
int grouped_sendrecv(const float* send_buf, const size_t* send_offset, const size_t* send_count, float* recv_buf,
                     const size_t* recv_offset, const size_t* recv_count, const int* peers, int n_peers, hipStream_t on,
                     const char* who)
{
  for (int i = 0; i < n_peers; ++i)
    if (peers[i] < 0 || peers[i] >= R.n_ranks)  // the own rank is a legal peer: RCCL pairs the send with the recv locally
      return f3d::fail("%s: bad peer %d", who, peers[i]);
  hipEvent_t ta = nullptr, tb = nullptr;
  if (T.on && (ta = T.get()) != nullptr) (void)hipEventRecord(ta, on);
  F3D_NCCL(R.GroupStart());
  ncclResult_t first = ncclSuccess;
  for (int i = 0; i < n_peers && first == ncclSuccess; ++i) {
    if (send_count[i]) first = R.Send(send_buf + send_offset[i], send_count[i], ncclFloat, peers[i], R.comm, on);
    if (first == ncclSuccess && recv_count[i]) first = R.Recv(recv_buf + recv_offset[i], recv_count[i], ncclFloat, peers[i], R.comm, on);
  }
  const ncclResult_t end = R.GroupEnd();
  if (first != ncclSuccess || end != ncclSuccess) {
    const ncclResult_t e = first != ncclSuccess ? first : end;
    if (R.CommAbort && R.comm) {
      (void)R.CommAbort(R.comm);
      R.comm = nullptr;
    }
    return f3d::fail("%s: RCCL error %d (%s); the communicator was aborted", who, static_cast<int>(e), R.GetErrorString ? R.GetErrorString(e) : "?");
  }
  unsigned long long bytes = 0;
  for (int i = 0; i < n_peers; ++i) bytes += send_count[i] * sizeof(float);
  R.sent_bytes += bytes;
  ++R.exchanges;
  if (ta && (tb = T.get()) != nullptr) {
    (void)hipEventRecord(tb, on);
    T.spans.push_back({ta, tb, 2, bytes});
  }
  return 0;
}

}  // namespace


extern "C" {

int f3d_comm_unique_id(void* id128)
{
  if (!id128) return f3d::fail("f3d_comm_unique_id: null argument");
  if (want_shm()) {  // "f3dshm:" + a session name unique to this launch
    std::memset(id128, 0, 128);
    std::snprintf(static_cast<char*>(id128), 128, "f3dshm:%d_%llx", static_cast<int>(getpid()),
                  static_cast<unsigned long long>(std::chrono::steady_clock::now().time_since_epoch().count()));
    return 0;
  }
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
  if (R.comm || M.active) return f3d::fail("f3d_comm_init: communicator already initialised");
  if (std::memcmp(id128, "f3dshm:", 7) == 0) {
    if (shm_init(static_cast<const char*>(id128), rank, n_ranks)) return 1;
    R.rank = rank;
    R.n_ranks = n_ranks;
    return 0;
  }
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
  shm_destroy();
  R.open = false;
  if (R.side) {
    (void)hipStreamSynchronize(R.side);
    (void)hipEventDestroy(R.packed);
    (void)hipEventDestroy(R.arrived);
    (void)hipStreamDestroy(R.side);
    R.side = nullptr;
    R.packed = R.arrived = nullptr;
  }
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
  R.sent_bytes = R.exchanges = 0;
  return 0;
}

int f3d_comm_rank(int* rank, int* n_ranks)
{
  if (rank) *rank = R.rank;
  if (n_ranks) *n_ranks = R.n_ranks;
  return 0;
}

int f3d_comm_info(int* backend, int* comm_ranks, int* comm_rank, int* comm_device, unsigned long long* sent_bytes,
                  unsigned long long* exchanges)
{
  int be = 0, n = 0, r = -1, dev = -1;
  if (M.active) {
    be = 2;
    n = M.n_ranks;
    r = M.rank;
  } else if (R.comm) {
    be = 1;  // the counts are the communicator's own answers, not what f3d_comm_init was told
    if (R.CommCount) F3D_NCCL(R.CommCount(R.comm, &n));
    if (R.CommUserRank) F3D_NCCL(R.CommUserRank(R.comm, &r));
    if (R.CommCuDevice) F3D_NCCL(R.CommCuDevice(R.comm, &dev));
  }
  if (backend) *backend = be;
  if (comm_ranks) *comm_ranks = n;
  if (comm_rank) *comm_rank = r;
  if (comm_device) *comm_device = dev;
  if (sent_bytes) *sent_bytes = R.sent_bytes;
  if (exchanges) *exchanges = R.exchanges;
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

int f3d_copy_plane_segments(const f3d_devptr* dst, const int* dst_plane0, const f3d_devptr* src, const int* src_plane0, const int* count,
                            int n_segments, size_t width, size_t height)
{
  F3D_REQUIRE_READY("f3d_copy_plane_segments");
  if (n_segments < 0 || (n_segments > 0 && (!dst || !dst_plane0 || !src || !src_plane0 || !count)))
    return f3d::fail("f3d_copy_plane_segments: bad argument");
  const f3d_size4& c = f3d::container();
  for (int i = 0; i < n_segments; i += kMaxSeg) {
    const int n = std::min(kMaxSeg, n_segments - i);
    CopyTable t = {};
    int max_count = 0;
    for (int k = 0; k < n; ++k) {
      if (check_planes(dst_plane0[i + k], count[i + k], width, height, "f3d_copy_plane_segments") ||
          check_planes(src_plane0[i + k], count[i + k], width, height, "f3d_copy_plane_segments"))
        return 1;
      t.dst[k] = f3d_ptr<float>(dst[i + k]);
      t.src[k] = f3d_ptr<const float>(src[i + k]);
      t.dst_plane0[k] = dst_plane0[i + k];
      t.src_plane0[k] = src_plane0[i + k];
      t.count[k] = count[i + k];
      max_count = std::max(max_count, count[i + k]);
    }
    if (max_count == 0) continue;
    const dim3 grid((width + 63) / 64, (height + 3) / 4, n * max_count), block(64, 4, 1);
    hipLaunchKernelGGL(k_copy_planes_batched, grid, block, 0, f3d::stream(), t, static_cast<int>(width), static_cast<int>(height),
                       static_cast<int>(c.height), static_cast<int>(c.pitch / sizeof(float)), max_count);
    F3D_HIP(hipGetLastError());
  }
  return 0;
}

int f3d_comm_sendrecv(f3d_devptr send_buf, const size_t* send_offset, const size_t* send_count, f3d_devptr recv_buf,
                      const size_t* recv_offset, const size_t* recv_count, const int* peers, int n_peers)
{
  F3D_REQUIRE_READY("f3d_comm_sendrecv");
  if (M.active)
    return shm_sendrecv(f3d_ptr<const float>(send_buf), send_offset, send_count, f3d_ptr<float>(recv_buf), recv_offset, recv_count,
                        peers, n_peers);
  if (!R.comm) return f3d::fail("f3d_comm_sendrecv: f3d_comm_init() has not been called");
  return grouped_sendrecv(f3d_ptr<const float>(send_buf), send_offset, send_count, f3d_ptr<float>(recv_buf), recv_offset, recv_count,
                          peers, n_peers, f3d::stream(), "f3d_comm_sendrecv");
}

int f3d_comm_sendrecv_begin(f3d_devptr send_buf, const size_t* send_offset, const size_t* send_count, f3d_devptr recv_buf,
                            const size_t* recv_offset, const size_t* recv_count, const int* peers, int n_peers)
{
  F3D_REQUIRE_READY("f3d_comm_sendrecv_begin");
  if (R.open) return f3d::fail("f3d_comm_sendrecv_begin: the previous exchange has not been ended");
  if (M.active) {
    Q.send_offset.assign(send_offset, send_offset + n_peers);
    Q.send_count.assign(send_count, send_count + n_peers);
    Q.recv_offset.assign(recv_offset, recv_offset + n_peers);
    Q.recv_count.assign(recv_count, recv_count + n_peers);
    Q.peers.assign(peers, peers + n_peers);
    Q.send_buf = f3d_ptr<const float>(send_buf);
    Q.recv_buf = f3d_ptr<float>(recv_buf);
    // the staging buffer must be complete now: kernels issued after this call may overwrite what was packed from
    F3D_HIP(hipStreamSynchronize(f3d::stream()));
    R.open = true;
    return 0;
  }
  if (!R.comm) return f3d::fail("f3d_comm_sendrecv_begin: f3d_comm_init() has not been called");
  if (!R.side) {
    F3D_HIP(hipStreamCreateWithFlags(&R.side, hipStreamNonBlocking));
    F3D_HIP(hipEventCreateWithFlags(&R.packed, hipEventDisableTiming));
    F3D_HIP(hipEventCreateWithFlags(&R.arrived, hipEventDisableTiming));
  }
  F3D_HIP(hipEventRecord(R.packed, f3d::stream()));
  F3D_HIP(hipStreamWaitEvent(R.side, R.packed, 0));
  if (grouped_sendrecv(f3d_ptr<const float>(send_buf), send_offset, send_count, f3d_ptr<float>(recv_buf), recv_offset, recv_count, peers,
                       n_peers, R.side, "f3d_comm_sendrecv_begin"))
    return 1;
  F3D_HIP(hipEventRecord(R.arrived, R.side));
  R.open = true;
  return 0;
}

int f3d_comm_sendrecv_end(void)
{
  F3D_REQUIRE_READY("f3d_comm_sendrecv_end");
  if (!R.open) return f3d::fail("f3d_comm_sendrecv_end: no exchange is open");
  R.open = false;
  if (M.active)
    return shm_sendrecv(Q.send_buf, Q.send_offset.data(), Q.send_count.data(), Q.recv_buf, Q.recv_offset.data(), Q.recv_count.data(),
                        Q.peers.data(), static_cast<int>(Q.peers.size()));
  F3D_HIP(hipStreamWaitEvent(f3d::stream(), R.arrived, 0));
  return 0;
}

int f3d_comm_timing(int enable)
{
  F3D_REQUIRE_READY("f3d_comm_timing");
  if (enable) {
    for (auto& c : T.sum) c = ExchangeTimer::Sum();
    for (auto& sp : T.spans) {
      T.pool.push_back(sp.a);
      T.pool.push_back(sp.b);
    }
    T.spans.clear();
  }
  T.on = enable != 0;
  return 0;
}

int f3d_comm_mark(int what, int cls)
{
  if (!T.on) return 0;
  F3D_REQUIRE_READY("f3d_comm_mark");
  if (what == 0) {
    if (!T.open_a) T.open_a = T.get();
    if (T.open_a) (void)hipEventRecord(T.open_a, f3d::stream());
    T.open_cls = cls == 1 ? 1 : 0;
    T.open_bytes0 = R.sent_bytes;
  } else if (T.open_a) {
    hipEvent_t b = T.get();
    if (b) {
      (void)hipEventRecord(b, f3d::stream());
      T.spans.push_back({T.open_a, b, T.open_cls, R.sent_bytes - T.open_bytes0});
      T.open_a = nullptr;
    }
  }
  return 0;
}

int f3d_comm_timing_read(int cls, double* total_us, unsigned long long* count, double* min_us, double* max_us, unsigned long long* bytes)
{
  F3D_REQUIRE_READY("f3d_comm_timing_read");
  if (cls < 0 || cls > 2) return f3d::fail("f3d_comm_timing_read: class %d (0 blocking exchange, 1 overlapped exchange, 2 transfer alone)", cls);
  if (!T.spans.empty()) {
    F3D_HIP(hipStreamSynchronize(f3d::stream()));
    if (R.side) F3D_HIP(hipStreamSynchronize(R.side));
    for (auto& sp : T.spans) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
        auto& c = T.sum[sp.cls];
        const double us = static_cast<double>(ms) * 1e3;
        c.min_us = c.n == 0 ? us : std::min(c.min_us, us);
        c.max_us = c.n == 0 ? us : std::max(c.max_us, us);
        c.us += us;
        c.bytes += sp.bytes;
        ++c.n;
      }
      T.pool.push_back(sp.a);
      T.pool.push_back(sp.b);
    }
    T.spans.clear();
  }
  const auto& c = T.sum[cls];
  if (total_us) *total_us = c.us;
  if (count) *count = c.n;
  if (min_us) *min_us = c.min_us;
  if (max_us) *max_us = c.max_us;
  if (bytes) *bytes = c.bytes;
  return 0;
}

int f3d_comm_allreduce_max_f32(float* value)
{
  F3D_REQUIRE_READY("f3d_comm_allreduce_max_f32");
  if (!value) return f3d::fail("f3d_comm_allreduce_max_f32: null argument");
  if (M.active) return M.n_ranks == 1 ? 0 : shm_allreduce_max(value);
  if (!R.comm || R.n_ranks == 1) return 0;
  F3D_HIP(hipMemcpyAsync(R.d_scalar, value, sizeof(float), hipMemcpyHostToDevice, f3d::stream()));
  F3D_NCCL(R.AllReduce(R.d_scalar, R.d_scalar, 1, ncclFloat, ncclMax, R.comm, f3d::stream()));
  F3D_HIP(hipMemcpyAsync(value, R.d_scalar, sizeof(float), hipMemcpyDeviceToHost, f3d::stream()));
  F3D_HIP(hipStreamSynchronize(f3d::stream()));
  return 0;
}

}  // extern "C"
